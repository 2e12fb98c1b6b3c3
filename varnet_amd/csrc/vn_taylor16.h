// Host entry point of vn_taylor16.hip (kept out of vn_internal.h, which every kernel's source hash covers).
#pragma once
#include "vn_internal.h"

// res = -u_t + diff Lap(u) - (vel - ddx) . grad(u) + src at n points (TFModel.py:743-754), second-order forward mode on the
// matrix pipe; u may be nullptr.  Networks the 8-wave fused kernel serves (vn_fused16_net_supported), dim <= 3;
// hipErrorInvalidValue otherwise.  src, ddx may be nullptr.  ncu = CUs of the device.
// what vn_taylor16_residual takes (the caller routes everything else to the per-thread kernel of vn_pointwise.hip)
inline bool vn_taylor16_supported(const VnNet& net, int td) { return net.dim <= 3 && net.d_in <= 8 && net.dim + (td ? 1 : 0) <= net.d_in; }
hipError_t vn_taylor16_residual(const VnNet& net, const float* theta, const float* X, const float* diff, const float* vel,
                                const float* src, const float* ddx, int td, long n, float* u, float* res, int ncu, hipStream_t s);

// fp64 forms on the fp64 matrix pipe (vn_taylor16d.hip): res == nullptr -> the value alone (vn_forward_f64), else the strong
// residual in second-order forward mode (vn_residual_f64).  vn_taylor16d_supported: networks of the 8-wave family whose
// double-precision weight images fit the LDS (up to 6 x 50, 5 x 64, 8 x 32); others stay on the per-thread kernels.
bool vn_taylor16d_supported(const VnNet& net);
hipError_t vn_taylor16d_launch(const VnNet& net, const double* theta, const double* X, const double* diff, const double* vel,
                               const double* src, const double* ddx, int td, long n, double* u, double* res, int ncu, hipStream_t s);
