// Host entry point of vn_taylor16.hip (kept out of vn_internal.h, which every kernel's source hash covers).
#pragma once
#include "vn_internal.h"

// res = -u_t + diff Lap(u) - (vel - ddx) . grad(u) + src at n points (TFModel.py:743-754), second-order forward mode on the
// matrix pipe; u may be nullptr.  Networks the 8-wave fused kernel serves (vn_fused16_net_supported), dim <= 3;
// hipErrorInvalidValue otherwise.  src, ddx may be nullptr.  ncu = CUs of the device.
hipError_t vn_taylor16_residual(const VnNet& net, const float* theta, const float* X, const float* diff, const float* vel,
                                const float* src, const float* ddx, int td, long n, float* u, float* res, int ncu, hipStream_t s);
