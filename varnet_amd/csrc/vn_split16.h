// Host entry points of vn_split16.hip (kept out of vn_internal.h, which every kernel's source hash covers).
#pragma once
#include "vn_internal.h"

// Networks of the 8-wave family with hidden widths 33..64 and 2..7 hidden layers (6 at widths > 50), dim <= 3: the value
// (vn_forward) and the strong residual (vn_residual) with the hidden-layer products on the bf16 matrix pipe as six products of
// exact bf16 pieces -- fp32-class accuracy, the parity bars of vn_pgrad16 / vn_taylor16.  hipErrorInvalidValue otherwise.
bool vn_split16_supported(const VnNet& net);      // the instantiation exists (any dim; the residual and the gradient need dim <= 3)
hipError_t vn_split16_forward(const VnNet& net, const float* theta, const float* X, long n, float* u, int ncu, hipStream_t s);
hipError_t vn_split16_residual(const VnNet& net, const float* theta, const float* X, const float* diff, const float* vel,
                               const float* src, const float* ddx, int td, long n, float* u, float* res, int ncu, hipStream_t s);
// (u, du/dx_d) in one pass, both sweeps on the bf16 pipe (the outputs of vn_pgrad16_launch: out_u [n], out_g [n, dim], out_pack [n, 4])
hipError_t vn_split16_pgrad(const VnNet& net, const float* theta, const float* X, long n, float* out_u, float* out_g, float* out_pack,
                            int ncu, hipStream_t s);
