// Host entry point of vn_pgrad16.hip (kept out of vn_internal.h, which every kernel's source hash covers).
#pragma once
#include "vn_internal.h"

// Value and input gradient at n points in one pass (value forward + value-adjoint sweep to the inputs, 2 F_pt per point;
// TFModel.py:536-541); every network vn_fused16_net_supported accepts, dim <= 3.  Outputs, any of which may be nullptr:
//   out_u [n], out_g [n, net.dim]                      separate arrays (vn_forward_grad; out_g == nullptr and out_pack == nullptr:
//                                                      value only, F_pt per point: vn_forward)
//   out_pack [n, 4] = (u, du/dx_0, du/dx_1, du/dx_2)   one 16-byte record per point (absent coordinates 0): what the
//                                                      de-duplicated assembly gathers per row -- one cache line, not three
// ncu = CUs of the device; wgs_per_cu = 0: as many resident workgroups per CU as fit, at most 2.
hipError_t vn_pgrad16_launch(const VnNet& net, const float* theta, const float* X, long n, float* out_u, float* out_g,
                             float* out_pack, int ncu, int wgs_per_cu, hipStream_t s);
