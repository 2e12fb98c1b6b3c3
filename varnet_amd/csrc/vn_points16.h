// Shared by the point kernels of the 8-wave family that need nothing but the weight images in LDS (vn_pgrad16.hip: value and
// input gradient; vn_taylor16.hip: strong residual): the LDS layout of the images and the prologue that stages them.
#pragma once
#include "vn_fused16_common.h"

// The (hidden layers, k-steps per layer) instantiations of the point kernels = those of vn_fused16.hip.  Networks of
// VN_POINT16_SPLIT_CASES -- hidden widths 33..64 (KS = 13, 16) with 2..7 hidden layers, 6 beyond 50 wide -- are served by the
// bf16-piece kernels of vn_split16.hip; the f32-MFMA forms of those networks (vn_pgrad16 / vn_taylor16) are what the tests compare
// them with and are instantiated in the tests' cross-check library only (-DVN_XCHECK_F32_POINT, Makefile target xcheck).
#define VN_POINT16_SPLIT_CASES(X) \
  X(2, 13) X(3, 13) X(4, 13) X(5, 13) X(6, 13) X(7, 13) \
  X(2, 16) X(3, 16) X(4, 16) X(5, 16) X(6, 16)
#define VN_POINT16_F32_ONLY_CASES(X) \
  X(1, 5) X(2, 5) X(3, 5) X(4, 5) X(5, 5) X(6, 5) X(7, 5) X(8, 5)  \
  X(1, 8) X(2, 8) X(3, 8) X(4, 8) X(5, 8) X(6, 8) X(7, 8) X(8, 8)  \
  X(1, 13) X(8, 13) X(1, 16)
#ifdef VN_XCHECK_F32_POINT
#define VN_POINT16_F32_CASES(X) VN_POINT16_F32_ONLY_CASES(X) VN_POINT16_SPLIT_CASES(X)
#else
#define VN_POINT16_F32_CASES(X) VN_POINT16_F32_ONLY_CASES(X)
#endif

namespace vn16 {

template <int L, int KS>
struct PLay {
  static constexpr int HP = 4 * KS;
  static constexpr int HPWS = al4(HP * WS);
  static constexpr int W1_OFF = 0;                          // [8][WS]
  static constexpr int WH_OFF = al4(8 * WS);                // [L-1][HP][WS]
  static constexpr int BI_OFF = WH_OFF + (L - 1) * HPWS;    // [L][64] biases in (tile, g, i) order
  static constexpr int WO_OFF = BI_OFF + L * 64;            // [4*KS]
  static constexpr int TOTAL = WO_OFF + al4(4 * KS);
};

// Weight images as in vn_fused16.hip: parameters are read in their own row-major order (coalesced), every load is issued before
// anything waits, and scattered into the images [in-feature][out-position] (row stride WS); padding rows / columns are exact
// zeros.  Ends with the images written but NOT yet visible to other waves: the caller synchronises.
template <int L, int KS>
__device__ __forceinline__ void stage_weight_images(const VnNet& net, const float* theta, float* lds, int tid) {
  float* W1 = lds + PLay<L, KS>::W1_OFF;
  float* WH = lds + PLay<L, KS>::WH_OFF;
  float* BI = lds + PLay<L, KS>::BI_OFF;
  float* WO = lds + PLay<L, KS>::WO_OFF;
  const int d_in = net.d_in, H1 = net.H[1];
  constexpr int NSRC = (PLay<L, KS>::HP * PLay<L, KS>::HP + NTHREADS - 1) / NTHREADS;
  static_assert(8 * 64 <= NTHREADS, "layer 1: one parameter per thread");
  float v1 = 0.f, vh[L > 1 ? L - 1 : 1][NSRC];
  if (tid < d_in * H1) v1 = theta[net.woff[1] + tid];
#pragma unroll
  for (int l = 2; l <= L; ++l) {
    const int n = net.H[l - 1] * net.H[l];
    const float* src = theta + net.woff[l];
#pragma unroll
    for (int it = 0; it < NSRC; ++it) {
      const int j = tid + it * NTHREADS;
      vh[l - 2][it] = j < n ? src[j] : 0.f;
    }
  }
  static_assert(L * 64 <= NTHREADS && 4 * KS <= NTHREADS, "one bias / output weight per thread");
  float vb = 0.f, vo = 0.f;
  if (tid < L * 64) {
    const int l = tid / 64 + 1, idx = tid % 64;
    const int mt = idx >> 4, g = (idx >> 2) & 3, r = idx & 3;       // [tile][g][i]
    const int ks = 4 * mt + r, f = 4 * ks + g;
    vb = (ks < KS && f < net.H[l]) ? theta[net.boff[l] + f] : 0.f;
  }
  if (tid < 4 * KS) vo = (tid < net.H[L]) ? theta[net.woff[L + 1] + tid] : 0.f;
  static_assert(PLay<L, KS>::BI_OFF % 4 == 0, "16-byte zero fill");
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < PLay<L, KS>::BI_OFF / 4; i += NTHREADS) reinterpret_cast<f32x4a*>(lds)[i] = z4;                    // W1 | WH
  __syncthreads();
  if (tid < d_in * H1) {
    const int k = tid / H1, f = tid - k * H1;
    W1[k * WS + vpos(f >> 2, f & 3)] = v1;
  }
#pragma unroll
  for (int l = 2; l <= L; ++l) {
    float* Wl = WH + (l - 2) * PLay<L, KS>::HPWS;
    const int Hout = net.H[l], n = net.H[l - 1] * Hout;
    const int dq = NTHREADS / Hout, dr = NTHREADS - dq * Hout;      // j -> j + NTHREADS: k += dq, f += dr (one carry)
    int k = tid / Hout, f = tid - k * Hout;
#pragma unroll
    for (int it = 0; it < NSRC; ++it) {
      if (tid + it * NTHREADS < n) Wl[k * WS + vpos(f >> 2, f & 3)] = vh[l - 2][it];
      f += dr; k += dq;
      if (f >= Hout) { f -= Hout; ++k; }
    }
  }
  if (tid < L * 64) BI[tid] = vb;
  if (tid < 4 * KS) WO[tid] = vo;
}

}  // namespace vn16
