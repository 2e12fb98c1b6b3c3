// Device helpers shared by the 8-wave kernels of the fused family (vn_fused16.hip: the training step;
// vn_pgrad16.hip: value + input gradient at points): the feature <-> (k-step, lane group, accumulator row) layout,
// activation arithmetic on register pairs, cross-lane sums.
#pragma once
#include "vn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4a __attribute__((may_alias));

namespace vn16 {

constexpr int NW = 8;
constexpr int NTHREADS = 64 * NW;
constexpr int TILE = 128;
constexpr int CW = 16;        // points per wave
constexpr int WS = 65;        // weight image row stride
constexpr int KS0 = 2;        // input layer k-steps (d_in <= 8)

__host__ __device__ constexpr int al4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int vpos(int ks, int g) { return 16 * (ks >> 2) + 4 * g + (ks & 3); }
__host__ __device__ constexpr int vks(int pos) { return 4 * (pos >> 4) + (pos & 3); }
__host__ __device__ constexpr int vfeat(int pos) { return 4 * vks(pos) + ((pos >> 2) & 3); }

// What the branches over padding-only k-steps / row tiles (KSKIP, live_k / live_m) rely on: k-step ks holds features 4ks..4ks+3 and
// nothing else, row tile m (positions 16m..16m+15) holds features 16m..16m+15 and nothing else -- so "width H" bounds the live
// k-steps by ceil(H/4) and the live row tiles by ceil(H/16).  A change of vpos / vfeat that breaks this must not compile.
__host__ __device__ constexpr bool layout_ties_ksteps_and_tiles_to_features() {
  for (int ks = 0; ks < 16; ++ks)
    for (int g = 0; g < 4; ++g)
      if (vks(vpos(ks, g)) != ks || vfeat(vpos(ks, g)) != 4 * ks + g) return false;
  for (int pos = 0; pos < 64; ++pos)
    if (vfeat(pos) < 16 * (pos >> 4) || vfeat(pos) >= 16 * (pos >> 4) + 16 || vpos(vks(pos), (pos >> 2) & 3) != pos) return false;
  return true;
}
static_assert(layout_ties_ksteps_and_tiles_to_features(), "KSKIP: k-step = feature >> 2, row tile = feature >> 4");
__host__ __device__ constexpr int mtiles(int KS) { return (KS + 3) / 4; }

__device__ __forceinline__ float opaque(float x) {
  asm("" : "+v"(x));
  return x;
}

// Activation (uniform over the hidden layers): sigmoid, or tanh = 2*sigmoid(2z) - 1 (VarNet.py:97).  Everything the
// kernel needs is a function of the stored activation a:  sigma' = a(1-a) | 1-a^2,  sigma''/sigma' = 1-2a | -2a.
template <bool TANH>
__device__ __forceinline__ float act_exp(float z) {      // the exponential inside the sigmoid
  return __builtin_amdgcn_exp2f((TANH ? -2.8853900817779268f : -1.4426950408889634f) * z);
}
template <bool TANH>
__device__ __forceinline__ float act_fin(float e) {      // e = exp(-z) | exp(-2z)  ->  activation
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  return TANH ? __builtin_fmaf(2.f, s, -1.f) : s;
}
template <bool TANH>
__device__ __forceinline__ float act_d1(float a) { return TANH ? __builtin_fmaf(-a, a, 1.f) : a * (1.f - a); }
template <bool TANH>
__device__ __forceinline__ float act_d2r(float a) { return TANH ? -2.f * a : 1.f - 2.f * a; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Per-lane value arrays indexed by k-step, kept as even-aligned REGISTER PAIRS: k-steps 2j and 2j+1 share a pair, so
// the elementwise chains of the reverse pass run as v_pk_mul_f32 / v_pk_fma_f32 on two k-steps per instruction
// (packed fp32 issues at the scalar rate on gfx950, and every vector instruction costs matrix time here).  A pair of
// accumulator rows (ks, ks+1), ks even, of an MFMA tile is a register pair already.
template <int N>
struct PA {
  static constexpr int NP = (N + 1) / 2;
  f32x2 p[NP];
  __device__ __forceinline__ float operator[](int i) const { return p[i >> 1][i & 1]; }
  __device__ __forceinline__ void set(int i, float v) { p[i >> 1][i & 1] = v; }
};
__device__ __forceinline__ f32x2 opaque2(f32x2 x) {
  asm("" : "+v"(x));
  return x;
}
template <bool TANH>
__device__ __forceinline__ f32x2 act_exp2(f32x2 z) {      // two exponentials: one packed scale, two v_exp
  const float c = TANH ? -2.8853900817779268f : -1.4426950408889634f;
  const f32x2 t = z * f32x2{c, c};
  return f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
}
template <bool TANH>
__device__ __forceinline__ f32x2 act_fin2(f32x2 e) {
  const f32x2 d = e + f32x2{1.f, 1.f};
  const f32x2 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return TANH ? (s * f32x2{2.f, 2.f} - f32x2{1.f, 1.f}) : s;
}
template <bool TANH>
__device__ __forceinline__ f32x2 act_d1_2(f32x2 a) {
  const f32x2 one = {1.f, 1.f};
  return TANH ? (one - a * a) : (a - a * a);
}
template <bool TANH>
__device__ __forceinline__ f32x2 act_d2r_2(f32x2 a) {
  const f32x2 one = {1.f, 1.f}, two = {2.f, 2.f};
  return TANH ? (-two * a) : (one - two * a);
}

// x summed over the four 16-lane rows of the wave, in every lane: (r0 + r1) + (r2 + r3)
__device__ __forceinline__ float rowsum4(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
  const float s = a + b;
  float c = s, d = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));     // c = [lo lo], d = [hi hi]
  return c + d;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// x summed over the 16 lanes of its row, in every lane of the row (fixed order: quads, then 8, then 16)
__device__ __forceinline__ float rowsum16(float x) {
  x += dpp_f32<0xB1>(x);         // quad_perm [1,0,3,2]
  x += dpp_f32<0x4E>(x);         // quad_perm [2,3,0,1]
  x += dpp_f32<0x141>(x);        // row_half_mirror
  x += dpp_f32<0x140>(x);        // row_mirror
  return x;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Edge rows: every lane group holds a partial sum (its share of the k index) of each of the NV edge features; group g
// must end up with the TOTAL of feature g.  A reduce-scatter instead of NV all-reduces: at each of the two exchange
// steps a lane sends only the partials its partner keeps, so NV = 2 takes 2 cross-lane moves instead of 4, NV = 4 takes
// 3 instead of 8 (ds_bpermute costs ~14 issue cycles on the shared vector path).  Groups >= NV get 0.
template <int NV>
__device__ __forceinline__ float edge_reduce_scatter(const float (&e)[NV], int g) {
  if constexpr (NV == 2) {
    // step 1 (partner g^1): keep feature g&1, hand the other one over
    const float keep = (g & 1) ? e[1] : e[0], give = (g & 1) ? e[0] : e[1];
    float t = keep + __shfl_xor(give, 16, 64);
    t += __shfl_xor(t, 32, 64);                       // step 2 (partner g^2): both hold the same feature
    return g < 2 ? t : 0.f;
  } else {
    static_assert(NV == 4, "edge features: 2 or 4");
    // step 1 (partner g^1): keep the two features with the parity of g
    const bool odd = (g & 1) != 0;
    const float k0 = odd ? e[1] : e[0], k1 = odd ? e[3] : e[2];
    const float g0 = odd ? e[0] : e[1], g1 = odd ? e[2] : e[3];
    const float a = k0 + __shfl_xor(g0, 16, 64);      // feature (g&1)
    const float b = k1 + __shfl_xor(g1, 16, 64);      // feature (g&1) + 2
    // step 2 (partner g^2): keep feature g
    const bool hi = (g & 2) != 0;
    return (hi ? b : a) + __shfl_xor(hi ? a : b, 32, 64);
  }
}

}  // namespace vn16
