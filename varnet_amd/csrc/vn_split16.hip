// Point kernels of the 8-wave family on the bf16 matrix pipe with fp32-class accuracy (round 6): the model value
// (vn_forward) and the strong residual in second-order forward mode (vn_residual, the algorithm of vn_taylor16.hip:
// TFModel.py:543-545, 743-754) for the networks whose hidden layers are 33..64 wide (KS = 13, 16).
//
// Why.  On gfx950 the f32 MFMA shares its datapath with the f32 vector unit: vn_taylor16 / vn_pgrad16 sit at
// 0.60-0.62 of the f32 MFMA peak with the matrix pipe 0.63 busy, and nothing hides behind an f32 MFMA.  The bf16 MFMA
// (v_mfma_f32_16x16x32_bf16: 8 x the K per instruction at half the cycles) has its own pipe.  Every f32 operand is cut EXACTLY
// into three bf16 pieces x = h + m + l (3 x 8 significand bits, truncation), and a layer product is the six terms
// hh, hm, mh, hl, lh, mm with f32 accumulation, small terms first; the dropped terms (ml, lm, ll) are <= 3 * 2^-24 relative:
// an fp32-class product (profiles/r2_micro_split_bf16.md: one 50-wide layer against fp64 1.3e-6 where the f32 MFMA gives 1.4e-6,
// both dominated by v_exp_f32 / v_rcp_f32; the parity bars of vn_taylor16 / vn_pgrad16 are the bars of this file, unchanged).
//
// Layout.  The feature <-> accumulator-row mapping of the family (vn_fused16_common.h: feature f in k-step f/4, lane group f%4,
// accumulator row 16(ks>>2) + 4g + (ks&3)) is kept, so a layer's accumulator tiles ARE the next layer's B operand with no lane
// movement: B fragment q (k-steps 8q..8q+7) of a lane = its 8 values of row tiles 2q, 2q+1, packed pairwise.  Weight image of
// a hidden layer: 24 blocks [piece 3][q 2][row tile 4] of 1 KB = [g 4][c ^ 12(g&1)][8 bf16]: lane (g, c) reads the 8
// in-features 4(8q+j)+g, j = 0..7, of out-position 16 mt + c with ONE conflict-free ds_read_b128 (the micro study's layout).
// 24 KB per hidden layer: 96 KB at 5 x 50, one workgroup per CU (two waves per SIMD, 256 registers per lane).
// The input layer (d_in <= 8: K = 8) stays on v_mfma_f32_16x16x4_f32, the output layer on the vector unit, as in vn_taylor16.
#include "vn_points16.h"
#include "vn_split16.h"

#include <atomic>

namespace {
using namespace vn16;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4a __attribute__((may_alias));

constexpr int BLK = 1024;                                  // bytes of one (piece, q, row tile) block
constexpr int IMG = 24 * BLK;                              // bytes of one hidden layer's image

template <int L>
struct SLay {                                              // byte offsets
  static constexpr int WH_OFF = 0;                         // [L-1] images
  static constexpr int W1_OFF = (L - 1) * IMG;             // f32 [8][WS]: in-feature row, out-position column
  static constexpr int BI_OFF = W1_OFF + al4(8 * WS) * 4;  // f32 [L][64] biases in (tile, g, i) = position order
  static constexpr int WO_OFF = BI_OFF + L * 64 * 4;       // f32 [64] output weights by feature
  static constexpr int TOTAL = WO_OFF + 64 * 4;
};

struct VnSplitArgsD {
  VnNet net;
  const float* theta;
  const float* X;            // [n, d_in]
  const float* diff;         // [n]            (residual)
  const float* vel;          // [n, dim]
  const float* src;          // [n] or nullptr
  const float* ddx;          // [n, dim] or nullptr (grad kappa)
  int td;
  long n;
  float* u;                  // [n] or nullptr
  float* res;                // [n] (residual)
};

__device__ __forceinline__ u32 fu(float x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ float uf(u32 x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ u32 pack_hi(u32 u1, u32 u0) { return __builtin_amdgcn_perm(u1, u0, 0x07060302u); }   // (hi16(u1) << 16) | hi16(u0)

// exact three-way split of two f32 values into packed bf16 pairs (truncation: h = the top 8 significand bits, m the next 8 of
// the remainder, l the next 8).  Scalar subtracts: packed f32 instructions are expensive beside bf16 MFMAs (the study's raw table).
__device__ __forceinline__ void split2(float x0, float x1, u32& h, u32& m, u32& l) {
  const u32 u0 = fu(x0), u1 = fu(x1);
  h = pack_hi(u1, u0);
  const float r0 = x0 - uf(u0 & 0xffff0000u), r1 = x1 - uf(u1 & 0xffff0000u);
  const u32 v0 = fu(r0), v1 = fu(r1);
  m = pack_hi(v1, v0);
  const float s0 = r0 - uf(v0 & 0xffff0000u), s1 = r1 - uf(v1 & 0xffff0000u);
  l = pack_hi(fu(s1), fu(s0));
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the six products of one (row tile, K fragment), small terms first; the streams that share the weight fragment are interleaved
// product by product (independent accumulators: no MFMA waits for its predecessor's result)
template <int N>
__device__ __forceinline__ void six(const u32x4 (&A)[3], const u32x4 (*const (&B)[N])[3], f32x4* const (&acc)[N]) {
  constexpr int pa[6] = {1, 2, 0, 1, 0, 0}, pb[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int s = 0; s < N; ++s) *acc[s] = mfma_bf16(A[pa[i]], (*B[s])[pb[i]], *acc[s]);
  }
}

// Prologue of both kernels: the bf16-piece images of the hidden layers -- one 16-byte entry (8 in-features of one out-position)
// per thread and layer, cut into its three pieces here -- and the f32 input-layer image, biases and output weights; padding exact
// zeros.  Ends with the images written but not yet visible to other waves: the caller synchronises.
template <int L>
__device__ __forceinline__ void stage_split_images(const VnNet& net, const float* theta, char* ldsb, int tid) {
  using LY = SLay<L>;
  float* W1 = reinterpret_cast<float*>(ldsb + LY::W1_OFF);
  float* BI = reinterpret_cast<float*>(ldsb + LY::BI_OFF);
  float* WO = reinterpret_cast<float*>(ldsb + LY::WO_OFF);
  const int q = tid >> 8, mt = (tid >> 6) & 3, g = (tid >> 4) & 3, c = tid & 15;
  const int pos = 16 * mt + c, fo = vfeat(pos);
  const int ent = (g * 16 + (c ^ (12 * (g & 1)))) * 16;
#pragma unroll
  for (int l = 2; l <= L; ++l) {
    const int Hin = net.H[l - 1], Hout = net.H[l];
    const float* src = theta + net.woff[l];
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int fi = 4 * (8 * q + j) + g;
      w[j] = (fi < Hin && fo < Hout) ? src[fi * Hout + fo] : 0.f;
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      u32 h, m, lo;
      split2(w[2 * jj], w[2 * jj + 1], h, m, lo);
      ph[jj] = h; pm[jj] = m; pl[jj] = lo;
    }
    char* img = ldsb + LY::WH_OFF + (l - 2) * IMG;
    *reinterpret_cast<u32x4a*>(img + ((0 * 2 + q) * 4 + mt) * BLK + ent) = ph;
    *reinterpret_cast<u32x4a*>(img + ((1 * 2 + q) * 4 + mt) * BLK + ent) = pm;
    *reinterpret_cast<u32x4a*>(img + ((2 * 2 + q) * 4 + mt) * BLK + ent) = pl;
  }
  for (int i = tid; i < al4(8 * WS); i += NTHREADS) W1[i] = 0.f;
  __syncthreads();
  const int H1 = net.H[1];
  if (tid < net.d_in * H1) {
    const int k = tid / H1, f = tid - k * H1;
    W1[k * WS + vpos(f >> 2, f & 3)] = theta[net.woff[1] + tid];
  }
  static_assert(L * 64 <= NTHREADS, "one bias per thread");
  if (tid < L * 64) {
    const int l = tid / 64 + 1, idx = tid % 64;
    const int bm = idx >> 4, bg = (idx >> 2) & 3, br = idx & 3;       // [tile][g][i]
    const int ks = 4 * bm + br, f = 4 * ks + bg;
    BI[tid] = (f < net.H[l]) ? theta[net.boff[l] + f] : 0.f;
  }
  if (tid < 64) WO[tid] = (tid < net.H[L]) ? theta[net.woff[L + 1] + tid] : 0.f;
}

// NS = 1: value only (vn_forward); NS = 3: value, first and second directional derivative per coordinate (vn_residual)
template <int L, int KS, bool TANH, int NS>
__global__ __launch_bounds__(NTHREADS, 1) void vn_split16_kernel(VnSplitArgsD A) {
  static_assert(KS == 13 || KS == 16, "two K fragments of 32: hidden widths 33..64");
  static_assert(NS == 1 || NS == 3, "value, or the three Taylor streams");
  using LY = SLay<L>;
  constexpr int MT = 4;
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const VnNet& net = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* W1 = reinterpret_cast<float*>(ldsb + LY::W1_OFF);
  float* BI = reinterpret_cast<float*>(ldsb + LY::BI_OFF);
  float* WO = reinterpret_cast<float*>(ldsb + LY::WO_OFF);
  const float bo = A.theta[net.boff[L + 1]];

  stage_split_images<L>(net, A.theta, ldsb, tid);
  __syncthreads();

  const int g = lane >> 4, c = lane & 15;
  const int offF = g * WS + c;                       // input layer A fragment: in-feature 4s+g, out-position 16m+c
  const char* rd = ldsb + LY::WH_OFF + (g * 16 + (c ^ (12 * (g & 1)))) * 16;
  const int dim = net.dim, nd1 = (NS == 1) ? 1 : dim;
  // Streams of a pass (NS = 3): value, first and second derivative along e_d -- and, in pass 0 of a time-dependent problem, the
  // first derivative along t as a FOURTH stream (it needs no second derivative; a pass of its own would recompute the value
  // stream and every activation for it: 3 dim + 1 streams per point instead of vn_taylor16's 3 dim + 2).

  const long nchunks = (A.n + CW - 1) / CW;
  for (long chunk = (long)blockIdx.x * NW + wave; chunk < nchunks; chunk += (long)gridDim.x * NW) {
    const long row = chunk * CW + c;
    const bool valid = row < A.n;
    float xin[KS0];
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int f = 4 * s + g;
      xin[s] = (valid && f < net.d_in) ? A.X[row * net.d_in + f] : 0.f;
    }
    float uval = 0.f, lap = 0.f, adv = 0.f, ut = 0.f;
#pragma unroll 1
    for (int d = 0; d < nd1; ++d) {                  // one pass per spatial direction e_d
      asm volatile("" ::: "memory");                 // keep LDS fragment loads inside the loop
      const bool tstr = NS == 3 && A.td && d == 0;   // wave-uniform: this pass carries the time tangent as well
      // ---------------------------------------------------------------- input layer (f32 MFMA; z.. = 0)
      f32x4 pv[MT], p1[MT], p2[MT], pt[MT];          // pre-activations: value, d/de_d, d2/de_d^2, d/dt
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        pv[m] = *reinterpret_cast<const f32x4a*>(&BI[m * 16 + g * 4]);
        p1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        p2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        pt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        if (4 * s < net.d_in) {
          const float gin = (4 * s + g == d) ? 1.f : 0.f, gt = (4 * s + g == dim) ? 1.f : 0.f;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float wf = W1[4 * s * WS + offF + 16 * m];
            pv[m] = mfma16(wf, xin[s], pv[m]);
            if (NS == 3) {
              p1[m] = mfma16(wf, gin, p1[m]);
              if (tstr) pt[m] = mfma16(wf, gt, pt[m]);
            }
          }
        }
      }
      // ---------------------------------------------------------------- hidden layers (bf16 pieces), K fragment by K fragment
#pragma unroll
      for (int l = 2; l <= L; ++l) {
        const char* rl = rd + (l - 2) * IMG;
        f32x4 nv[MT], n1[MT], n2[MT], nt[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          nv[mt] = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + mt * 16 + g * 4]);
          n1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          n2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          nt[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          // B fragments of k-steps 8q..8q+7: activation of the previous layer, the derivative streams, each cut into three pieces
          u32x4 Bv[3], B1[3], B2[3], Bt[3];
#pragma unroll
          for (int e = 0; e < 4; ++e) {              // pair e = k-steps 8q+2e, 8q+2e+1
            const int k0 = 8 * q + 2 * e;
            if (k0 >= KS) {                          // padding k-steps (KS = 13: 13, 14, 15 -- their weights are zeros as well)
#pragma unroll
              for (int p = 0; p < 3; ++p) { Bv[p][e] = 0u; B1[p][e] = 0u; B2[p][e] = 0u; Bt[p][e] = 0u; }
              continue;
            }
            const int t = k0 >> 2, i = k0 & 3;
            const bool full = k0 + 1 < KS;
            const float a0 = act_fin<TANH>(act_exp<TANH>(pv[t][i]));
            const float a1 = full ? act_fin<TANH>(act_exp<TANH>(pv[t][i + 1])) : 0.f;
            u32 h, m, lo;
            split2(a0, a1, h, m, lo);
            Bv[0][e] = h; Bv[1][e] = m; Bv[2][e] = lo;
            if (NS == 3) {
              const float s0 = act_d1<TANH>(a0), s1 = full ? act_d1<TANH>(a1) : 0.f;
              const float zd0 = p1[t][i], zd1 = p1[t][i + 1];
              split2(s0 * zd0, s1 * zd1, h, m, lo);
              B1[0][e] = h; B1[1][e] = m; B1[2][e] = lo;
              // a..' = sigma'(z) ((sigma''/sigma')(z) z.^2 + z..)
              const float w0 = s0 * __builtin_fmaf(act_d2r<TANH>(a0) * zd0, zd0, p2[t][i]);
              const float w1 = s1 * __builtin_fmaf(act_d2r<TANH>(a1) * zd1, zd1, p2[t][i + 1]);
              split2(w0, w1, h, m, lo);
              B2[0][e] = h; B2[1][e] = m; B2[2][e] = lo;
              if (tstr) {
                split2(s0 * pt[t][i], s1 * pt[t][i + 1], h, m, lo);
                Bt[0][e] = h; Bt[1][e] = m; Bt[2][e] = lo;
              }
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            u32x4 Af[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) Af[p] = *reinterpret_cast<const u32x4a*>(rl + ((p * 2 + q) * 4 + mt) * BLK);
            if (NS == 1) {
              const u32x4 (*const Bs[1])[3] = {&Bv};
              f32x4* const as[1] = {&nv[mt]};
              six<1>(Af, Bs, as);
            } else if (tstr) {
              const u32x4 (*const Bs[4])[3] = {&Bv, &B1, &B2, &Bt};
              f32x4* const as[4] = {&nv[mt], &n1[mt], &n2[mt], &nt[mt]};
              six<4>(Af, Bs, as);
            } else {
              const u32x4 (*const Bs[3])[3] = {&Bv, &B1, &B2};
              f32x4* const as[3] = {&nv[mt], &n1[mt], &n2[mt]};
              six<3>(Af, Bs, as);
            }
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) { pv[m] = nv[m]; p1[m] = n1[m]; p2[m] = n2[m]; pt[m] = nt[m]; }
      }
      // ---------------------------------------------------------------- last activation + output layer (vector unit)
      float us = 0.f, u1s = 0.f, u2s = 0.f, uts = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float wv = WO[4 * ks + g];
        const float av = act_fin<TANH>(act_exp<TANH>(pv[ks >> 2][ks & 3]));
        us = __builtin_fmaf(wv, av, us);
        if (NS == 3) {
          const float zd = p1[ks >> 2][ks & 3];
          const float sp = wv * act_d1<TANH>(av);
          u1s = __builtin_fmaf(sp, zd, u1s);
          u2s = __builtin_fmaf(sp, __builtin_fmaf(act_d2r<TANH>(av) * zd, zd, p2[ks >> 2][ks & 3]), u2s);
          if (tstr) uts = __builtin_fmaf(sp, pt[ks >> 2][ks & 3], uts);
        }
      }
      uval = rowsum4(us) + bo;
      if (NS == 3) {
        const float ud = rowsum4(u1s);
        lap += rowsum4(u2s);                         // TFModel.py:750-754
        if (tstr) ut = rowsum4(uts);
        float vd = 0.f;
        if (valid) {
          vd = A.vel[row * dim + d];
          if (A.ddx) vd -= A.ddx[row * dim + d];
        }
        adv += vd * ud;
      }
    }
    if (valid && g == 0) {
      if (NS == 1) {
        A.u[row] = uval;
      } else {
        float out = A.td ? -ut : 0.f;
        out += A.diff[row] * lap;
        out -= adv;
        if (A.src) out += A.src[row];
        if (A.u) A.u[row] = uval;
        A.res[row] = out;
      }
    }
  }
}

// Value and input gradient in ONE pass (the algorithm of vn_pgrad16.hip: value forward, value-adjoint sweep back to the inputs
// with seed 1 -- tf.gradients(model(Input), Input), TFModel.py:536-541 -- 2 F_pt per point), both sweeps on the bf16 pipe.  The
// sweep back contracts over a layer's OUT-features, i.e. needs the transposed weight fragments: they come from the SAME images
// through ds_read_b64_tr_b16 (per 16-lane group a 4 x 16 block of 16-bit elements delivered column-major; the XOR of the entry
// layout keeps it 2-way): lane (g, c) wants W[in-feature of position 16 mt + c][out-feature 4(8q+j)+g], j = 0..7 -- in the forward
// image those are element 4(mt&1) + (c&3) of the entries (g_in = (c>>2)&3, c_out = 4g + (j&3)) of blocks (q_in = mt>>1,
// mt_out = 2q + (j>>2)): two transposed reads per piece.
struct VnSplitPgArgsD {
  VnNet net;
  const float* theta;
  const float* X;            // [n, d_in]
  long n;
  float* out_u;              // [n] or nullptr
  float* out_g;              // [n, dim] or nullptr
  float* out_pack;           // [n, 4] = (u, du/dx_0, du/dx_1, du/dx_2) or nullptr
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int L, int KS, bool TANH>
__global__ __launch_bounds__(NTHREADS, 1) void vn_split16_pgrad_kernel(VnSplitPgArgsD A) {
  static_assert(KS == 13 || KS == 16, "two K fragments of 32: hidden widths 33..64");
  using LY = SLay<L>;
  constexpr int MT = 4;
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const VnNet& net = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* W1 = reinterpret_cast<float*>(ldsb + LY::W1_OFF);
  float* BI = reinterpret_cast<float*>(ldsb + LY::BI_OFF);
  float* WO = reinterpret_cast<float*>(ldsb + LY::WO_OFF);
  const float bo = A.theta[net.boff[L + 1]];
  stage_split_images<L>(net, A.theta, ldsb, tid);
  __syncthreads();

  const int g = lane >> 4, c = lane & 15;
  const int offF = g * WS + c;
  const char* rd = ldsb + LY::WH_OFF + (g * 16 + (c ^ (12 * (g & 1)))) * 16;                    // row read (forward)
  // transposed read: lane 4r + p of its group supplies row r (c_out = 4g + r), columns 4p..4p+3 (entry g_in = p)
  const int tr_r = c >> 2, tr_p = c & 3;
  const char* rt = ldsb + LY::WH_OFF + (tr_p * 16 + ((4 * g + tr_r) ^ (12 * (tr_p & 1)))) * 16;

  const long nchunks = (A.n + CW - 1) / CW;
  for (long chunk = (long)blockIdx.x * NW + wave; chunk < nchunks; chunk += (long)gridDim.x * NW) {
    asm volatile("" ::: "memory");                   // keep LDS fragment loads inside the loop
    const long row = chunk * CW + c;
    const bool valid = row < A.n;
    float xin[KS0];
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int f = 4 * s + g;
      xin[s] = (valid && f < net.d_in) ? A.X[row * net.d_in + f] : 0.f;
    }
    float a[L][KS];                                  // activations of every layer, kept for the sweep back
    // ---------------------------------------------------------------- value forward
    f32x4 pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pv[m] = *reinterpret_cast<const f32x4a*>(&BI[m * 16 + g * 4]);
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      if (4 * s < net.d_in) {
#pragma unroll
        for (int m = 0; m < MT; ++m) pv[m] = mfma16(W1[4 * s * WS + offF + 16 * m], xin[s], pv[m]);
      }
    }
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const char* rl = rd + (l - 2) * IMG;
      f32x4 nv[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) nv[mt] = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + mt * 16 + g * 4]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        u32x4 Bv[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k0 = 8 * q + 2 * e;
          if (k0 >= KS) {
#pragma unroll
            for (int p = 0; p < 3; ++p) Bv[p][e] = 0u;
            continue;
          }
          const bool full = k0 + 1 < KS;
          const float a0 = act_fin<TANH>(act_exp<TANH>(pv[k0 >> 2][k0 & 3]));
          const float a1 = full ? act_fin<TANH>(act_exp<TANH>(pv[k0 >> 2][(k0 & 3) + 1])) : 0.f;
          a[l - 2][k0] = a0;
          if (full) a[l - 2][k0 + 1] = a1;
          u32 h, m, lo;
          split2(a0, a1, h, m, lo);
          Bv[0][e] = h; Bv[1][e] = m; Bv[2][e] = lo;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          u32x4 Af[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) Af[p] = *reinterpret_cast<const u32x4a*>(rl + ((p * 2 + q) * 4 + mt) * BLK);
          const u32x4 (*const Bs[1])[3] = {&Bv};
          f32x4* const as[1] = {&nv[mt]};
          six<1>(Af, Bs, as);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) pv[m] = nv[m];
    }
    // ---------------------------------------------------------------- last activation, output layer and its adjoint (seed 1)
    float zb[KS];
    float us = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float wv = WO[4 * ks + g];
      const float av = act_fin<TANH>(act_exp<TANH>(pv[ks >> 2][ks & 3]));
      a[L - 1][ks] = av;
      us = __builtin_fmaf(wv, av, us);
      zb[ks] = wv * act_d1<TANH>(av);                // d u / d z_L
    }
    const float u = rowsum4(us) + bo;
    // ---------------------------------------------------------------- value-adjoint sweep to the inputs
#pragma unroll
    for (int l = L; l >= 2; --l) {
      const char* tl = rt + (l - 2) * IMG;
      f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 2; ++q) {                  // K fragment over the OUT-features 4(8q+j)+g of layer l
        u32x4 Bz[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k0 = 8 * q + 2 * e;
          if (k0 >= KS) {
#pragma unroll
            for (int p = 0; p < 3; ++p) Bz[p][e] = 0u;
            continue;
          }
          u32 h, m, lo;
          split2(zb[k0], (k0 + 1 < KS) ? zb[k0 + 1] : 0.f, h, m, lo);
          Bz[0][e] = h; Bz[1][e] = m; Bz[2][e] = lo;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {            // row tile over the IN-positions 16 mt + c
          u32x4 At[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const char* b0 = tl + ((p * 2 + (mt >> 1)) * 4 + 2 * q) * BLK + 8 * (mt & 1);
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0 + BLK));
            const unsigned long long l64 = __builtin_bit_cast(unsigned long long, lo4), h64 = __builtin_bit_cast(unsigned long long, hi4);
            At[p] = u32x4{(u32)l64, (u32)(l64 >> 32), (u32)h64, (u32)(h64 >> 32)};
          }
          const u32x4 (*const Bs[1])[3] = {&Bz};
          f32x4* const as[1] = {&acc[mt]};
          six<1>(At, Bs, as);
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) zb[ks] = acc[ks >> 2][ks & 3] * act_d1<TANH>(a[l - 2][ks]);
    }
    // input layer: du/dx_d = sum_f W_1[d][f] zbar_1[f]; this lane holds feature 4ks+g of every k-step, the four lane groups are
    // summed by row swaps (same pairing in every lane: all agree bit for bit)
    float xg[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d < net.dim) {
        float t = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) t = __builtin_fmaf(W1[d * WS + vpos(ks, 0) + 4 * g], zb[ks], t);
        xg[d] = rowsum4(t);
      }
    }
    if (valid) {
      const float mine = (g == 3) ? u : (g == 0) ? xg[0] : (g == 1) ? xg[1] : xg[2];       // xg of an absent coordinate is 0
      if (A.out_pack) A.out_pack[row * 4 + ((g + 1) & 3)] = mine;                          // (u, g0, g1, g2): lane group 3 holds u
      if (g == 3) { if (A.out_u) A.out_u[row] = u; }
      else if (g < net.dim && A.out_g) A.out_g[row * net.dim + g] = mine;
    }
  }
}

template <int L, int KS, bool TANH, int NS>
hipError_t launch_one(const VnSplitArgsD& a, int ncu, hipStream_t s) {
  constexpr size_t bytes = (size_t)SLay<L>::TOTAL;
  static_assert(bytes <= 160 * 1024, "images of L - 1 hidden layers must fit the LDS");
  static std::atomic<unsigned long long> attr_done{0};   // per device and sticky: set once per device (vn_pgrad16.hip)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_split16_kernel<L, KS, TANH, NS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const long wgs = ((a.n + CW - 1) / CW + NW - 1) / NW;
  const int grid = (int)(wgs < ncu ? wgs : ncu);
  hipLaunchKernelGGL((vn_split16_kernel<L, KS, TANH, NS>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

template <int L, int KS, bool TANH>
hipError_t launch_pg(const VnSplitPgArgsD& a, int ncu, hipStream_t s) {
  constexpr size_t bytes = (size_t)SLay<L>::TOTAL;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_split16_pgrad_kernel<L, KS, TANH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const long wgs = ((a.n + CW - 1) / CW + NW - 1) / NW;
  const int grid = (int)(wgs < ncu ? wgs : ncu);
  hipLaunchKernelGGL((vn_split16_pgrad_kernel<L, KS, TANH>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

}  // namespace

// hidden widths 33..64 of the 8-wave family, 2..7 hidden layers (L = 1 has no hidden product; 8 x 24 KB of images do not fit)
#define VN_SPLIT16_CASES(X) VN_POINT16_SPLIT_CASES(X)

bool vn_split16_supported(const VnNet& net) {
  if (net.d_in > 4 * KS0) return false;
  if (net.act != VN_ACT_SIGMOID && net.act != VN_ACT_TANH) return false;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK) if (net.L == LL && ks == KK) return true;
  VN_SPLIT16_CASES(X)
#undef X
  return false;
}

hipError_t vn_split16_forward(const VnNet& net, const float* theta, const float* X, long n, float* u, int ncu, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (!vn_split16_supported(net) || !u) return hipErrorInvalidValue;
  VnSplitArgsD a{};
  a.net = net; a.theta = theta; a.X = X; a.n = n; a.u = u;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK)                                                                              \
  if (net.L == LL && ks == KK)                                                                  \
    return net.act == VN_ACT_TANH ? launch_one<LL, KK, true, 1>(a, ncu, s) : launch_one<LL, KK, false, 1>(a, ncu, s);
  VN_SPLIT16_CASES(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t vn_split16_residual(const VnNet& net, const float* theta, const float* X, const float* diff, const float* vel,
                               const float* src, const float* ddx, int td, long n, float* u, float* res, int ncu, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (!vn_split16_supported(net) || net.dim > 3 || net.dim + (td ? 1 : 0) > net.d_in) return hipErrorInvalidValue;
  VnSplitArgsD a{};
  a.net = net; a.theta = theta; a.X = X; a.diff = diff; a.vel = vel; a.src = src; a.ddx = ddx; a.td = td; a.n = n; a.u = u; a.res = res;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK)                                                                              \
  if (net.L == LL && ks == KK)                                                                  \
    return net.act == VN_ACT_TANH ? launch_one<LL, KK, true, 3>(a, ncu, s) : launch_one<LL, KK, false, 3>(a, ncu, s);
  VN_SPLIT16_CASES(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t vn_split16_pgrad(const VnNet& net, const float* theta, const float* X, long n, float* out_u, float* out_g, float* out_pack,
                            int ncu, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (!vn_split16_supported(net) || net.dim > 3) return hipErrorInvalidValue;
  VnSplitPgArgsD a{};
  a.net = net; a.theta = theta; a.X = X; a.n = n; a.out_u = out_u; a.out_g = out_g; a.out_pack = out_pack;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK)                                                                              \
  if (net.L == LL && ks == KK)                                                                  \
    return net.act == VN_ACT_TANH ? launch_pg<LL, KK, true>(a, ncu, s) : launch_pg<LL, KK, false>(a, ncu, s);
  VN_SPLIT16_CASES(X)
#undef X
  return hipErrorInvalidValue;
}
