// GEMMs of the layer-by-layer route (vn_layered.hip), hand-written for gfx950: the products a layer needs when its width is
// beyond the tile kernels of vn_wide.hip (hidden widths above 256; TFModel.py:208-221 accepts any list).
//
//   forward          (a | ad) [S][c][N] = epilogue(A [S][c][K] W [K x N] + b)   vn_gemm_fwd   (K = H_in, N = H_out; a = act(z + b),
//                                                                                ad = act'(z) zd fused into the product)
//   input gradient   dA [M x N]  = Zb [M x K] (W^T) [K x N]                      vn_transpose + vn_gemm_nn   (K = H_out, N = H_in)
//   weight gradient  P_g [K1 x N] = A_g^T Zb_g over the rows of group g          vn_gemm_tn_parts   (a fixed-order sum adds the groups)
//   fp64 forward     C [M x N]   = A [M x K] W [K x N]                           vn_dgemm_nn   (fp64 entry points)
//
// All matrices row-major.  M is millions of rows, K / N a few hundred: the shapes are tall and skinny and compute-bound on the
// fp32 matrix pipe (43 FLOP per byte of operand traffic per 128 x 256 x 16 step, 32 per 128 x 128 x 16 step).
//
// Geometry: workgroup = 2 x WC waves, C tile 128 x 64 WC (WC = 4: 128 x 256, 512 threads; WC = 2: 128 x 128, 256 threads --
// whichever pads N less), K step 16, v_mfma_f32_16x16x4_f32; wave (wm, wn) owns a 64 x 64 quadrant = 4 x 4 accumulator tiles
// (64 registers), 128 registers per wave, 16 waves per CU.  Operand tiles are double-buffered in LDS, one workgroup barrier per
// K step; the global loads of step k+1 are issued before the MFMAs of step k and written to LDS after them.  Quadrant tiles
// that lie outside the matrix are skipped (whole 16 x 16 tiles), and the wave -> quadrant map rotates with the workgroup so
// that the idle quadrants of edge tiles do not always idle the same SIMD.
// Two LDS tile layouts, chosen so that the global side is always 16-byte loads along the contiguous dimension:
//   * [row][k]   (row stride 20 floats): operands whose K runs along memory (the A side of nn / fwd).  A lane's fragment of
//     one 16-row tile for all four MFMAs of a K step is ONE ds_read_b128: MFMA step s of lane group lk uses k = 4 lk + s --
//     any assignment of the 16 k's to (step, lane group) is valid as long as both operands use the same one.
//   * [k][col]   (row stride 64 WC + 4 floats): operands whose K runs across rows (W of nn / fwd; both operands of tn, whose K is
//     the row index of the chunk).  Fragment = four ds_read_b32 (conflict-free: 16 consecutive columns per lane group, lane
//     groups 4 rows = 16 banks apart).
// The input layer (K = d_in <= 32) is served by streaming kernels in both directions: an MFMA tile would be 97 % padding.
// No atomics anywhere; every sum has a fixed order, so results are bitwise reproducible.
// Dimensions that are not multiples of 4, or operands that are not 16-byte aligned, take the same kernels with element-wise
// guarded loads (VEC = false): correct for any width the reference accepts, slower.
#include "vn_internal.h"

#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BK = 16;
constexpr int LDK = 20;        // [row][k] tiles: 16 k's + 4 (16-byte aligned rows, b128 fragment reads 4 banks apart)
constexpr int ROWK_SZ = BM * LDK;
// WC = wave columns of a workgroup: 2 x WC waves, C tile 128 x 64 WC (WC = 4: 128 x 256, 512 threads -- a third less
// operand traffic per FLOP than 128 x 128, which is what the L2 -> CU path of these kernels is short of)
template <int WC>
struct Geo {
  static constexpr int BN = 64 * WC;
  static constexpr int NTHR = 128 * WC;
  static constexpr int LDN = BN + 4;            // [k][col] tiles of the N side
  static constexpr int KCOLN_SZ = BK * LDN;
};
constexpr int LDM = BM + 4;                    // [k][col] tiles of a 128-wide side (tn: the K1 side)
constexpr int KCOLM_SZ = BK * LDM;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int E> struct Stage { f32x4 v[E]; };

// 128 rows x 16 k's of a row-major matrix with leading dimension ld, as 512 16-byte pieces over NTHR threads
template <bool VEC, int NTHR>
__device__ __forceinline__ Stage<512 / NTHR> load_rowk(const float* __restrict__ P, long ld, long row0, long nrows, int k0, int K, int t) {
  Stage<512 / NTHR> s;
#pragma unroll
  for (int e = 0; e < 512 / NTHR; ++e) {
    const int idx = t + e * NTHR;
    const long row = row0 + (idx >> 2);
    const int kq = k0 + (idx & 3) * 4;
    const float* p = P + row * ld + kq;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) {
      if constexpr (VEC) {
        if (kq < K) v = *(const f32x4*)p;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (kq + c < K) v[c] = p[c];
      }
    }
    s.v[e] = v;
  }
  return s;
}
template <int NTHR>
__device__ __forceinline__ void store_rowk(float* T, const Stage<512 / NTHR>& s, int t) {
#pragma unroll
  for (int e = 0; e < 512 / NTHR; ++e) {
    const int idx = t + e * NTHR;
    *(f32x4*)(T + (idx >> 2) * LDK + (idx & 3) * 4) = s.v[e];
  }
}

// 16 k-rows x COLS columns of a row-major matrix with leading dimension ld, as 4 COLS 16-byte pieces over NTHR threads
template <bool VEC, int NTHR, int COLS>
__device__ __forceinline__ Stage<4 * COLS / NTHR> load_kcol(const float* __restrict__ P, long ld, long krow0, long nkrows, int c0, int C,
                                                            int t) {
  constexpr int E = 4 * COLS / NTHR, PR = COLS / 4;      // pieces per row
  Stage<E> s;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = t + e * NTHR;
    const long kr = krow0 + idx / PR;
    const int cq = c0 + (idx % PR) * 4;
    const float* p = P + kr * ld + cq;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kr < nkrows) {
      if constexpr (VEC) {
        if (cq < C) v = *(const f32x4*)p;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (cq + c < C) v[c] = p[c];
      }
    }
    s.v[e] = v;
  }
  return s;
}
template <int NTHR, int COLS>
__device__ __forceinline__ void store_kcol(float* T, const Stage<4 * COLS / NTHR>& s, int t) {
  constexpr int E = 4 * COLS / NTHR, PR = COLS / 4;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = t + e * NTHR;
    *(f32x4*)(T + (idx / PR) * (COLS + 4) + (idx % PR) * 4) = s.v[e];
  }
}

// fragments of the four 16-row (or 16-column) tiles of a wave's 64-wide quadrant
__device__ __forceinline__ void frag_rowk(const float* T, int q0, int lm, int lk, f32x4 (&f)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = *(const f32x4*)(T + (q0 + 16 * i + lm) * LDK + 4 * lk);
}
template <int LD>
__device__ __forceinline__ void frag_kcol(const float* T, int q0, int lm, int lk, f32x4 (&f)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int s = 0; s < 4; ++s) f[i][s] = T[(4 * lk + s) * LD + q0 + 16 * i + lm];
  }
}
__device__ __forceinline__ void mma_step(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i][s], b[j][s], acc[i][j]);
    }
  }
}
// edge quadrants: only the 16 x 16 tiles that reach into the matrix (wave-uniform bounds; whole tiles, so the k order of every
// accumulator stays the same and a quadrant entirely outside costs nothing)
__device__ __forceinline__ void mma_step_edge(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x4 (&acc)[4][4], int in, int jn) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < in) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < jn) {
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[i][j] = mfma16(a[i][s], b[j][s], acc[i][j]);
        }
      }
    }
  }
}
// number of 16-wide tiles of a 64-wide quadrant starting at q0 that reach below `n`
__device__ __forceinline__ int live_tiles(long q0, long n) {
  const long v = (n - q0 + 15) / 16;
  return v < 0 ? 0 : v > 4 ? 4 : (int)v;
}
// accumulator tile (i, j), register e: row 64 wm + 16 i + 4 lk + e, column 64 wn + 16 j + lm
__device__ __forceinline__ void store_c(float* __restrict__ C, long ldc, long row0, long nrows, int col0, int ncols, int wm, int wn,
                                        int lm, int lk, const f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long row = row0 + 64 * wm + 16 * i + 4 * lk + e;
      if (row < nrows) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = col0 + 64 * wn + 16 * j + lm;
          if (col < ncols) C[row * ldc + col] = acc[i][j][e];
        }
      }
    }
  }
}

// The K loop of one workgroup, written once and instantiated twice per kernel: EDGE = false for quadrants that lie fully
// inside the matrix (straight-line 64 MFMAs per step), EDGE = true for the others.  The kernel branches ONCE, outside the loop:
// with both bodies in one loop the compiler kept both sets of live ranges and spilled 130+ registers at the 128 cap.
template <int WC, bool VEC, bool EDGE>
__device__ __forceinline__ void gemm_loop(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, long M, int N,
                                          int K, long m0, int n0, int wm, int wn, int in, int jn, float (*sA)[ROWK_SZ],
                                          float (*sB)[Geo<WC>::KCOLN_SZ], int t, int lm, int lk) {
  using G = Geo<WC>;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nk = (K + BK - 1) / BK;
  auto ga = load_rowk<VEC, G::NTHR>(A, K, m0, M, 0, K, t);
  auto gb = load_kcol<VEC, G::NTHR, G::BN>(B, N, 0, K, n0, N, t);
  store_rowk<G::NTHR>(sA[0], ga, t);
  store_kcol<G::NTHR, G::BN>(sB[0], gb, t);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const int k1 = (kt + 1) * BK;
      ga = load_rowk<VEC, G::NTHR>(A, K, m0, M, k1, K, t);
      gb = load_kcol<VEC, G::NTHR, G::BN>(B, N, k1, K, n0, N, t);
    }
    if (!EDGE || (in > 0 && jn > 0)) {
      f32x4 fa[4], fb[4];
      frag_rowk(sA[cur], 64 * wm, lm, lk, fa);
      frag_kcol<G::LDN>(sB[cur], 64 * wn, lm, lk, fb);
      if (EDGE) mma_step_edge(fa, fb, acc, in, jn); else mma_step(fa, fb, acc);
    }
    if (kt + 1 < nk) {
      store_rowk<G::NTHR>(sA[cur ^ 1], ga, t);
      store_kcol<G::NTHR, G::BN>(sB[cur ^ 1], gb, t);
    }
    __syncthreads();
  }
  store_c(C, N, m0, M, n0, N, wm, wn, lm, lk, acc);
}

// C[M x N] = A[M x K] W[K x N].  Consecutive workgroups share a row block of A (they differ in the column block), so the big
// operand is read from HBM once and from L2 afterwards.
template <int WC, bool VEC>
__global__ __launch_bounds__(128 * WC, 4) void vn_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ C, long M, int N, int K, int ntn) {
  using G = Geo<WC>;
  __shared__ __attribute__((aligned(16))) float sA[2][ROWK_SZ];
  __shared__ __attribute__((aligned(16))) float sB[2][G::KCOLN_SZ];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), lm = lane & 15, lk = lane >> 4;
  const long bid = blockIdx.x;
  // wave -> quadrant rotates with the workgroup: the quadrants of an edge tile that lie outside the matrix are skipped, and
  // wave w of every workgroup sits on SIMD w mod 4 -- without the rotation the idle quadrants would always idle the same SIMD
  const int quad = (wave + (int)(bid % (2 * WC))) % (2 * WC);
  const int wm = quad / WC, wn = quad % WC;
  const long m0 = (bid / ntn) * BM;
  const int n0 = (int)(bid % ntn) * G::BN;
  const int in = live_tiles(m0 + 64 * wm, M), jn = live_tiles(n0 + 64 * wn, N);
  if (in == 4 && jn == 4) gemm_loop<WC, VEC, false>(A, B, C, M, N, K, m0, n0, wm, wn, in, jn, sA, sB, t, lm, lk);
  else gemm_loop<WC, VEC, true>(A, B, C, M, N, K, m0, n0, wm, wn, in, jn, sA, sB, t, lm, lk);
}

template <int WC, bool VEC, bool EDGE>
__device__ __forceinline__ void gemm_tn_loop(const float* __restrict__ A, const float* __restrict__ Z, float* __restrict__ out, int K1,
                                             int N, long r0, long r1, int k0, int n0, int wm, int wn, int in, int jn,
                                             float (*sA)[KCOLM_SZ], float (*sZ)[Geo<WC>::KCOLN_SZ], int t, int lm, int lk) {
  using G = Geo<WC>;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const long nk = (r1 - r0 + BK - 1) / BK;
  auto ga = load_kcol<VEC, G::NTHR, BM>(A, K1, r0, r1, k0, K1, t);
  auto gz = load_kcol<VEC, G::NTHR, G::BN>(Z, N, r0, r1, n0, N, t);
  store_kcol<G::NTHR, BM>(sA[0], ga, t);
  store_kcol<G::NTHR, G::BN>(sZ[0], gz, t);
  __syncthreads();
  for (long kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nk) {
      ga = load_kcol<VEC, G::NTHR, BM>(A, K1, r0 + (kt + 1) * BK, r1, k0, K1, t);
      gz = load_kcol<VEC, G::NTHR, G::BN>(Z, N, r0 + (kt + 1) * BK, r1, n0, N, t);
    }
    if (!EDGE || (in > 0 && jn > 0)) {
      f32x4 fa[4], fb[4];
      frag_kcol<LDM>(sA[cur], 64 * wm, lm, lk, fa);
      frag_kcol<G::LDN>(sZ[cur], 64 * wn, lm, lk, fb);
      if (EDGE) mma_step_edge(fa, fb, acc, in, jn); else mma_step(fa, fb, acc);
    }
    if (kt + 1 < nk) {
      store_kcol<G::NTHR, BM>(sA[cur ^ 1], ga, t);
      store_kcol<G::NTHR, G::BN>(sZ[cur ^ 1], gz, t);
    }
    __syncthreads();
  }
  store_c(out, N, k0, K1, n0, N, wm, wn, lm, lk, acc);
}

// parts[g][K1 x N] = sum over the rows r of group g of A[r][:]^T Z[r][:]   (group g = rows g * rows .. min(M, (g+1) * rows))
template <int WC, bool VEC>
__global__ __launch_bounds__(128 * WC, 4) void vn_gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ Z,
                                                                 float* __restrict__ parts, long M, int K1, int N, long rows, int ntn) {
  using G = Geo<WC>;
  __shared__ __attribute__((aligned(16))) float sA[2][KCOLM_SZ];
  __shared__ __attribute__((aligned(16))) float sZ[2][G::KCOLN_SZ];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), lm = lane & 15, lk = lane >> 4;
  const int quad = (wave + (int)((blockIdx.x + blockIdx.y) % (2 * WC))) % (2 * WC);
  const int wm = quad / WC, wn = quad % WC;
  const int k0 = (int)(blockIdx.x / ntn) * BM;
  const int n0 = (int)(blockIdx.x % ntn) * G::BN;
  const int in = live_tiles(k0 + 64 * wm, K1), jn = live_tiles(n0 + 64 * wn, N);
  const long r0 = (long)blockIdx.y * rows;
  const long r1 = (r0 + rows < M) ? r0 + rows : M;
  float* out = parts + (long)blockIdx.y * K1 * N;
  if (in == 4 && jn == 4) gemm_tn_loop<WC, VEC, false>(A, Z, out, K1, N, r0, r1, k0, n0, wm, wn, in, jn, sA, sZ, t, lm, lk);
  else gemm_tn_loop<WC, VEC, true>(A, Z, out, K1, N, r0, r1, k0, n0, wm, wn, in, jn, sA, sZ, t, lm, lk);
}

// ---- forward product with the layer's epilogue fused: a = act(z + b), ad = act'(z) zd -------------------------------------
// A and C are the stacked matrices of a chunk, [S][c][*]: stream 0 = values, stream 1 = tangents of the same c points.  With
// S = 2 (PAIRED) a workgroup's 128 tile rows are 64 points' value rows (wave row 0) and the same points' tangent rows (wave
// row 1), so act'(z) of a point and its zd are in two waves of one workgroup: the value waves hand s1 = act'(z) over through
// LDS (the operand buffers are free after the K loop), 32 columns per pass.  Saves one read and one write of the M x N
// matrix per layer (the elementwise kernel this replaces ran at the HBM roofline).
__device__ __forceinline__ float gact_f(float z, int act) { return act == VN_ACT_TANH ? tanhf(z) : 1.0f / (1.0f + expf(-z)); }
__device__ __forceinline__ float gact_s1(float a, int act) { return act == VN_ACT_TANH ? 1.f - a * a : a * (1.f - a); }

// tile row tr of a PAIRED workgroup -> row of the stacked matrix; pt = the point
template <bool VEC, int NTHR, bool PAIRED>
__device__ __forceinline__ Stage<512 / NTHR> load_rowk_st(const float* __restrict__ P, long ld, long p0, long c, int k0, int K, int t) {
  Stage<512 / NTHR> s;
#pragma unroll
  for (int e = 0; e < 512 / NTHR; ++e) {
    const int idx = t + e * NTHR;
    const int tr = idx >> 2;
    const long pt = PAIRED ? p0 + (tr & 63) : p0 + tr;
    const long row = PAIRED ? (tr < 64 ? pt : c + pt) : pt;
    const int kq = k0 + (idx & 3) * 4;
    const float* p = P + row * ld + kq;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (pt < c) {
      if constexpr (VEC) {
        if (kq < K) v = *(const f32x4*)p;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (kq + q < K) v[q] = p[q];
      }
    }
    s.v[e] = v;
  }
  return s;
}

template <int WC, bool VEC, bool PAIRED, bool EDGE>
__device__ __forceinline__ void gemm_st_loop(const float* __restrict__ A, const float* __restrict__ B, long c, int N, int K, long p0,
                                             int n0, int wm, int wn, int in, int jn, float* sA0, float* sB0, int t, int lm, int lk,
                                             f32x4 (&acc)[4][4]) {
  using G = Geo<WC>;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nk = (K + BK - 1) / BK;
  auto ga = load_rowk_st<VEC, G::NTHR, PAIRED>(A, K, p0, c, 0, K, t);
  auto gb = load_kcol<VEC, G::NTHR, G::BN>(B, N, 0, K, n0, N, t);
  store_rowk<G::NTHR>(sA0, ga, t);
  store_kcol<G::NTHR, G::BN>(sB0, gb, t);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const int k1 = (kt + 1) * BK;
      ga = load_rowk_st<VEC, G::NTHR, PAIRED>(A, K, p0, c, k1, K, t);
      gb = load_kcol<VEC, G::NTHR, G::BN>(B, N, k1, K, n0, N, t);
    }
    if (!EDGE || (in > 0 && jn > 0)) {
      f32x4 fa[4], fb[4];
      frag_rowk(sA0 + cur * ROWK_SZ, 64 * wm, lm, lk, fa);
      frag_kcol<G::LDN>(sB0 + cur * G::KCOLN_SZ, 64 * wn, lm, lk, fb);
      if (EDGE) mma_step_edge(fa, fb, acc, in, jn); else mma_step(fa, fb, acc);
    }
    if (kt + 1 < nk) {
      store_rowk<G::NTHR>(sA0 + (cur ^ 1) * ROWK_SZ, ga, t);
      store_kcol<G::NTHR, G::BN>(sB0 + (cur ^ 1) * G::KCOLN_SZ, gb, t);
    }
    __syncthreads();
  }
}

constexpr int XLD = 36;      // exchange buffer: [64 rows][32 columns + 4] per wave pair

// (the 256-thread geometry carries two staged 16-byte pieces per operand and thread: at the 128-register cap its K loop
// spills once the epilogue's operands are live across it, so it is compiled for three waves per SIMD)
template <int WC> struct FusedOcc { static constexpr int W = WC == 2 ? 3 : 4; };
template <int WC, bool VEC, bool PAIRED>
__global__ __launch_bounds__(128 * WC, FusedOcc<WC>::W) void vn_gemm_fwd_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, float* __restrict__ C, long c, int N,
                                                                  int K, int act, int ntn) {
  using G = Geo<WC>;
  __shared__ __attribute__((aligned(16))) float smem[2 * ROWK_SZ + 2 * G::KCOLN_SZ];
  static_assert(WC * 64 * XLD <= 2 * ROWK_SZ + 2 * G::KCOLN_SZ, "exchange buffer fits the operand buffers");
  float* sA0 = smem;
  float* sB0 = smem + 2 * ROWK_SZ;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), lm = lane & 15, lk = lane >> 4;
  const long bid = blockIdx.x;
  const int quad = (wave + (int)(bid % (2 * WC))) % (2 * WC);
  const int wm = quad / WC, wn = quad % WC;
  const long p0 = (bid / ntn) * (PAIRED ? 64 : BM);           // first point of this workgroup
  const int n0 = (int)(bid % ntn) * G::BN;
  const long q0 = PAIRED ? p0 : p0 + 64 * wm;                  // first point of this wave's quadrant
  const int in = live_tiles(q0, c), jn = live_tiles(n0 + 64 * wn, N);
  f32x4 acc[4][4];
  if (in == 4 && jn == 4) gemm_st_loop<WC, VEC, PAIRED, false>(A, W, c, N, K, p0, n0, wm, wn, in, jn, sA0, sB0, t, lm, lk, acc);
  else gemm_st_loop<WC, VEC, PAIRED, true>(A, W, c, N, K, p0, n0, wm, wn, in, jn, sA0, sB0, t, lm, lk, acc);
  // epilogue
  float bs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + 64 * wn + 16 * j + lm;
    bs[j] = col < N ? bias[col] : 0.f;
  }
  if constexpr (!PAIRED) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long pt = q0 + 16 * i + 4 * lk + e;
        if (pt < c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = n0 + 64 * wn + 16 * j + lm;
            if (col < N) C[pt * N + col] = gact_f(acc[i][j][e] + bs[j], act);
          }
        }
      }
    }
  } else {
    float* X = smem + wn * (64 * XLD);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (wm == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 16 * i + 4 * lk + e;
            const long pt = q0 + row;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int j = 2 * pass + jj;
              const int col = n0 + 64 * wn + 16 * j + lm;
              const float a = gact_f(acc[i][j][e] + bs[j], act);
              X[row * XLD + 16 * jj + lm] = gact_s1(a, act);
              if (pt < c && col < N) C[pt * N + col] = a;
            }
          }
        }
      }
      __syncthreads();
      if (wm == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 16 * i + 4 * lk + e;
            const long pt = q0 + row;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int j = 2 * pass + jj;
              const int col = n0 + 64 * wn + 16 * j + lm;
              if (pt < c && col < N) C[(c + pt) * N + col] = X[row * XLD + 16 * jj + lm] * acc[i][j][e];
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

// the input layer (K = d_in <= 32) with the same epilogue: a thread forms 4 (VEC) or 1 outputs of a point, both streams
template <bool VEC>
__global__ __launch_bounds__(256) void vn_gemm_fwd_thin_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ C, long c, int S,
                                                              int N, int K, int act) {
  constexpr int V = VEC ? 4 : 1;
  const int nq = (N + V - 1) / V;
  const long total = c * nq;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / nq;
    const int n = (int)(idx - r * nq) * V;
    const float* a = A + r * K;
    const float* ad = A + (c + r) * K;
    float z[V], zd[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { z[v] = bias[n + v]; zd[v] = 0.f; }
    for (int k = 0; k < K; ++k) {
      const float av = a[k], adv = S == 2 ? ad[k] : 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float w = W[(long)k * N + n + v];
        z[v] += av * w;
        zd[v] += adv * w;
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float av = gact_f(z[v], act);
      C[r * N + n + v] = av;
      if (S == 2) C[(c + r) * N + n + v] = gact_s1(av, act) * zd[v];
    }
  }
}

// Wt[N x K] = W[K x N]^T (the input-gradient product runs as  dA = Zb (W^T)  on the kernel above; W is a few hundred KB)
__global__ __launch_bounds__(256) void vn_transpose_kernel(const float* __restrict__ W, float* __restrict__ Wt, int K, int N) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (k0 + i < K && n0 + tx < N) tile[i][tx] = W[(long)(k0 + i) * N + n0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (n0 + i < N && k0 + tx < K) Wt[(long)(n0 + i) * K + k0 + tx] = tile[tx][i];
}

// ---- thin products: the input layer (K = d_in, a handful) --------------------------------------------------------------
// 128 x 128 x 16 MFMA tiles would spend 97 % of their work on padding there; both products are plain streaming kernels bound
// by the M x N matrix they write or read.
// C[M x N] = A[M x K] W[K x N], K <= 32: a thread forms 4 (VEC) or 1 consecutive outputs of a row
template <bool VEC>
__global__ __launch_bounds__(256) void vn_gemm_nn_thin_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ C, long M, int N, int K) {
  constexpr int V = VEC ? 4 : 1;
  const int nq = (N + V - 1) / V;
  const long total = M * nq;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / nq;
    const int n = (int)(idx - r * nq) * V;
    const float* a = A + r * K;
    if constexpr (VEC) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < K; ++k) acc += a[k] * *(const f32x4*)(W + (long)k * N + n);
      *(f32x4*)(C + r * N + n) = acc;
    } else {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += a[k] * W[(long)k * N + n];
      C[r * N + n] = acc;
    }
  }
}
// parts[g][K1 x N] = A_g^T Z_g, K1 <= 32: a workgroup takes the rows of group g for 64 V columns (V = 4: 16-byte loads of Z);
// 4 row lanes x 64 column lanes, eight rows of A^T per pass in registers, the row lanes meet in LDS in a fixed order
template <int V>
__global__ __launch_bounds__(256) void vn_gemm_tn_thin_kernel(const float* __restrict__ A, const float* __restrict__ Z,
                                                             float* __restrict__ parts, long M, int K1, int N, long rows) {
  __shared__ float red[4][8][64 * V];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (blockIdx.y * 64 + tx) * V;
  const long r0 = (long)blockIdx.x * rows;
  const long r1 = (r0 + rows < M) ? r0 + rows : M;
  float* out = parts + (long)blockIdx.x * K1 * N;
  for (int kc = 0; kc < K1; kc += 8) {
    const int kn = (K1 - kc < 8) ? K1 - kc : 8;
    float acc[8][V];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int v = 0; v < V; ++v) acc[j][v] = 0.f;
    }
    if (c < N) {
      for (long r = r0 + ty; r < r1; r += 4) {
        float z[V];
        if constexpr (V == 4) { const f32x4 z4 = *(const f32x4*)(Z + r * N + c); z[0] = z4[0]; z[1] = z4[1]; z[2] = z4[2]; z[3] = z4[3]; }
        else z[0] = Z[r * N + c];
        const float* a = A + r * K1 + kc;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < kn) {
            const float aj = a[j];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[j][v] += aj * z[v];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int v = 0; v < V; ++v) red[ty][j][tx * V + v] = acc[j][v];
    }
    __syncthreads();
    if (c < N) {
      for (int j = ty; j < kn; j += 4) {
#pragma unroll
        for (int v = 0; v < V; ++v)
          out[(long)(kc + j) * N + c + v] = (red[0][j][tx * V + v] + red[1][j][tx * V + v]) + (red[2][j][tx * V + v] + red[3][j][tx * V + v]);
      }
    }
    __syncthreads();
  }
}

// y[r] = beta * y[r] + sum_h A[r][h] w[h]   (the output layer: one column)
__global__ __launch_bounds__(256) void vn_rowdot_kernel(const float* __restrict__ A, const float* __restrict__ w, float* __restrict__ y,
                                                        long M, int H, float beta) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);           // one wave per row, four rows per workgroup
  for (long r = wid; r < M; r += (long)gridDim.x * 4) {
    float acc = 0.f;
    for (int h = lane; h < H; h += 64) acc += A[r * H + h] * w[h];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) y[r] = (beta != 0.f ? beta * y[r] : 0.f) + acc;
  }
}

// ---- fp64: the forward product of the fp64 entry points (vn_forward_f64, vn_residual_f64 of nets on this route) --------
// C[M x N] = A[M x K] W[K x N] in double on v_mfma_f64_16x16x4_f64: workgroup = 4 waves, C tile 64 x 64, wave quadrant
// 32 x 32 (2 x 2 accumulator tiles of 4 doubles), K step 16, one LDS buffer pair, guarded element-wise loads (any
// dimensions).  A check path (the strong residual of Operator_1DtMOR.py:216-224 style post-processing), not a training path:
// written for correctness and a sane fraction of the fp64 matrix rate, not tuned further.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int DLA = 18;       // [row][k] doubles: 16 + 2
constexpr int DLB = 66;       // [k][col] doubles: 64 + 2
__global__ __launch_bounds__(256) void vn_dgemm_nn_kernel(const double* __restrict__ A, const double* __restrict__ W,
                                                        double* __restrict__ C, long M, int N, int K, int ntn) {
  __shared__ double sA[64 * DLA];
  __shared__ double sB[16 * DLB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)(blockIdx.x / ntn) * 64;
  const int n0 = (int)(blockIdx.x % ntn) * 64;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0., 0., 0., 0.};
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    // A tile 64 x 16: thread t -> row t / 4, four k's; W tile 16 x 64: thread t -> k-row t / 16, four columns
    {
      const long row = m0 + (t >> 2);
      const int kq = k0 + (t & 3) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) sA[(t >> 2) * DLA + (t & 3) * 4 + e] = (row < M && kq + e < K) ? A[row * K + kq + e] : 0.;
      const int kr = k0 + (t >> 4);
      const int cq = n0 + (t & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) sB[(t >> 4) * DLB + (t & 15) * 4 + e] = (kr < K && cq + e < N) ? W[(long)kr * N + cq + e] : 0.;
    }
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sA[(32 * wm + 16 * i + lm) * DLA + 4 * s4 + lk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = sB[(4 * s4 + lk) * DLB + 32 * wn + 16 * j + lm];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // accumulator tile (i, j), register e: row 32 wm + 16 i + 4 e + lk, column 32 wn + 16 j + lm  (the fp64 MFMA interleaves the
  // rows of a lane group: register e holds row 4 e + lk, where the fp32 16x16x4 holds row 4 lk + e)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long row = m0 + 32 * wm + 16 * i + 4 * e + lk;
      if (row < M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + 32 * wn + 16 * j + lm;
          if (col < N) C[row * N + col] = acc[i][j][e];
        }
      }
    }
  }
}
__global__ __launch_bounds__(256) void vn_drowdot_kernel(const double* __restrict__ A, const double* __restrict__ w, double* __restrict__ y,
                                                        long M, int H, double beta) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (long r = wid; r < M; r += (long)gridDim.x * 4) {
    double acc = 0.;
    for (int h = lane; h < H; h += 64) acc += A[r * H + h] * w[h];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) y[r] = (beta != 0. ? beta * y[r] : 0.) + acc;
  }
}

bool aligned16(const void* p) { return ((size_t)p & 15) == 0; }

}  // namespace

namespace {
// C tile 128 x 256 (WC = 4) where that pads N no further than 128 x 128 tiles (WC = 2) would: 512 -> 512 either way, but
// 300 -> 512 against 384 (measured, 6.4 M points: 512,512 277.7 vs 280.5 ms per step, 300 x 3 306 vs 244 ms).
// VN_GEMM_WC=2|4 (diagnostic) forces one geometry.
int wave_cols(int N) {
  static const int forced = [] { const char* e = getenv("VN_GEMM_WC"); return (e && (*e == '2' || *e == '4')) ? *e - '0' : 0; }();
  if (forced) return N <= 128 ? 2 : forced;
  return ((N + 255) / 256) * 256 == ((N + 127) / 128) * 128 ? 4 : 2;
}
}  // namespace

int vn_gemm_nn(const float* A, const float* W, float* C, long M, int N, int K, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (K <= 32) {
    const bool v4 = (N % 4 == 0) && aligned16(W) && aligned16(C);
    const long total = M * (v4 ? N / 4 : N);
    long nbt = (total + 255) / 256;
    if (nbt > 256 * 64) nbt = 256 * 64;
    if (v4) hipLaunchKernelGGL(vn_gemm_nn_thin_kernel<true>, dim3((unsigned)nbt), dim3(256), 0, s, A, W, C, M, N, K);
    else hipLaunchKernelGGL(vn_gemm_nn_thin_kernel<false>, dim3((unsigned)nbt), dim3(256), 0, s, A, W, C, M, N, K);
    return (int)hipGetLastError();
  }
  const int wc = wave_cols(N);
  const int bn = 64 * wc;
  const int ntn = (N + bn - 1) / bn;
  const long nb = ((M + BM - 1) / BM) * ntn;
  const bool vec = (K % 4 == 0) && (N % 4 == 0) && aligned16(A) && aligned16(W);
#define VN_LAUNCH_NN(WC_, VEC_) hipLaunchKernelGGL((vn_gemm_kernel<WC_, VEC_>), dim3((unsigned)nb), dim3(128 * WC_), 0, s, A, W, C, M, N, K, ntn)
  if (wc == 4) { if (vec) VN_LAUNCH_NN(4, true); else VN_LAUNCH_NN(4, false); }
  else         { if (vec) VN_LAUNCH_NN(2, true); else VN_LAUNCH_NN(2, false); }
#undef VN_LAUNCH_NN
  return (int)hipGetLastError();
}

// C[S][c][N] = epilogue(A[S][c][K] W[K x N] + b): a = act(z + b) in stream 0, ad = act'(z) zd in stream 1 (S = 2)
int vn_gemm_fwd(const float* A, const float* W, const float* bias, float* C, long c, int S, int N, int K, int act, hipStream_t s) {
  if (c <= 0 || N <= 0 || K <= 0) return 0;
  if (K <= 32) {
    const bool v4 = (N % 4 == 0);
    const long total = c * (v4 ? N / 4 : N);
    long nbt = (total + 255) / 256;
    if (nbt > 256 * 64) nbt = 256 * 64;
    if (v4) hipLaunchKernelGGL(vn_gemm_fwd_thin_kernel<true>, dim3((unsigned)nbt), dim3(256), 0, s, A, W, bias, C, c, S, N, K, act);
    else hipLaunchKernelGGL(vn_gemm_fwd_thin_kernel<false>, dim3((unsigned)nbt), dim3(256), 0, s, A, W, bias, C, c, S, N, K, act);
    return (int)hipGetLastError();
  }
  const int wc = wave_cols(N);
  const int bn = 64 * wc;
  const int ntn = (N + bn - 1) / bn;
  const long nb = ((c + (S == 2 ? 63 : BM - 1)) / (S == 2 ? 64 : BM)) * ntn;
  const bool vec = (K % 4 == 0) && (N % 4 == 0) && aligned16(A) && aligned16(W) && (S == 1 || (c * (long)K) % 4 == 0);
#define VN_LAUNCH_FW(WC_, VEC_, P_) hipLaunchKernelGGL((vn_gemm_fwd_kernel<WC_, VEC_, P_>), dim3((unsigned)nb), dim3(128 * WC_), 0, s, A, W, bias, C, c, N, K, act, ntn)
  if (S == 2) {
    if (wc == 4) { if (vec) VN_LAUNCH_FW(4, true, true); else VN_LAUNCH_FW(4, false, true); }
    else         { if (vec) VN_LAUNCH_FW(2, true, true); else VN_LAUNCH_FW(2, false, true); }
  } else {
    if (wc == 4) { if (vec) VN_LAUNCH_FW(4, true, false); else VN_LAUNCH_FW(4, false, false); }
    else         { if (vec) VN_LAUNCH_FW(2, true, false); else VN_LAUNCH_FW(2, false, false); }
  }
#undef VN_LAUNCH_FW
  return (int)hipGetLastError();
}

int vn_transpose(const float* W, float* Wt, int K, int N, hipStream_t s) {
  if (K <= 0 || N <= 0) return 0;
  hipLaunchKernelGGL(vn_transpose_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0, s, W, Wt, K, N);
  return (int)hipGetLastError();
}

// rows per group for vn_gemm_tn_parts: the MFMA kernel keeps 16 waves per CU resident, so (output tiles x groups) is made to
// fill those slots evenly (a launch of 585 equal workgroups on 256 CUs runs as long as one of 768); the thin kernel streams
// and wants many small groups
long vn_gemm_tn_rows(long M, int K1, int N, int ncu) {
  if (K1 <= 32) return M < 512 ? (M > 0 ? M : 1) : 512;
  const int wc = wave_cols(N);
  const int bn = 64 * wc;
  const long tiles = ((K1 + BM - 1) / BM) * (long)((N + bn - 1) / bn);
  long groups = ((wc == 4 ? 2l : 4l) * ncu) / tiles;
  if (groups < 1) groups = 1;
  long rows = (M + groups - 1) / groups;
  rows = (rows + BK - 1) / BK * BK;
  if (rows < 1024) rows = 1024;                      // never more partials than the sum is worth
  return rows;
}

int vn_gemm_tn_parts(const float* A, const float* Z, float* parts, long M, int K1, int N, long rows, hipStream_t s) {
  if (M <= 0 || N <= 0 || K1 <= 0 || rows <= 0) return 0;
  if (K1 <= 32) {
    if (N % 4 == 0 && aligned16(Z))
      hipLaunchKernelGGL(vn_gemm_tn_thin_kernel<4>, dim3((unsigned)((M + rows - 1) / rows), (unsigned)((N + 255) / 256)), dim3(256), 0, s, A,
                         Z, parts, M, K1, N, rows);
    else
      hipLaunchKernelGGL(vn_gemm_tn_thin_kernel<1>, dim3((unsigned)((M + rows - 1) / rows), (unsigned)((N + 63) / 64)), dim3(256), 0, s, A, Z,
                         parts, M, K1, N, rows);
    return (int)hipGetLastError();
  }
  const int wc = wave_cols(N);
  const int bn = 64 * wc;
  const int ntn = (N + bn - 1) / bn;
  const int nb = ((K1 + BM - 1) / BM) * ntn;
  const long groups = (M + rows - 1) / rows;
  const bool vec = (K1 % 4 == 0) && (N % 4 == 0) && aligned16(A) && aligned16(Z);
#define VN_LAUNCH_TN(WC_, VEC_) hipLaunchKernelGGL((vn_gemm_tn_kernel<WC_, VEC_>), dim3((unsigned)nb, (unsigned)groups), dim3(128 * WC_), 0, s, A, Z, parts, M, K1, N, rows, ntn)
  if (wc == 4) { if (vec) VN_LAUNCH_TN(4, true); else VN_LAUNCH_TN(4, false); }
  else         { if (vec) VN_LAUNCH_TN(2, true); else VN_LAUNCH_TN(2, false); }
#undef VN_LAUNCH_TN
  return (int)hipGetLastError();
}

int vn_rowdot(const float* A, const float* w, float* y, long M, int H, float beta, hipStream_t s) {
  if (M <= 0) return 0;
  long nb = (M + 3) / 4;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(vn_rowdot_kernel, dim3((unsigned)nb), dim3(256), 0, s, A, w, y, M, H, beta);
  return (int)hipGetLastError();
}

int vn_dgemm_nn(const double* A, const double* W, double* C, long M, int N, int K, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int ntn = (N + 63) / 64;
  const long nb = ((M + 63) / 64) * ntn;
  hipLaunchKernelGGL(vn_dgemm_nn_kernel, dim3((unsigned)nb), dim3(256), 0, s, A, W, C, M, N, K, ntn);
  return (int)hipGetLastError();
}

int vn_drowdot(const double* A, const double* w, double* y, long M, int H, double beta, hipStream_t s) {
  if (M <= 0) return 0;
  long nb = (M + 3) / 4;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(vn_drowdot_kernel, dim3((unsigned)nb), dim3(256), 0, s, A, w, y, M, H, beta);
  return (int)hipGetLastError();
}
