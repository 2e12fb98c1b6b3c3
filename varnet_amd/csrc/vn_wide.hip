// Tile kernels of the layer-by-layer route for hidden widths up to 256 (vn_layered.hip hands such networks over).
//
// Why: with one GEMM per layer a 128-wide layer moves 16 B of HBM per 2*128 FLOP in each direction -- the GEMM route is
// HBM-bound there (3 x 128 at 6.4 M points: 91 ms = 0.18 of the fp32 MFMA peak).  Here a workgroup carries a tile of 32
// points (64 columns: value | tangent) through ALL layers with the activations in LDS, so HBM sees the inputs, the
// outputs and one write + one read of the stored activations (a_l, ad_l), nothing else.
//
// Same (value, one tangent) recurrences as every other route (oracle/tangent_ref.py; TFModel.py:536 input gradient,
// :653-661 integrand, :709 parameter gradient):
//   forward  z = W^T a + b, zd = W^T ad, a' = act(z), ad' = act'(z) zd
//   reverse  zdbar = s1 adbar, zbar = s1 abar + s2r ad adbar, abar_{l-1} = W zbar, dW = a zbar^T + ad zdbar^T
//
// Geometry (8 waves, v_mfma_f32_16x16x4_f32):
//   * activations of a tile: LDS matrix [feature][64 columns], row stride 68; column 4 lm + ct = point 16 (ct & 1) + lm of
//     stream ct >> 1, i.e. the four accumulator tiles of a lane side by side (one ds_read_b128 / ds_write_b128 per row);
//   * a layer is D[feature x column] = W^T[feature x k] A[k x column]: wave w owns the 16 features of row tile w and all
//     four column tiles (value 0..15, 16..31 | tangent 0..15, 16..31), so value and tangent of one (feature, point) sit
//     in the same lane and the activation runs in registers -- one workgroup barrier per layer, no elementwise sweep;
//   * weights never touch LDS: a tiny pack kernel rewrites W (and W^T for the reverse pass) in MFMA fragment order
//     ([row tile][k quad][lane][4 k-steps]) once per call, a wave streams its fragments from L2 with one coalesced
//     16-byte load per 16 MFMAs, three loads in flight;
//   * forward writes (a_l | ad_l) of every layer to HBM in accumulator order ([tile][layer][row tile][column tile]
//     [lane][4]: 1 KB per store instruction), the reverse kernel reads them back with the same wave assignment, one
//     layer ahead of their use;
//   * reverse: per layer the weight gradient contracts over the 64 columns with both operands read transposed from
//     LDS; the 8 waves are a 4 x 2 grid of BM x BN blocks of 16 x 16 output tiles (BM + BN fragment reads for BM BN
//     MFMAs), accumulated in registers over all tiles of the launch and written once as a per-workgroup partial that a
//     fixed-order sum adds to the gradient (no atomics: same bits every run); the input gradient is the forward loop with
//     the W^T image, its epilogue forms (zbar, zdbar) of the layer below in registers.  Two barriers per layer.
//   * nets up to 64 wide have at most four row tiles: waves w and w + 4 then share row tile w, each with the value and the
//     tangent column tile of one half of the points (SPLIT; +3 %: one wave per SIMD already kept the matrix pipes busy, the
//     second hides latency);
//   * widths 129..256: two row-tile passes per wave and layer (LDS matrices of 256 rows); nets whose accumulators do not fit
//     the registers all at once (5+ layers wider than 96, 7+ wider than 64, widths above 128) run the reverse pass one layer per
//     launch (vn_wide_lbwd_kernel: accumulators of one layer in registers, the adjoints travel through HBM; up to 128 wide it
//     is cut to 128 registers and two workgroups share a CU).
// Measured at 6.4 M points (profiles/r2_layered_perf.txt): 3 x 128 26.7 ms = 0.61 of peak (GEMM form 86 ms), 4 x 128 0.63,
// 8 x 128 0.66, 3 x 256 120.8 ms = 0.53 (GEMM form 172 ms).
#include "vn_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#ifndef VN_WIDE_ABL
#define VN_WIDE_ABL 0      // diagnostic builds: 1 no activation stores, 2 no transcendentals, 3 no MFMAs in the layer GEMMs
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TP = 32;        // points per tile
constexpr int LDW = 68;       // LDS row stride (floats): 16-byte aligned rows for the b128 accesses
constexpr int NT = 512;       // threads per workgroup
constexpr int BROWS = 128;    // rows of the reverse kernel's two LDS matrices (always the widest net: see its weight-gradient loop)
constexpr int WL = VN_MAX_LAYERS;   // most hidden layers these kernels take (the ABI's limit)

struct Plan {
  int nrt[WL + 1];    // 16-row tiles of layer l's activations (l = 0: the inputs)
  int wfo[WL + 1];    // float offset of layer l's fragment image for the forward GEMM (l = 1..L)
  int wto[WL + 1];    // ... for the input-gradient GEMM (W^T; l = 2..L)
  int ko[WL + 1];     // float offset of layer l's (a | ad) inside a tile's block of stored activations (l = 1..L)
  int kept_tile;      // floats per tile
  int rows;           // LDS rows per matrix
  int wf_floats;      // both images
};

// Both activations in one branch-free form (the choice is per layer and wave-uniform, but a branch or a select per element
// costs vector issue slots, which the f32 MFMAs share):  a = sA rcp(1 + 2^(c z)) - sB  with (sA, sB) = (1, 0) for the
// sigmoid and (2, 1) for tanh = 2 sigmoid(2z) - 1, c = -sA log2(e);  act' = (1 - a)(a + sB);  act''/act' = (1 - sB) - 2a.
struct ActK { float sA, sB, c; };
__device__ __forceinline__ ActK act_consts(int act) {
  ActK k;
  // selects between literals, not arithmetic: the condition is wave-uniform, so these stay scalar registers
  k.sA = act == VN_ACT_TANH ? 2.f : 1.f;
  k.sB = act == VN_ACT_TANH ? 1.f : 0.f;
  k.c = act == VN_ACT_TANH ? -2.8853900817779268f : -1.4426950408889634f;
  return k;
}
__device__ __forceinline__ float w_act(float z, const ActK& k) {
  return fmaf(k.sA, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k.c * z)), -k.sB);
}
__device__ __forceinline__ float w_d1(float a, const ActK& k) { return (1.f - a) * (a + k.sB); }
__device__ __forceinline__ float w_d2r(float a, const ActK& k) { return fmaf(-2.f, a, 2.f - k.sA); }

// runtime-selected forms (the reverse kernel: its register budget does not like the straight-line variant, measured)
__device__ __forceinline__ float w_d1(float a, int act) { return act == VN_ACT_TANH ? 1.f - a * a : a * (1.f - a); }
__device__ __forceinline__ float w_d2r(float a, int act) { return act == VN_ACT_TANH ? -2.f * a : 1.f - 2.f * a; }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Column order of the LDS matrices: column 4 lm + ct holds point 16 (ct & 1) + lm of stream ct >> 1 (0 value, 1 tangent),
// i.e. the four accumulator tiles of a lane side by side -- a lane reads its four B operands of a k-step with one
// ds_read_b128 and writes a row of its results with one ds_write_b128.  The weight-gradient contraction runs over the
// columns in storage order (both operands use the same order).

// ---- fragment images of the weights ---------------------------------------------------------------------------------
// forward image of layer l:   [rt][q][lane][j] = W[k][m],  k = 16 q + 4 j + lane/16 (input),  m = 16 rt + lane%16 (output)
// transposed image (l >= 2):  [rt][q][lane][j] = W[r][n],  r = 16 rt + lane%16 (input),      n = 16 q + 4 j + lane/16 (output)
// zero outside the matrix, so padded rows and k-steps contribute nothing.
__global__ __launch_bounds__(256) void vn_wide_pack_kernel(VnNet net, Plan pl, const float* __restrict__ theta,
                                                           float* __restrict__ wf) {
  const int l = blockIdx.y + 1;
  const int Hin = net.H[l - 1], Hout = net.H[l];
  const float* W = theta + net.woff[l];
  const int nf = pl.nrt[l] * pl.nrt[l - 1] * 256;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 2 * nf; idx += gridDim.x * 256) {
    const bool tr = idx >= nf;
    if (tr && l == 1) break;
    const int e = tr ? idx - nf : idx;
    const int j = e & 3, lane = (e >> 2) & 63, blk = e >> 8;
    const int nq = tr ? pl.nrt[l] : pl.nrt[l - 1];
    const int q = blk % nq, rt = blk / nq;
    const int kk = 16 * q + 4 * j + (lane >> 4), mm = 16 * rt + (lane & 15);
    float v = 0.f;
    if (!tr) { if (kk < Hin && mm < Hout) v = W[kk * Hout + mm]; }
    else     { if (mm < Hin && kk < Hout) v = W[mm * Hout + kk]; }
    wf[(tr ? pl.wto[l] : pl.wfo[l]) + e] = v;
  }
}

// inputs of a tile, through registers so that the loads of the next tile fly under the current one
struct TileIn { float v[4]; };
__device__ __forceinline__ TileIn tile_in_issue(const VnNet& net, const VnRows& sg, long r0, int rows0, int tid) {
  TileIn t;
  const int d_in = net.d_in, dim = net.dim;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = tid + e * NT;
    float v = 0.f;
    if (i < rows0 * 64) {
      const int k = i >> 6, c = i & 63;
      const long row = r0 + 16 * (c & 1) + (c >> 2);
      if (row < sg.n && k < d_in) {
        if (!(c & 2)) v = sg.X[row * d_in + k];
        else if (sg.G != nullptr && k < dim) v = sg.G[row * dim + k];
      }
    }
    t.v[e] = v;
  }
  return t;
}
__device__ __forceinline__ void tile_in_write(const TileIn& t, int rows0, float* dst, int tid) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = tid + e * NT;
    if (i < rows0 * 64) dst[(i >> 6) * LDW + (i & 63)] = t.v[e];
  }
}

// the first three 16-byte fragment loads of a wave's GEMM, issued ahead of the barrier in front of it
struct Frag { f32x4 w0, w1, w2; };
__device__ __forceinline__ Frag frag_issue(const float* __restrict__ img, int wave, int lane, int nq) {
  const f32x4* wp = (const f32x4*)img + (long)wave * nq * 64 + lane;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Frag f;
  f.w0 = wp[0];
  f.w1 = nq > 1 ? wp[64] : zero4;
  f.w2 = nq > 2 ? wp[128] : zero4;
  return f;
}

// acc[ct] += sum_k image[this wave's row tile][k] * B[k][4 lm + ct]: the GEMM of a layer (forward: B = activations of the
// layer below; reverse: B = (zbar | zdbar), image = W^T).  nq = k quads of 16.
//  * Three 16-byte fragment loads stay in flight.  The quad loop is unrolled by four over a register ring with static
//    indices: rotating the ring with moves makes the move of the newest fragment wait for its own load (vmcnt(0) in every
//    iteration -- the prefetch was void).
//  * The B operand of the next k-step is read while the four MFMAs of the current one issue; the scheduler otherwise
//    sinks every ds_read to its use to save registers (read, wait, 4 MFMAs, read, ...), hence the group barriers.
//  * NC = 4: the wave owns all four column tiles of its row tile.  NC = 2 (nets up to 64 wide, whose four row tiles would
//    leave waves 4..7 idle): waves w and w + 4 share row tile w; wave half `ph` owns the value and the tangent column tile of
//    points 16 ph .. 16 ph + 15 (columns 4 lm + ph and 4 lm + 2 + ph: two ds_read_b32 instead of one ds_read_b128).
template <int NC>
__device__ __forceinline__ void wave_gemm(const float* __restrict__ img, int wave, int lane, int nq, const float* B, f32x4 (&acc)[NC],
                                          Frag f, int ph = 0) {
  static_assert(NC == 4 || NC == 2, "column tiles per wave");
  const int lm = lane & 15, lk = lane >> 4;
  const f32x4* wp = (const f32x4*)img + (long)wave * nq * 64 + lane;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 ring[4] = {f.w0, f.w1, f.w2, zero4};
  const float* cb = B + lk * LDW + 4 * lm + (NC == 2 ? ph : 0);
  auto rdb = [](const float* q) {
    if constexpr (NC == 4) return *(const f32x4*)q;
    else return f32x4{q[0], q[2], 0.f, 0.f};
  };
  f32x4 bv = rdb(cb);
#pragma unroll 1
  for (int q0 = 0; q0 < nq; q0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u;
      if (q < nq) {
        // unconditional (clamped): a conditional load makes the wait-count bookkeeping fall back to vmcnt(0) at the join
        ring[(u + 3) & 3] = wp[(q + 3 < nq ? q + 3 : nq - 1) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cb += 4 * LDW;
          // one row past the last k-step is still inside the matrix or the one behind it (never used)
          const f32x4 bn = rdb(cb);
          const float av = ring[u][j];
#if VN_WIDE_ABL == 3
          acc[j % NC] += av * bv;
#else
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16(av, bv[ct], acc[ct]);
#endif
          __builtin_amdgcn_sched_group_barrier(0x100, NC == 4 ? 1 : 2, 0);     // the next step's ds_read(s) first ...
          __builtin_amdgcn_sched_group_barrier(0x008, NC, 0);                  // ... then this step's MFMAs
          bv = bn;
        }
      }
    }
  }
}

// ---- forward: rows -> (u, ud) [+ stored activations] ----------------------------------------------------------------
// RP = row-tile passes per layer: wave w owns row tiles w, w + 8, ... (RP = 1: widths <= 128; RP = 2: <= 256)
// SPLIT (nets up to 64 wide, RP = 1): waves w and w + 4 share row tile w, each with the value and tangent column tile of one
// half of the tile's 32 points (wave_gemm<2>), so all eight waves run the layer GEMMs of a net that has only four row tiles.
template <int RP, bool SPLIT>
__global__ __launch_bounds__(NT) void vn_wide_fwd_kernel(VnNet net, Plan pl, const float* __restrict__ theta,
                                                         const float* __restrict__ wf, VnRows sg, long ntiles,
                                                         float* __restrict__ kept) {
  static_assert(!SPLIT || RP == 1, "the column split serves nets with at most four row tiles");
  constexpr int NC = SPLIT ? 2 : 4;          // column tiles per wave
  constexpr int NH = NC / 2;                 // point halves per wave
  extern __shared__ float lds[];
  // the wave index as a scalar: everything addressed by it (row tile, fragment image, stored-activation block) is then
  // formed on the scalar unit -- the vector unit shares its datapath with the f32 MFMAs
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lm = lane & 15, lk = lane >> 4;
  const int wrt = SPLIT ? (wave & 3) : wave;     // first row tile of this wave
  const int ph = SPLIT ? (wave >> 2) : 0;        // SPLIT: which half of the points
  float* buf0 = lds;
  float* buf1 = lds + pl.rows * LDW;
  float* red = buf1 + pl.rows * LDW;         // [8][64]
  const int L = net.L;
  const int rows0 = 16 * pl.nrt[0];

  TileIn tin = tile_in_issue(net, sg, (long)blockIdx.x * TP, rows0, tid);
  Frag fr{};
  if (wrt < pl.nrt[1]) fr = frag_issue(wf + pl.wfo[1], wrt, lane, pl.nrt[0]);

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long r0 = tile * TP;
    tile_in_write(tin, rows0, buf0, tid);
    if (tile + gridDim.x < ntiles) tin = tile_in_issue(net, sg, (tile + gridDim.x) * TP, rows0, tid);
    __syncthreads();
    float* cur = buf0;
    float* nxt = buf1;
    for (int l = 1; l <= L; ++l) {
      const int Hout = net.H[l], act = net.actl[l];
#pragma unroll
      for (int rp = 0; rp < RP; ++rp) {
        const int rt = wrt + 8 * rp;
        const bool active = rt < pl.nrt[l];
        f32x4 acc[NC];
        float bs[4];
        if (active) {
          const float* bias = theta + net.boff[l];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = 16 * rt + 4 * lk + i;
            bs[i] = m < Hout ? bias[m] : 0.f;
          }
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
          wave_gemm<NC>(wf + pl.wfo[l], rt, lane, pl.nrt[l - 1], cur, acc, fr, ph);
        }
        {   // fragments of the next GEMM this wave runs (its next row tile, the next layer, or layer 1 of the next tile):
            // ahead of the epilogue and the barrier
          const bool same = rp + 1 < RP && rt + 8 < pl.nrt[l];
          const int ln = same ? l : (l < L ? l + 1 : 1);
          const int rn = same ? rt + 8 : wrt;
          if (rn < pl.nrt[ln]) fr = frag_issue(wf + pl.wfo[ln], rn, lane, pl.nrt[ln - 1]);
        }
        if (active) {
          f32x4 av[NH], adv[NH];
          const ActK ak = act_consts(act);
          // rows past the layer's width (only in its last row tile) must hold zeros, not act(0)
          const bool ragged = 16 * rt + 16 > Hout;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = 16 * rt + 4 * lk + i;
            const float vm = (!ragged || m < Hout) ? 1.f : 0.f;
            f32x4 o;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
#if VN_WIDE_ABL == 2
              float a = 0.5f + 0.25f * (acc[h][i] + bs[i]);
#else
              float a = w_act(acc[h][i] + bs[i], ak);
#endif
              if (ragged) a *= vm;                       // wave-uniform branch; acc[NH + h] is zero there (zero weights)
              const float ad = w_d1(a, ak) * acc[NH + h][i];
              av[h][i] = a; adv[h][i] = ad;
              o[h] = a; o[2 + h] = ad;
            }
            if constexpr (SPLIT) {
              nxt[m * LDW + 4 * lm + ph] = o[0];
              nxt[m * LDW + 4 * lm + 2 + ph] = o[2];
            } else {
              *(f32x4*)(nxt + m * LDW + 4 * lm) = o;
            }
          }
#if VN_WIDE_ABL != 1
          if (kept != nullptr) {
            // [column tile][lane] blocks of 16 bytes: value tiles 0, 1, tangent tiles 2, 3 (the same block layout either way)
            f32x4* kp = (f32x4*)(kept + tile * pl.kept_tile + pl.ko[l] + rt * 1024);
            if constexpr (SPLIT) { kp[ph * 64 + lane] = av[0]; kp[(2 + ph) * 64 + lane] = adv[0]; }
            else { kp[lane] = av[0]; kp[64 + lane] = av[1]; kp[128 + lane] = adv[0]; kp[192 + lane] = adv[1]; }
          }
#endif
        }
      }
      __syncthreads();
      float* t = cur; cur = nxt; nxt = t;
    }
    // u = w_o . a_L + b_o,  ud = w_o . ad_L : eight feature slices per column, added in a fixed order
    {
      const int HL = net.H[L];
      const float* wo = theta + net.woff[L + 1];
      const int c = tid & 63, k0 = 16 * RP * (tid >> 6);
      float p = 0.f;
      for (int k = k0; k < k0 + 16 * RP && k < HL; ++k) p += wo[k] * cur[k * LDW + c];
      red[tid] = p;
    }
    __syncthreads();
    if (tid < 64) {
      float acc = 0.f;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) acc += red[sl * 64 + tid];
      const long row = r0 + 16 * (tid & 1) + (tid >> 2);
      if (row < sg.n) {
        if (!(tid & 2)) { if (sg.u != nullptr) sg.u[row] = acc + theta[net.boff[L + 1]]; }
        else if (sg.ud != nullptr) sg.ud[row] = acc;
      }
    }
  }
}

// ---- reverse: rows + seeds + stored activations -> per-workgroup partial parameter gradient ---------------------------
// Diagnostic build -DVN_WIDE_CAP128 (round-2 experiment, re-run in round 3: DESIGN.md appendix): cap the one-launch reverse
// kernel at 128 registers so that two workgroups share a CU (it spills 114-269 VGPRs) -- grid and partial buffer double.
#ifdef VN_WIDE_CAP128
#define VN_WIDE_BWD_BOUNDS __launch_bounds__(NT, 4)
constexpr int BWD_WG_PER_CU = 2;
#else
#define VN_WIDE_BWD_BOUNDS __launch_bounds__(NT)
constexpr int BWD_WG_PER_CU = 1;
#endif
// SPLIT (nets up to 64 wide): as in the forward kernel, waves w and w + 4 share row tile w in everything that is per
// (feature, point) -- the output-layer seeds, the input-gradient GEMMs and their epilogues -- each with one half of the
// points; the weight-gradient blocks and the bias sums contract over all 64 columns of the LDS matrices and are unchanged.
template <int ML, int BM, int BN, bool SPLIT>
__global__ VN_WIDE_BWD_BOUNDS void vn_wide_bwd_kernel(VnNet net, Plan pl, const float* __restrict__ theta,
                                                         const float* __restrict__ wf, VnRows sg, long ntiles,
                                                         const float* __restrict__ kept, float* __restrict__ partial) {
  constexpr int NC = SPLIT ? 2 : 4;          // column tiles per wave: [stream][point half] = NH value tiles, then NH tangent tiles
  constexpr int NH = NC / 2;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave0 = tid >> 6, lm0 = lane & 15, lk0 = lane >> 4;
  const int wrt0 = SPLIT ? (wave0 & 3) : wave0, ph0 = SPLIT ? (wave0 >> 2) : 0;
  float* T = lds;                            // (zbar | zdbar) of the current layer
  float* PV = lds + BROWS * LDW;             // (a | ad) of the layer below (layer 1: the inputs)
  float* sub = PV + BROWS * LDW;             // [TP] ubar
  float* sudb = sub + TP;                    // [TP] udbar
  const int L = net.L;
  const int rows0 = 16 * pl.nrt[0];

  // weight-gradient accumulators, live over the whole launch.  Hidden-to-hidden layers (l >= 2): the BM x BN block of this
  // wave.  Layer 1 has at most 2 x 8 tiles (d_in <= 32): column tile `wave`, both row tiles.
  f32x4 wacc[ML > 1 ? ML - 1 : 1][BM * BN];
  f32x4 wacc1[2];
  float bacc[ML];
  wacc1[0] = wacc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < ML; ++l) bacc[l] = 0.f;
#pragma unroll
  for (int l = 0; l < (ML > 1 ? ML - 1 : 1); ++l) {
#pragma unroll
    for (int j = 0; j < BM * BN; ++j) wacc[l][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float woacc[4] = {0.f, 0.f, 0.f, 0.f};
  float boacc = 0.f;

  // stored activations of the last two layers and the seeds of a tile are fetched while the tile before it is processed
  f32x4 ka[NC], kn[NC];
  float su = 0.f, sd = 0.f;
  const bool ownL = wrt0 < pl.nrt[L];
  const bool ownL1 = L > 1 && wrt0 < pl.nrt[L > 1 ? L - 1 : 0];
  // stored block of a row tile: [column tile 0..3][lane] x 16 bytes; this wave's tile j is column tile ctile(j)
  auto ctile = [&](int j, int ph) { return SPLIT ? 2 * j + ph : j; };
  auto fetch_head = [&](long tile) {
    const float* kt = kept + tile * pl.kept_tile;
    if (ownL) {
      const f32x4* kp = (const f32x4*)(kt + pl.ko[L] + wrt0 * 1024);
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) ka[ct] = kp[ctile(ct, ph0) * 64 + lane];
    }
    if (ownL1) {
      const f32x4* kp = (const f32x4*)(kt + pl.ko[L - 1] + wrt0 * 1024);
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) kn[ct] = kp[ctile(ct, ph0) * 64 + lane];
    }
    if (tid < TP) {
      const long row = tile * TP + tid;
      su = (row < sg.n) ? sg.ubar[row] : 0.f;
      sd = (row < sg.n && sg.udbar != nullptr) ? sg.udbar[row] : 0.f;
    }
  };
  fetch_head(blockIdx.x);

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // The layer loop below is unrolled (the accumulators are registers), and the compiler would hoist every layer's LDS and
    // global addresses out of this loop into registers that stay live for the whole launch (126 spilled VGPRs in the
    // 16-layer instantiation).  An opaque zero makes the lane coordinates a value of this iteration, so addresses are
    // formed where they are used.
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const int lm = lm0 + oz, lk = lk0 + oz, wave = __builtin_amdgcn_readfirstlane(wave0) + oz, wm = wave >> 1, wn = wave & 1;
    const int wrt = SPLIT ? (wave & 3) : wave, ph = SPLIT ? (wave >> 2) : 0;      // row tile / point half of this wave
    const long r0 = tile * TP;
    const float* kt = kept + tile * pl.kept_tile;
    TileIn tin{};                                                     // inputs: consumed at layer 1, the end of the tile
    if (L == 1) tin = tile_in_issue(net, sg, r0, rows0, tid);
    if (tid < TP) { sub[tid] = su; sudb[tid] = sd; }
    __syncthreads();        // seeds visible; every wave is past the previous tile's reads of T and PV

    // output layer: d w_o, d b_o and (zbar | zdbar) of the last hidden layer
    if (ownL) {
      const int HL = net.H[L], act = net.actl[L];
      const float* wo = theta + net.woff[L + 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * wrt + 4 * lk + i;
        const bool valid = m < HL;
        const float wom = valid ? wo[m] : 0.f;
        f32x4 o;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int p = 16 * (SPLIT ? ph : h) + lm;
          const float ub = sub[p], udb = sudb[p];
          const float a = ka[h][i], ad = ka[NH + h][i];
          woacc[i] += ub * a + udb * ad;
          const float ab = ub * wom, adb = udb * wom;
          const float sp = w_d1(a, act);
          o[h] = valid ? ab * sp + w_d2r(a, act) * ad * adb : 0.f;
          o[2 + h] = valid ? adb * sp : 0.f;
        }
        if constexpr (SPLIT) {
          T[m * LDW + 4 * lm + ph] = o[0];
          T[m * LDW + 4 * lm + 2 + ph] = o[2];
        } else {
          *(f32x4*)(T + m * LDW + 4 * lm) = o;
        }
      }
    }
    if (wave == 0 && lane < TP) boacc += sub[lane];

#pragma unroll
    for (int l = ML; l >= 1; --l) {
      if (l <= L) {
        const int Hout = net.H[l];
        const bool own = l > 1 && wrt < pl.nrt[l - 1];
        if (l > 1) {
          if (own) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int m = 16 * wrt + 4 * lk + i;
              if constexpr (SPLIT) {
                PV[m * LDW + 4 * lm + ph] = kn[0][i];
                PV[m * LDW + 4 * lm + 2 + ph] = kn[1][i];
              } else {
                *(f32x4*)(PV + m * LDW + 4 * lm) = f32x4{kn[0][i], kn[1][i], kn[2][i], kn[3][i]};
              }
            }
          }
        } else {
          // the input rows are written by all threads, and rows 0..31 of PV belong to waves 0/1, which may still be in
          // their layer-2 epilogue (it re-reads PV): one extra barrier per tile
          if (L > 1) __syncthreads();
          tile_in_write(tin, rows0, PV, tid);
        }
        __syncthreads();      // #1: T and PV of this layer complete
        // loads that fly under this layer's work: the W^T fragments of its input-gradient GEMM, the stored activations
        // two layers down (layer 1: the head of the next tile)
        Frag fr{};
        if (own) fr = frag_issue(wf + pl.wto[l], wrt, lane, pl.nrt[l]);
        if (l > 2) {
          if (wrt < pl.nrt[l - 2]) {
            const f32x4* kp = (const f32x4*)(kt + pl.ko[l - 2] + wrt * 1024);
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) kn[ct] = kp[ctile(ct, ph) * 64 + lane];
          }
        } else if (l == 1) {
          if (tile + gridDim.x < ntiles) fetch_head(tile + gridDim.x);
        }
        if (l == 2) tin = tile_in_issue(net, sg, r0, rows0, tid);
        // bias gradient: row sums of the value columns; thread t takes 8 of the 32 value columns of row t / 4 (four 8-byte
        // reads), the four partial sums of a row meet at the flush
        {
          const int row = tid >> 2;
          if (row < Hout) {
            const float* tr = T + row * LDW + 16 * (tid & 3);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float2 v = *(const float2*)(tr + 4 * e);
              acc += v.x + v.y;
            }
            bacc[l - 1] += acc;
          }
        }
        // weight gradient: G[k][n] += sum_col PV[k][col] T[n][col]
        if (l == 1) {
          if (wave < pl.nrt[1]) {
            const float* pa = PV + lm * LDW + lk;
            const float* pb = T + (16 * wave + lm) * LDW + lk;
            const bool a1 = pl.nrt[0] > 1;
#pragma unroll 4
            for (int cs = 0; cs < 16; ++cs) {
              const float bv = pb[4 * cs];
              wacc1[0] = mfma16(pa[4 * cs], bv, wacc1[0]);
              if (a1) wacc1[1] = mfma16(pa[16 * LDW + 4 * cs], bv, wacc1[1]);
            }
          }
        } else {
          const int li = l > 1 ? l - 2 : 0;
          const int ntm = pl.nrt[l - 1], ntn = pl.nrt[l];
          const int tm0 = BM * wm, tn0 = BN * wn;
          if (tm0 < ntm && tn0 < ntn) {
            // Straight-line body: a block at the ragged edge of the tile grid (widths that are not a multiple of the block)
            // also multiplies its absent tiles -- rows of LDS that exist (both matrices are allocated 128 rows) but hold
            // whatever an earlier layer left there; those accumulators are never written out.  Per-tile conditions cost a
            // scalar branch around every MFMA and a conservative wait count after every read.
            const float* pa = PV + (16 * tm0 + lm) * LDW + lk;
            const float* pb = T + (16 * tn0 + lm) * LDW + lk;
            float av[BM], bv[BN];
#pragma unroll
            for (int a = 0; a < BM; ++a) av[a] = pa[16 * a * LDW];
#pragma unroll
            for (int b = 0; b < BN; ++b) bv[b] = pb[16 * b * LDW];
#pragma unroll 2
            for (int cs = 0; cs < 16; ++cs) {
              // operands of the next 4 columns are read while the MFMAs of the current ones issue
              const int cn = cs < 15 ? 4 * (cs + 1) : 0;
              float nav[BM], nbv[BN];
#pragma unroll
              for (int a = 0; a < BM; ++a) nav[a] = pa[16 * a * LDW + cn];
#pragma unroll
              for (int b = 0; b < BN; ++b) nbv[b] = pb[16 * b * LDW + cn];
#pragma unroll
              for (int b = 0; b < BN; ++b) {
#pragma unroll
                for (int a = 0; a < BM; ++a) wacc[li][a * BN + b] = mfma16(av[a], bv[b], wacc[li][a * BN + b]);
              }
              __builtin_amdgcn_sched_group_barrier(0x100, BM + BN, 0);     // the next step's reads first ...
              __builtin_amdgcn_sched_group_barrier(0x008, BM * BN, 0);     // ... then this step's MFMAs
#pragma unroll
              for (int a = 0; a < BM; ++a) av[a] = nav[a];
#pragma unroll
              for (int b = 0; b < BN; ++b) bv[b] = nbv[b];
            }
          }
        }
        // input gradient of the layer: (abar | adbar)_{l-1}[k][col] = sum_n W[k][n] T[n][col]
        f32x4 acc[NC];
        if (own) {
#pragma unroll
          for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
          wave_gemm<NC>(wf + pl.wto[l], wrt, lane, pl.nrt[l], T, acc, fr, ph);
        }
        __syncthreads();      // #2: every wave is done reading T and PV
        if (own) {
          const int Hp = net.H[l - 1], actp = net.actl[l - 1];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = 16 * wrt + 4 * lk + i;
            const bool valid = m < Hp;
            // (a | ad) of this lane's positions: columns this wave wrote
            f32x4 pv;
            if constexpr (SPLIT) pv = f32x4{PV[m * LDW + 4 * lm + ph], 0.f, PV[m * LDW + 4 * lm + 2 + ph], 0.f};
            else pv = *(const f32x4*)(PV + m * LDW + 4 * lm);
            f32x4 o;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
              const float a = pv[h], ad = pv[2 + h];
              const float ab = acc[h][i], adb = acc[NH + h][i];
              const float sp = w_d1(a, actp);
              o[h] = valid ? ab * sp + w_d2r(a, actp) * ad * adb : 0.f;
              o[2 + h] = valid ? adb * sp : 0.f;
            }
            if constexpr (SPLIT) {
              T[m * LDW + 4 * lm + ph] = o[0];
              T[m * LDW + 4 * lm + 2 + ph] = o[2];
            } else {
              *(f32x4*)(T + m * LDW + 4 * lm) = o;
            }
          }
        }
      }
    }
  }

  // ---- this workgroup's partial gradient (flat parameter layout) ----
  const int lm = lm0, lk = lk0, wave = wave0, wm = wave >> 1, wn = wave & 1;
  float* out = partial + (long)blockIdx.x * net.P;
#pragma unroll
  for (int l = 1; l <= ML; ++l) {
    if (l <= L) {
      const int Hin = net.H[l - 1], Hout = net.H[l];
      const int ntm = pl.nrt[l - 1], ntn = pl.nrt[l];
      if (l == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (a < ntm && wave < ntn) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 16 * a + 4 * lk + i, col = 16 * wave + lm;
              if (row < Hin && col < Hout) out[net.woff[1] + row * Hout + col] = wacc1[a][i];
            }
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < BM; ++a) {
#pragma unroll
          for (int b = 0; b < BN; ++b) {
            const int tm = BM * wm + a, tn = BN * wn + b;
            if (tm < ntm && tn < ntn) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int row = 16 * tm + 4 * lk + i, col = 16 * tn + lm;
                if (row < Hin && col < Hout) out[net.woff[l] + row * Hout + col] = wacc[l > 1 ? l - 2 : 0][a * BN + b][i];
              }
            }
          }
        }
      }
      {
        float v = bacc[l - 1];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if ((tid & 3) == 0 && (tid >> 2) < Hout) out[net.boff[l] + (tid >> 2)] = v;
      }
    }
  }
  {
    const int HL = net.H[L];
    float wsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = woacc[i];
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
      wsum[i] = v;
    }
    if constexpr (SPLIT) {
      // the two waves of a row tile each hold the sum over their half of the points: second half -> LDS -> first half
      __syncthreads();                       // every wave has left the tile loop: T is free
      if (ph0 == 1 && lm == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) T[(wrt0 * 4 + lk) * 4 + i] = wsum[i];
      }
      __syncthreads();
      if (ph0 == 0 && lm == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wsum[i] += T[(wrt0 * 4 + lk) * 4 + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * wrt0 + 4 * lk + i;
      if (ph0 == 0 && lm == 0 && m < HL) out[net.woff[L + 1] + m] = wsum[i];
    }
    if (wave == 0) {
      float v = lane < TP ? boacc : 0.f;
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) out[net.boff[L + 1]] = v;
    }
  }
}

// ---- reverse, ONE LAYER PER LAUNCH ---------------------------------------------------------------------------------
// For nets whose weight-gradient accumulators do not fit the registers all at once (5+ layers wider than 96, 7+ wider than 64;
// widths 129..256: a 256 x 256 layer alone is 128 registers per lane).  Launch l = L..1 reads (zbar | zdbar)_l of a tile from HBM (layer L:
// forms it from the seeds), the stored (a | ad)_{l-1}, accumulates dW_l, db_l over all tiles in registers and writes
// (zbar | zdbar)_{l-1} back in the same block layout -- 2 x 2 H floats more HBM traffic per point and layer than the
// register-resident kernel above, still far from the HBM roofline at these widths.
// partial: [workgroup][W_l, b_l (, w_o, b_o when l = L)] -- the flat parameter order, so one fixed-order sum adds it to grad.
// (one row-tile pass: capped at 128 registers so that two workgroups share a CU -- one layer's accumulators are only 32)
template <int RP, int BM, int BN>
__global__ __launch_bounds__(NT, RP == 1 ? 4 : 2) void vn_wide_lbwd_kernel(VnNet net, Plan pl, int l, const float* __restrict__ theta,
                                                          const float* __restrict__ wf, VnRows sg, long ntiles,
                                                          const float* __restrict__ kept, const float* __restrict__ zin,
                                                          float* __restrict__ zout, int zstride, float* __restrict__ partial,
                                                          int plen) {
  extern __shared__ float lds[];
  constexpr int R = 128 * RP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lm = lane & 15, lk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  float* T = lds;
  float* PV = lds + R * LDW;
  float* sub = PV + R * LDW;
  float* sudb = sub + TP;
  const int L = net.L;
  const bool top = l == L;
  const int Hin = net.H[l - 1], Hout = net.H[l];
  const int ntm = pl.nrt[l - 1], ntn = pl.nrt[l];
  const int rows0 = 16 * pl.nrt[0];

  f32x4 wacc[BM * BN];      // layer 1 (at most 2 x 16 tiles): wacc[2 rp + a] = tile (a, wave + 8 rp)
  float bacc[RP], woacc[RP][4], boacc = 0.f;
#pragma unroll
  for (int j = 0; j < BM * BN; ++j) wacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rp = 0; rp < RP; ++rp) {
    bacc[rp] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) woacc[rp][i] = 0.f;
  }

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long r0 = tile * TP;
    const float* kt = kept + tile * pl.kept_tile;
    // ---- stage T = (zbar | zdbar)_l and PV = (a | ad)_{l-1} (layer 1: the inputs) ----
    TileIn tin{};
    if (l == 1) tin = tile_in_issue(net, sg, r0, rows0, tid);
    f32x4 t4[RP][4];
#pragma unroll
    for (int rp = 0; rp < RP; ++rp) {
      const int rt = wave + 8 * rp;
      if (rt < ntn) {
        const f32x4* kp = top ? (const f32x4*)(kt + pl.ko[L] + rt * 1024) : (const f32x4*)(zin + tile * zstride + rt * 1024);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) t4[rp][ct] = kp[ct * 64 + lane];
      }
      if (l > 1 && rt < ntm) {
        // (a | ad)_{l-1}: HBM -> registers -> LDS; the epilogue of the input gradient reads it back from there
        const f32x4* kp = (const f32x4*)(kt + pl.ko[l - 1] + rt * 1024);
        f32x4 kb[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) kb[ct] = kp[ct * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 16 * rt + 4 * lk + i;
          *(f32x4*)(PV + m * LDW + 4 * lm) = f32x4{kb[0][i], kb[1][i], kb[2][i], kb[3][i]};
        }
      }
    }
    if (top) {
      if (tid < TP) {
        const long row = r0 + tid;
        sub[tid] = (row < sg.n) ? sg.ubar[row] : 0.f;
        sudb[tid] = (row < sg.n && sg.udbar != nullptr) ? sg.udbar[row] : 0.f;
      }
      __syncthreads();
      if (wave == 0 && lane < TP) boacc += sub[lane];
    }
#pragma unroll
    for (int rp = 0; rp < RP; ++rp) {
      const int rt = wave + 8 * rp;
      if (rt < ntn) {
        const int act = net.actl[L];
        const float* wo = theta + net.woff[L + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 16 * rt + 4 * lk + i;
          f32x4 o;
          if (top) {
            // output layer: d w_o and (zbar | zdbar)_L from the seeds and the stored (a | ad)_L
            const bool valid = m < Hout;
            const float wom = valid ? wo[m] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int p = 16 * h + lm;
              const float ub = sub[p], udb = sudb[p];
              const float a = t4[rp][h][i], ad = t4[rp][2 + h][i];
              woacc[rp][i] += ub * a + udb * ad;
              const float ab = ub * wom, adb = udb * wom;
              const float sp = w_d1(a, act);
              o[h] = valid ? ab * sp + w_d2r(a, act) * ad * adb : 0.f;
              o[2 + h] = valid ? adb * sp : 0.f;
            }
          } else {
            o = f32x4{t4[rp][0][i], t4[rp][1][i], t4[rp][2][i], t4[rp][3][i]};
          }
          *(f32x4*)(T + m * LDW + 4 * lm) = o;
        }
      }
    }
    if (l == 1) tile_in_write(tin, rows0, PV, tid);
    __syncthreads();      // #1: T and PV complete

    // bias gradient: row sums of the value columns (thread t: 8 of the 32 value columns of row t / 4 [+ 128])
#pragma unroll
    for (int rp = 0; rp < RP; ++rp) {
      const int row = (tid >> 2) + 128 * rp;
      if (row < Hout) {
        const float* tr = T + row * LDW + 16 * (tid & 3);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float2 v = *(const float2*)(tr + 4 * e);
          acc += v.x + v.y;
        }
        bacc[rp] += acc;
      }
    }
    // weight gradient: G[k][n] += sum_col PV[k][col] T[n][col]
    if (l == 1) {
#pragma unroll
      for (int rp = 0; rp < RP; ++rp) {
        const int tn = wave + 8 * rp;
        if (tn < ntn) {
          const float* pa = PV + lm * LDW + lk;
          const float* pb = T + (16 * tn + lm) * LDW + lk;
          const bool a1 = ntm > 1;
#pragma unroll 4
          for (int cs = 0; cs < 16; ++cs) {
            const float bv = pb[4 * cs];
            wacc[2 * rp] = mfma16(pa[4 * cs], bv, wacc[2 * rp]);
            if (a1) wacc[2 * rp + 1] = mfma16(pa[16 * LDW + 4 * cs], bv, wacc[2 * rp + 1]);
          }
        }
      }
    } else {
      const int tm0 = BM * wm, tn0 = BN * wn;
      if (tm0 < ntm && tn0 < ntn) {
        // straight-line block (see the register-resident kernel): absent tiles at the ragged edge are multiplied too, their
        // rows exist in LDS, their accumulators are never written out
        const float* pa = PV + (16 * tm0 + lm) * LDW + lk;
        const float* pb = T + (16 * tn0 + lm) * LDW + lk;
        float av[BM], bv[BN];
#pragma unroll
        for (int a = 0; a < BM; ++a) av[a] = pa[16 * a * LDW];
#pragma unroll
        for (int b = 0; b < BN; ++b) bv[b] = pb[16 * b * LDW];
#pragma unroll 2
        for (int cs = 0; cs < 16; ++cs) {
          const int cn = cs < 15 ? 4 * (cs + 1) : 0;
          float nav[BM], nbv[BN];
#pragma unroll
          for (int a = 0; a < BM; ++a) nav[a] = pa[16 * a * LDW + cn];
#pragma unroll
          for (int b = 0; b < BN; ++b) nbv[b] = pb[16 * b * LDW + cn];
#pragma unroll
          for (int b = 0; b < BN; ++b) {
#pragma unroll
            for (int a = 0; a < BM; ++a) wacc[a * BN + b] = mfma16(av[a], bv[b], wacc[a * BN + b]);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, BM + BN, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, BM * BN, 0);
#pragma unroll
          for (int a = 0; a < BM; ++a) av[a] = nav[a];
#pragma unroll
          for (int b = 0; b < BN; ++b) bv[b] = nbv[b];
        }
      }
    }
    // input gradient + adjoint of the activation below: (zbar | zdbar)_{l-1} straight from registers to HBM
    if (l > 1) {
      const int actp = net.actl[l - 1];
#pragma unroll
      for (int rp = 0; rp < RP; ++rp) {
        const int rt = wave + 8 * rp;
        if (rt < ntm) {
          f32x4 acc[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
          wave_gemm(wf + pl.wto[l], rt, lane, ntn, T, acc, frag_issue(wf + pl.wto[l], rt, lane, ntn));
          f32x4 zo[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = 16 * rt + 4 * lk + i;
            const bool valid = m < Hin;
            const f32x4 pv = *(const f32x4*)(PV + m * LDW + 4 * lm);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float a = pv[h], ad = pv[2 + h];
              const float ab = acc[h][i], adb = acc[2 + h][i];
              const float sp = w_d1(a, actp);
              zo[h][i] = valid ? ab * sp + w_d2r(a, actp) * ad * adb : 0.f;
              zo[2 + h][i] = valid ? adb * sp : 0.f;
            }
          }
          f32x4* zp = (f32x4*)(zout + tile * zstride + rt * 1024);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) zp[ct * 64 + lane] = zo[ct];
        }
      }
    }
    __syncthreads();      // #2: every wave is done with T and PV before the next tile overwrites them
  }

  // ---- this workgroup's partial of layer l ----
  float* out = partial + (long)blockIdx.x * plen;
  if (l == 1) {
#pragma unroll
    for (int rp = 0; rp < RP; ++rp) {
      const int tn = wave + 8 * rp;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (a < ntm && tn < ntn) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 16 * a + 4 * lk + i, col = 16 * tn + lm;
            if (row < Hin && col < Hout) out[row * Hout + col] = wacc[2 * rp + a][i];
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < BM; ++a) {
#pragma unroll
      for (int b = 0; b < BN; ++b) {
        const int tm = BM * wm + a, tn = BN * wn + b;
        if (tm < ntm && tn < ntn) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 16 * tm + 4 * lk + i, col = 16 * tn + lm;
            if (row < Hin && col < Hout) out[row * Hout + col] = wacc[a * BN + b][i];
          }
        }
      }
    }
  }
#pragma unroll
  for (int rp = 0; rp < RP; ++rp) {
    float v = bacc[rp];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    const int row = (tid >> 2) + 128 * rp;
    if ((tid & 3) == 0 && row < Hout) out[Hin * Hout + row] = v;
  }
  if (top) {
#pragma unroll
    for (int rp = 0; rp < RP; ++rp) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = woacc[rp][i];
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
        const int m = 16 * (wave + 8 * rp) + 4 * lk + i;
        if (lm == 0 && m < Hout) out[Hin * Hout + Hout + m] = v;
      }
    }
    if (wave == 0) {
      float v = lane < TP ? boacc : 0.f;
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) out[Hin * Hout + 2 * Hout] = v;
    }
  }
}

// dst[i] += sum_b part[b][i], fixed order: 64 elements x 8 groups of partials per workgroup; group g adds the contiguous run
// of partials [g n/8, (g+1) n/8), the eight group sums meet in LDS in a fixed order
constexpr int SUMG = 8;
__global__ __launch_bounds__(64 * SUMG) void vn_wide_sum_kernel(const float* __restrict__ part, int nparts, long len, float* __restrict__ dst) {
  __shared__ float red[SUMG][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (i < len) {
    const int per = (nparts + SUMG - 1) / SUMG;
    const int b0 = g * per, b1 = b0 + per < nparts ? b0 + per : nparts;
    const float* p = part + i;
    int bb = b0;
    for (; bb + 4 <= b1; bb += 4)
      acc += (p[(long)bb * len] + p[(long)(bb + 1) * len]) + (p[(long)(bb + 2) * len] + p[(long)(bb + 3) * len]);
    for (; bb < b1; ++bb) acc += p[(long)bb * len];
  }
  red[g][tx] = acc;
  __syncthreads();
  if (g == 0 && i < len) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < SUMG; ++k) v += red[k][tx];
    dst[i] += v;
  }
}

int wfail(char* err, size_t n, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (err && n) vsnprintf(err, n, fmt, ap);
  va_end(ap);
  return 1;
}
#define WHIP(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return wfail(err, errlen, "%s: %s", #expr, hipGetErrorString(e_));    \
  } while (0)

}  // namespace

struct VnWide {
  VnNet net{};
  Plan pl{};
  // reverse pass: 0 <4,2,4> (<= 4 layers <= 128 wide)  1 <6,2,3> (5-6 layers <= 96)  2 <16,1,2> (<= 64 wide) -- all layers in one
  // launch, accumulators in registers --  4 / 5: one launch per layer (5+ layers wider than 96, 7+ wider than 64 / widths 129..256)
  int variant = 0;
  int rp = 1;                   // row-tile passes per layer (2: widths 129..256)
  bool split = false;           // nets up to 64 wide: two waves per row tile, each with half of the points
  int cus = 256;
  size_t lds_f = 0, lds_b = 0;
  float* wf = nullptr;
  float* part = nullptr;
  int zstride = 0;              // floats per tile of a (zbar | zdbar) buffer of the layer-serial reverse pass
  float* zbuf[2] = {nullptr, nullptr};
  size_t zcap = 0;
  struct Kept { float* buf = nullptr; size_t cap = 0; const float* X = nullptr; long n = 0; bool valid = false; } kept[2];
};

bool vn_wide_supported(const VnNet& net) {
  const char* off = getenv("VN_LAYERED_NOWIDE");
  if (off && *off && *off != '0') return false;
  if (net.L < 1 || net.L > WL || net.d_in > 32) return false;
  int hmax = 0;
  for (int l = 1; l <= net.L; ++l) {
    if (net.H[l] > hmax) hmax = net.H[l];
    if (net.actl[l] != VN_ACT_SIGMOID && net.actl[l] != VN_ACT_TANH) return false;
  }
  return hmax <= 256;           // two LDS matrices of 256 rows x 64 columns are what a workgroup can hold
}

int vn_wide_create(VnWide** out, const VnNet& net, char* err, size_t errlen) {
  *out = nullptr;
  VnWide* w = new VnWide();
  w->net = net;
  Plan& pl = w->pl;
  int rows = 0, off = 0, koff = 0, hm = 0, maxnrt = 0, maxplen = net.P;
  for (int l = 0; l <= net.L; ++l) {
    pl.nrt[l] = (net.H[l] + 15) / 16;
    if (16 * pl.nrt[l] > rows) rows = 16 * pl.nrt[l];
    if (l >= 1 && net.H[l] > hm) hm = net.H[l];
    if (l >= 1 && pl.nrt[l] > maxnrt) maxnrt = pl.nrt[l];
  }
  for (int l = 1; l <= net.L; ++l) {
    const int nf = pl.nrt[l] * pl.nrt[l - 1] * 256;
    pl.wfo[l] = off; off += nf;
    pl.wto[l] = off; if (l > 1) off += nf;
    pl.ko[l] = koff; koff += pl.nrt[l] * 1024;
  }
  pl.wf_floats = off; pl.kept_tile = koff; pl.rows = rows;
  w->rp = hm > 128 ? 2 : 1;
  if (hm > 128) w->variant = 5;
  else if (hm <= 64) w->variant = net.L <= 4 ? 3 : 2;       // at most 4 x 4 tiles per layer: column-split kernels
  else w->variant = net.L <= 4 ? 0 : (net.L <= 6 && hm <= 96) ? 1 : 4;
  {
    // diagnostic: VN_WIDE_SERIAL=1 runs every net up to 128 wide on the layer-serial reverse pass (what it costs, measured)
    const char* sv = getenv("VN_WIDE_SERIAL");
    if (sv && *sv && *sv != '0' && hm <= 128) w->variant = 4;
  }
  w->split = hm <= 64;
  w->zstride = maxnrt * 1024;
  w->lds_f = ((size_t)2 * rows * LDW + 512) * sizeof(float);
  w->lds_b = ((size_t)2 * BROWS * w->rp * LDW + 2 * TP) * sizeof(float);
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    w->cus = prop.multiProcessorCount;
  const void* fk = w->rp == 2 ? (const void*)vn_wide_fwd_kernel<2, false>
                   : w->split ? (const void*)vn_wide_fwd_kernel<1, true> : (const void*)vn_wide_fwd_kernel<1, false>;
  const void* bk = w->variant == 0 ? (const void*)vn_wide_bwd_kernel<4, 2, 4, false>
                   : w->variant == 1 ? (const void*)vn_wide_bwd_kernel<6, 2, 3, false>
                   : w->variant == 2 ? (const void*)vn_wide_bwd_kernel<16, 1, 2, true>
                   : w->variant == 3 ? (const void*)vn_wide_bwd_kernel<4, 1, 2, true>
                   : w->variant == 4 ? (const void*)vn_wide_lbwd_kernel<1, 2, 4> : (const void*)vn_wide_lbwd_kernel<2, 4, 8>;
  hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w->lds_f);
  if (e == hipSuccess) e = hipFuncSetAttribute(bk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w->lds_b);
  if (e == hipSuccess) e = hipMalloc((void**)&w->wf, (size_t)pl.wf_floats * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&w->part, (size_t)w->cus * (w->variant == 4 ? 2 : BWD_WG_PER_CU) * maxplen * sizeof(float));
  if (e != hipSuccess) {
    vn_wide_destroy(w);
    return wfail(err, errlen, "vn_wide_create: %s", hipGetErrorString(e));
  }
  *out = w;
  return 0;
}

void vn_wide_destroy(VnWide* w) {
  if (!w) return;
  if (w->wf) (void)hipFree(w->wf);
  if (w->part) (void)hipFree(w->part);
  for (float* z : w->zbuf) if (z) (void)hipFree(z);
  for (auto& k : w->kept) if (k.buf) (void)hipFree(k.buf);
  delete w;
}

namespace {
int pack(VnWide* w, const float* theta, hipStream_t s, char* err, size_t errlen) {
  hipLaunchKernelGGL(vn_wide_pack_kernel, dim3(32, w->net.L), dim3(256), 0, s, w->net, w->pl, theta, w->wf);
  WHIP(hipGetLastError());
  return 0;
}
// grow a device buffer if half of the free memory allows it
bool reserve(float** buf, size_t* cap, size_t need) {
  if (need <= *cap) return true;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess || (need - *cap) * sizeof(float) > fr / 2) return false;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr; *cap = 0;
  if (hipMalloc((void**)buf, need * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return false; }
  *cap = need;
  return true;
}
}  // namespace

int vn_wide_forward(VnWide* w, const float* theta, const VnRows& seg, int keep_slot, bool may_keep, hipStream_t s, char* err,
                    size_t errlen) {
  if (keep_slot >= 0) w->kept[keep_slot].valid = false;
  if (seg.n <= 0) return 0;
  const long ntiles = (seg.n + TP - 1) / TP;
  float* kbuf = nullptr;
  if (keep_slot >= 0 && may_keep) {
    VnWide::Kept& k = w->kept[keep_slot];
    bool ok = reserve(&k.buf, &k.cap, (size_t)ntiles * w->pl.kept_tile);
    if (ok && w->variant >= 4) {
      // the layer-serial reverse pass also needs its two (zbar | zdbar) buffers
      const size_t zneed = (size_t)ntiles * w->zstride;
      if (zneed > w->zcap) {
        size_t c0 = w->zcap, c1 = w->zcap;
        ok = reserve(&w->zbuf[0], &c0, zneed) && reserve(&w->zbuf[1], &c1, zneed);
        w->zcap = ok ? zneed : 0;
        if (!ok) { for (float*& z : w->zbuf) { if (z) (void)hipFree(z); z = nullptr; } }
      }
    }
    if (ok) kbuf = k.buf;
  }
  if (int rc = pack(w, theta, s, err, errlen)) return rc;
  const long grid = ntiles < 2l * w->cus ? ntiles : 2l * w->cus;
  if (w->rp == 2)
    hipLaunchKernelGGL((vn_wide_fwd_kernel<2, false>), dim3((unsigned)grid), dim3(NT), w->lds_f, s, w->net, w->pl, theta, (const float*)w->wf,
                       seg, ntiles, kbuf);
  else if (w->split)
    hipLaunchKernelGGL((vn_wide_fwd_kernel<1, true>), dim3((unsigned)grid), dim3(NT), w->lds_f, s, w->net, w->pl, theta, (const float*)w->wf,
                       seg, ntiles, kbuf);
  else
    hipLaunchKernelGGL((vn_wide_fwd_kernel<1, false>), dim3((unsigned)grid), dim3(NT), w->lds_f, s, w->net, w->pl, theta, (const float*)w->wf,
                       seg, ntiles, kbuf);
  WHIP(hipGetLastError());
  if (kbuf) {
    VnWide::Kept& k = w->kept[keep_slot];
    k.X = seg.X; k.n = seg.n; k.valid = true;
  }
  return 0;
}

bool vn_wide_has_kept(const VnWide* w, int slot, const VnRows& seg) {
  if (slot < 0) return false;
  const VnWide::Kept& k = w->kept[slot];
  return k.valid && k.X == seg.X && k.n == seg.n;
}

int vn_wide_backward(VnWide* w, const float* theta, const VnRows& seg, float* grad, int keep_slot, hipStream_t s, char* err,
                     size_t errlen) {
  if (seg.n <= 0) return 0;
  if (!vn_wide_has_kept(w, keep_slot, seg)) return wfail(err, errlen, "vn_wide_backward: no stored activations for these rows");
  VnWide::Kept& k = w->kept[keep_slot];
  k.valid = false;                                  // theta moves after this step
  const long ntiles = (seg.n + TP - 1) / TP;
  if (int rc = pack(w, theta, s, err, errlen)) return rc;
  const long wgs = (long)w->cus * (w->variant == 4 ? 2 : w->variant < 4 ? BWD_WG_PER_CU : 1);   // layer-serial, one pass: two workgroups per CU
  const int grid = (int)(ntiles < wgs ? ntiles : wgs);
  const VnNet& net = w->net;
  if (w->variant >= 4) {
    // one launch per layer, last to first; (zbar | zdbar) travels through the two buffers
    if ((size_t)ntiles * w->zstride > w->zcap) return wfail(err, errlen, "vn_wide_backward: adjoint buffers missing");
    for (int l = net.L; l >= 1; --l) {
      const int plen = net.H[l - 1] * net.H[l] + net.H[l] + (l == net.L ? net.H[l] + 1 : 0);
      const float* zin = w->zbuf[(net.L - l + 1) & 1];
      float* zout = w->zbuf[(net.L - l) & 1];
      if (w->variant == 4)
        hipLaunchKernelGGL((vn_wide_lbwd_kernel<1, 2, 4>), dim3(grid), dim3(NT), w->lds_b, s, net, w->pl, l, theta, (const float*)w->wf,
                           seg, ntiles, (const float*)k.buf, zin, zout, w->zstride, w->part, plen);
      else
        hipLaunchKernelGGL((vn_wide_lbwd_kernel<2, 4, 8>), dim3(grid), dim3(NT), w->lds_b, s, net, w->pl, l, theta, (const float*)w->wf,
                           seg, ntiles, (const float*)k.buf, zin, zout, w->zstride, w->part, plen);
      WHIP(hipGetLastError());
      hipLaunchKernelGGL(vn_wide_sum_kernel, dim3((unsigned)((plen + 63) / 64)), dim3(64 * SUMG), 0, s, (const float*)w->part, grid,
                         (long)plen, grad + net.woff[l]);
      WHIP(hipGetLastError());
    }
    return 0;
  }
#define VN_WIDE_BWD(ML_, BM_, BN_, SP_)                                                                                \
  hipLaunchKernelGGL((vn_wide_bwd_kernel<ML_, BM_, BN_, SP_>), dim3(grid), dim3(NT), w->lds_b, s, w->net, w->pl, theta,    \
                     (const float*)w->wf, seg, ntiles, (const float*)k.buf, w->part)
  if (w->variant == 0) VN_WIDE_BWD(4, 2, 4, false);
  else if (w->variant == 1) VN_WIDE_BWD(6, 2, 3, false);
  else if (w->variant == 3) VN_WIDE_BWD(4, 1, 2, true);
  else VN_WIDE_BWD(16, 1, 2, true);
#undef VN_WIDE_BWD
  WHIP(hipGetLastError());
  hipLaunchKernelGGL(vn_wide_sum_kernel, dim3((unsigned)((w->net.P + 63) / 64)), dim3(64 * SUMG), 0, s, (const float*)w->part, grid,
                     (long)w->net.P, grad);
  WHIP(hipGetLastError());
  return 0;
}
