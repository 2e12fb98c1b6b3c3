// Fused gfx950 kernel for the VarNet variational-loss training step: forward (value + one
// directional tangent), weak-form epilogue (R_k, lossVec, seeds) and the complete reverse pass
// to parameter gradients in ONE persistent launch, for hidden widths <= 2*KS.
//
// Design (DESIGN.md "fused kernel"):
//   * one workgroup (4 waves, 1 wave / SIMD, up to 512 VGPRs) per CU, looping over 128-point
//     tiles; each wave owns 32 points;
//   * every layer is D[feature x point] = W^T . A with v_mfma_f32_32x32x2_f32; the accumulator
//     tile of layer l IS the B operand of layer l+1 (column = lane&31 = point, rows in the
//     16 registers x 2 lane halves), so activations never leave registers: feature f lives in
//     k-step ks = f/2, lane half g = f%2, at accumulator row pos(ks,g);
//   * activations (a_l, zdot_l) of all layers stay in registers for the reverse pass;
//   * weights sit in LDS once per workgroup as [in-feature][out-position] images (stride 65,
//     conflict-free for both the forward and the transposed backward fragment reads);
//   * weight gradients contract over points, which needs the operands transposed: all four waves
//     publish their 32 point-columns of (a | adot) and (zbar | zdbar) into two shared LDS images
//     [64 rows][128 points] (ds_write_b32), barrier, and wave w contracts ONE 32x32 output tile over
//     the 128 points (ds_read_b128 fragments) into a persistent register accumulator per layer -- no
//     atomics in the tile loop, fixed summation order; the bias gradient rides along as a
//     constant-one input row.  After the loop the accumulators are added wave by wave into an LDS
//     gradient image and one partial per workgroup goes to HBM.
//   * this is the 4-wave (one wave per SIMD) geometry; vn_fused16.hip is the 8-wave geometry of the
//     same algorithm, which VN_KERNEL_AUTO prefers where it is instantiated.
//
// Math: oracle/tangent_ref.py; reference graph TFModel.py:536, 643-668, 709.
#include "vn_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// vector view of LDS words that are also written as scalars: exempt from type-based alias analysis
typedef f32x4 f32x4a __attribute__((may_alias));

namespace {

constexpr int NTHREADS = 256;
constexpr int TILE = 128;     // points per workgroup tile (4 waves x 32)
constexpr int WS = 65;        // weight image row stride (floats)
constexpr int TSW = 132;      // transposition buffer row stride (floats): 128 points + 4, 16-B aligned rows
constexpr int TROWS = 64;
constexpr int KS0 = 4;        // input layer k-steps (d_in <= 8)

__host__ __device__ constexpr int al4(int x) { return (x + 3) & ~3; }
// accumulator row of feature (ks, g)
__host__ __device__ constexpr int vpos(int ks, int g) {
  return 32 * (ks >> 4) + 8 * ((ks & 15) >> 2) + 4 * g + (ks & 3);
}
// k-step owning accumulator row `pos`
__host__ __device__ constexpr int vks(int pos) {
  return 16 * (pos >> 5) + 4 * ((pos & 31) >> 3) + (pos & 3);
}
__host__ __device__ constexpr int vfeat(int pos) { return 2 * vks(pos) + (((pos & 31) >> 2) & 1); }
// first accumulator row not used by KS k-steps: carries the constant-one "bias" input
__host__ __device__ constexpr int vones(int KS) {
  for (int p = 0; p < 64; ++p)
    if (vks(p) >= KS) return p;
  return -1;
}
__host__ __device__ constexpr int mtiles(int KS) { return KS > 16 ? 2 : 1; }

template <int L, int KS>
struct Lay {
  static constexpr int HP = 2 * KS;
  static constexpr int HPWS = al4(HP * WS);
  static constexpr int W1_OFF = 0;                          // [8][WS]
  static constexpr int WH_OFF = al4(8 * WS);                // [L-1][HP][WS]
  static constexpr int BI_OFF = WH_OFF + (L - 1) * HPWS;    // [L][64] biases in (mt, g, i) order
  static constexpr int WO_OFF = BI_OFF + L * 64;            // [2*KS] output weights by feature
  static constexpr int MISC_OFF = WO_OFF + al4(2 * KS);     // sInt[128] | sR[128]
  static constexpr int T_OFF = MISC_OFF + 256;              // TA | TB: [TROWS][TSW] each (shared by the 4 waves)
  // gradient image (only used once, after the tile loop; aliases TA|TB): layer 1 block
  // [2*KS0+1][HP], hidden blocks [HP+1][HP], output block [HP+1][1]; last row = bias gradient.
  static constexpr int G1_SZ = (2 * KS0 + 1) * HP;
  static constexpr int GH_SZ = (HP + 1) * HP;
  static constexpr int GO_OFF = G1_SZ + (L - 1) * GH_SZ;
  static constexpr int G_SZ = al4(GO_OFF + HP + 1);
  static constexpr int T_SZ = (2 * TROWS * TSW > G_SZ) ? 2 * TROWS * TSW : G_SZ;
  static constexpr int TOTAL = T_OFF + T_SZ;
};

__device__ __forceinline__ float fsigmoid(float z) {
#ifdef VN_EXP_NOSIG   // timing-only diagnostic
  return z * 0.25f;
#endif
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}

// Identity the optimiser cannot see through.  The reverse pass recomputes sigma' = a(1-a) and
// adot = sigma' * zdot from the stored activations; without this, GVN merges those expressions
// with their forward-pass twins and keeps ~250 extra values alive across the whole tile (spills).
__device__ __forceinline__ float opaque(float x) {
  asm("" : "+v"(x));
  return x;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct LaneC {
  int g, c;
  int offF[2];     // forward A-fragment lane offset inside a weight image
  int offB[2];     // backward (transposed) A-fragment lane offset
  int twr;         // transposition write offset: (4g)*TSW + wave*32 + c
};

// One weight-gradient contraction for a layer, cooperative over the workgroup's 128-point tile:
//   G[in pos][out pos] += sum_pts Aside[in][pt] * Bside[out][pt]   (value half, then tangent half)
// Every wave writes its 32 point-columns of the transposed operands into the shared LDS images
// TA / TB; after a barrier wave w contracts ONE 32x32 output tile (m, n) over its share of the
// points into a persistent register accumulator (no atomics, fixed summation order).  With NT
// output tiles, tile = w % NT and the 128 points are split over the NS = 4/NT waves per tile.
// av/azd: A-side registers per k-step (a, zdot) in accumulator layout; RAWA: inputs (tangent = azd
// as is) instead of a*(1-a)*zd.  bv/bt: B-side registers (zbar, zdbar).
template <int KSA, int KSB>
struct WG {
  static constexpr int ONES = vones(KSA);
  static constexpr int MTA = (mtiles(KSA) > (ONES >> 5) + 1) ? mtiles(KSA) : (ONES >> 5) + 1;
  static constexpr int NTB = mtiles(KSB);
  static constexpr int NT = MTA * NTB;          // 1, 2 or 4
  static constexpr int NS = 4 / NT;             // point splits
  static constexpr int PTS = TILE / NS;         // points contracted by one wave
  static constexpr int ones_m = ONES >> 5;
  static constexpr int ones_i = 4 * ((ONES & 31) >> 3) + (ONES & 3);
  static constexpr int ones_g = ((ONES & 31) >> 2) & 1;
};

template <int KSA, int KSB, bool RAWA>
__device__ __forceinline__ void wgrad_layer(const float (&av)[KSA], const float (&azd)[KSA],
                                            const float (&bv)[KSB], const float (&bt)[KSB], float* TA,
                                            float* TB, const LaneC& lc, int wave, f32x16& acc) {
  using W = WG<KSA, KSB>;
  const int t = wave % W::NT, sidx = wave / W::NT;
  const int m = t / W::NTB, n = t % W::NTB;
  const int rdA = (32 * m + lc.c) * TSW + sidx * W::PTS + lc.g * (W::PTS / 2);
  const int rdB = (32 * n + lc.c) * TSW + sidx * W::PTS + lc.g * (W::PTS / 2);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) {
      float v;
      if (half == 0) v = av[ks];
      else if (RAWA) v = azd[ks];
      else {
        const float x = opaque(av[ks]);
        v = x * (1.f - x) * azd[ks];
      }
      TA[lc.twr + vpos(ks, 0) * TSW] = v;
    }
    if (lc.g == 0) TA[lc.twr + W::ONES * TSW] = (half == 0) ? 1.f : 0.f;
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) TB[lc.twr + vpos(ks, 0) * TSW] = (half == 0) ? bv[ks] : bt[ks];
    __syncthreads();
    // operand fragments are fetched one step (4 MFMAs = 256 cycles) ahead of their use
    f32x4 a4 = *reinterpret_cast<const f32x4a*>(&TA[rdA]);
    f32x4 b4 = *reinterpret_cast<const f32x4a*>(&TB[rdB]);
#pragma unroll
    for (int j = 0; j < W::PTS / 8; ++j) {
      f32x4 an = a4, bn = b4;
      if (j + 1 < W::PTS / 8) {
        an = *reinterpret_cast<const f32x4a*>(&TA[rdA + 4 * (j + 1)]);
        bn = *reinterpret_cast<const f32x4a*>(&TB[rdB + 4 * (j + 1)]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma32(a4[e], b4[e], acc);
      __builtin_amdgcn_sched_barrier(0);
      a4 = an;
      b4 = bn;
    }
    __syncthreads();
  }
}

// After the tile loop: add this wave's accumulator tile of one layer into the LDS gradient image
// block [2*KSA+1 rows][GS] (row = in-feature, last row = bias; column = out-feature).
template <int KSA, int KSB, int GS>
__device__ __forceinline__ void wgrad_flush(const f32x16& acc, float* Gl, const LaneC& lc, int wave) {
  using W = WG<KSA, KSB>;
  const int t = wave % W::NT;
  const int m = t / W::NTB, n = t % W::NTB;
  const int cpos = 32 * n + lc.c;
  const bool colok = (vks(cpos) < KSB) && (vfeat(cpos) < GS);
  const int col = vfeat(cpos);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ks = 16 * m + i;
    int row = -1;
    if (ks < KSA) row = 2 * ks + lc.g;
    else if (m == W::ones_m && i == W::ones_i && lc.g == W::ones_g) row = 2 * KSA;
    if (row >= 0 && colok) Gl[row * GS + col] += acc[i];
  }
}

#ifdef VN_STAMPS
#define STAMP(i)                                                          \
  do {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                    \
    unsigned long long t_;                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    stamp_acc[i] += t_ - stamp_prev;                                      \
    stamp_prev = t_;                                                      \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

struct VnFusedArgsD {
  VnNet net;
  const float* theta;
  const float* X; const float* G; const float* src;
  long nT, n_k; int integ_num;
  const float* feN; const float* fedNt; const float* feW;
  const float* Nrow; const float* dNtrow;
  const float* detJv; float detJ; int time_dependent;
  float* lossVec;
  const float* Xb; const float* label; long nB, bDof; float biDimVal;
  float w0, w1, w2;
  float* partial;
  float* losspart;
  unsigned long long* stamps;
};

template <int L, int KS>
__global__ __launch_bounds__(NTHREADS, 1) void vn_fused_kernel(VnFusedArgsD A) {
  using LY = Lay<L, KS>;
  constexpr int MT = mtiles(KS);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const VnNet& net = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = net.P;
  float* W1 = lds + LY::W1_OFF;
  float* WH = lds + LY::WH_OFF;
  float* BI = lds + LY::BI_OFF;
  float* WO = lds + LY::WO_OFF;
  float* sInt = lds + LY::MISC_OFF;
  float* sR = sInt + TILE;
  float* TA = lds + LY::T_OFF;
  float* TB = TA + TROWS * TSW;
  float* Gacc = lds + LY::T_OFF;                 // aliases TA|TB, used after the tile loop only

  // ------------------------------------------------------------------ prologue: LDS images
  {
    const int d_in = net.d_in, H1 = net.H[1];
    for (int i = tid; i < 8 * WS; i += NTHREADS) {
      const int k = i / WS, pos = i % WS;
      const int f = vfeat(pos & 63);
      W1[i] = (k < d_in && pos < 64 && f < H1) ? A.theta[net.woff[1] + k * H1 + f] : 0.f;
    }
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const int Hin = net.H[l - 1], Hout = net.H[l];
      float* Wl = WH + (l - 2) * LY::HPWS;
      for (int i = tid; i < LY::HP * WS; i += NTHREADS) {
        const int k = i / WS, pos = i % WS;
        const int f = vfeat(pos & 63);
        Wl[i] = (k < Hin && pos < 64 && f < Hout) ? A.theta[net.woff[l] + k * Hout + f] : 0.f;
      }
    }
    for (int i = tid; i < L * 64; i += NTHREADS) {
      const int l = i / 64 + 1, idx = i % 64;
      const int mt = idx >> 5, g = (idx >> 4) & 1, r = idx & 15;
      const int f = 2 * (16 * mt + r) + g;
      BI[i] = (16 * mt + r < KS && f < net.H[l]) ? A.theta[net.boff[l] + f] : 0.f;
    }
    for (int i = tid; i < 2 * KS; i += NTHREADS) WO[i] = (i < net.H[L]) ? A.theta[net.woff[L + 1] + i] : 0.f;
    for (int i = tid; i < LY::T_SZ; i += NTHREADS) lds[LY::T_OFF + i] = 0.f;
  }
  __syncthreads();

  LaneC lc;
  lc.g = lane >> 5;
  lc.c = lane & 31;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    lc.offF[m] = lc.g * WS + 32 * m + lc.c;
    int fin = vfeat(32 * m + lc.c);
    if (fin >= LY::HP) fin = 0;
    lc.offB[m] = fin * WS + 4 * lc.g;
  }
  lc.twr = 4 * lc.g * TSW + wave * 32 + lc.c;

  // persistent weight-gradient accumulators: one 32x32 tile per layer per wave
  f32x16 wacc[L + 1];
#pragma unroll
  for (int l = 0; l <= L; ++l)
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[l][i] = 0.f;

  const float bo = A.theta[net.boff[L + 1]];
  const int q = A.integ_num;
  const int TT = TILE / q;                                   // whole test functions per tile
  const int TPTS = TT * q;                                   // points used in an interior tile (<= TILE)
  const bool qtree = (TILE % q) == 0;                        // q divides the tile: shuffle-tree R_k
  const long ntiles_i = (A.n_k + TT - 1) / TT;
  const long ntiles = ntiles_i + (A.nB + TILE - 1) / TILE;
  float loss_var = 0.f, loss_bc = 0.f, loss_ic = 0.f;
  const long nI = A.nB - A.bDof;
  const float cb = A.bDof > 0 ? 2.f * A.w0 * A.biDimVal / (float)A.bDof : 0.f;
  const float ci = nI > 0 ? 2.f * A.w1 * A.biDimVal / (float)nI : 0.f;

  // the quadrature index of this lane's point is the same in every interior tile (tiles start at
  // whole test functions), so the periodic FE table entries are lane constants
  const int pq_l = (wave * 32 + lc.c) % q;
  const float tab_dnt = A.time_dependent ? A.fedNt[pq_l] : 0.f;
  const float tab_w = A.feW ? A.feW[pq_l] : 1.f;
  const float tab_N = A.feN[pq_l];
#ifdef VN_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // The weight images never change inside the loop, so LICM would hoist every fragment load
    // (~450 registers) out of it and spill them; fragments are meant to be re-read from LDS.
    asm volatile("" ::: "memory");
    const bool interior = tile < ntiles_i;
    const long r0 = interior ? tile * TPTS : (tile - ntiles_i) * TILE;
    const long nrows = interior ? A.nT : A.nB;
    const int pt = wave * 32 + lc.c;                         // point inside the tile
    const long row = r0 + pt;
    const bool valid = row < nrows && (!interior || pt < TPTS);

    // ---------------------------------------------------------------- inputs
    float xin[KS0], gin[KS0];
    {
      const float* Xp = interior ? A.X : A.Xb;
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        const int f = 2 * s + lc.g;
        xin[s] = (valid && f < net.d_in) ? Xp[row * net.d_in + f] : 0.f;
        gin[s] = (valid && interior && f < net.dim) ? A.G[row * net.dim + f] : 0.f;
      }
    }

    STAMP(0);   // inputs
    float a[L][KS], zd[L][KS];                               // stored activations (registers)

    // ---------------------------------------------------------------- forward
    // layer 1 (also re-run late in the reverse pass: its activations are cheap to recompute --
    // 2*MT MFMAs per input pair -- and not keeping them alive frees 2*KS registers)
    auto layer1 = [&](const float (&xi)[KS0], const float (&gi)[KS0]) {
      f32x16 accv[MT], acct[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b = *reinterpret_cast<const f32x4a*>(&BI[m * 32 + lc.g * 16 + 4 * j]);
          accv[m][4 * j + 0] = b[0]; accv[m][4 * j + 1] = b[1];
          accv[m][4 * j + 2] = b[2]; accv[m][4 * j + 3] = b[3];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acct[m][i] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        if (2 * s < net.d_in) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float wf = W1[2 * s * WS + lc.offF[m]];
            accv[m] = mfma32(wf, xi[s], accv[m]);
            acct[m] = mfma32(wf, gi[s], acct[m]);
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a[0][ks] = fsigmoid(accv[ks >> 4][ks & 15]);
        zd[0][ks] = acct[ks >> 4][ks & 15];
      }
    };
    // forward proper: pv/pt hold the raw (z, zdot) accumulators of the previous layer; its sigmoid
    // is applied inside the next layer's k-loop, one k-step ahead of the MFMAs that consume it,
    // so the VALU/transcendental work runs under the matrix pipe.
    f32x16 pv[MT], ptn[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4a*>(&BI[m * 32 + lc.g * 16 + 4 * j]);
        pv[m][4 * j + 0] = b[0]; pv[m][4 * j + 1] = b[1];
        pv[m][4 * j + 2] = b[2]; pv[m][4 * j + 3] = b[3];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) ptn[m][i] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      if (2 * s < net.d_in) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float wf = W1[2 * s * WS + lc.offF[m]];
          pv[m] = mfma32(wf, xin[s], pv[m]);
          ptn[m] = mfma32(wf, gin[s], ptn[m]);
        }
      }
    }
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const float* Wl = WH + (l - 2) * LY::HPWS;
      f32x16 nv[MT], nt[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + m * 32 + lc.g * 16 + 4 * j]);
          nv[m][4 * j + 0] = b[0]; nv[m][4 * j + 1] = b[1];
          nv[m][4 * j + 2] = b[2]; nv[m][4 * j + 3] = b[3];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) nt[m][i] = 0.f;
      }
      // Software pipeline over k-steps (all indices compile-time): while the MFMAs of k-step ks run,
      // the sigmoid of the previous layer advances by one stage for each of the next three k-steps
      //   A(j): e = 2^(-z_j*log2e)     B(j): s = 1/(1+e)     C(j): q = s(1-s)*zdot_j
      // so no VALU/transcendental dependency chain is longer than 3 ops inside one MFMA shadow.
      auto zin = [&](int j) { return pv[j >> 4][j & 15]; };
      auto zdin = [&](int j) { return ptn[j >> 4][j & 15]; };
      auto stA = [&](int j) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * zin(j)); };
      auto stB = [&](float e) { return __builtin_amdgcn_rcpf(1.0f + e); };
      float wf[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) wf[m] = Wl[lc.offF[m]];
      float cs = stB(stA(0));                       // sigmoid of k-step 0
      float cq = cs * (1.f - cs) * zdin(0);
      a[l - 2][0] = cs;
      zd[l - 2][0] = zdin(0);
      float s1 = (KS > 1) ? stB(stA(1)) : 0.f;      // sigmoid of k-step ks+1
      float e2 = (KS > 2) ? stA(2) : 0.f;           // exp of k-step ks+2
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wn[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) wn[m] = (ks + 1 < KS) ? Wl[2 * (ks + 1) * WS + lc.offF[m]] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        nv[0] = mfma32(wf[0], cs, nv[0]);
        float e3 = 0.f;
        if (ks + 3 < KS) e3 = stA(ks + 3);
        __builtin_amdgcn_sched_barrier(0);
        nt[0] = mfma32(wf[0], cq, nt[0]);
        float s2 = 0.f;
        if (ks + 2 < KS) s2 = stB(e2);
        __builtin_amdgcn_sched_barrier(0);
        float q1 = 0.f;
        if (MT == 2) nv[MT - 1] = mfma32(wf[MT - 1], cs, nv[MT - 1]);
        if (ks + 1 < KS) {
          const float zz = zdin(ks + 1);
          q1 = s1 * (1.f - s1) * zz;
          a[l - 2][ks + 1] = s1;
          zd[l - 2][ks + 1] = zz;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MT == 2) nt[MT - 1] = mfma32(wf[MT - 1], cq, nt[MT - 1]);
        __builtin_amdgcn_sched_barrier(0);
        cs = s1; cq = q1; s1 = s2; e2 = e3;
#pragma unroll
        for (int m = 0; m < MT; ++m) wf[m] = wn[m];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) { pv[m] = nv[m]; ptn[m] = nt[m]; }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a[L - 1][ks] = fsigmoid(pv[ks >> 4][ks & 15]);
      zd[L - 1][ks] = ptn[ks >> 4][ks & 15];
    }
    STAMP(1);   // forward GEMMs
    // output layer (VALU): u, udot; both lane halves end with the full sums
    float u = 0.f, ud = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float wv = WO[2 * ks + lc.g];
      const float av = a[L - 1][ks];
      u += wv * av;
      ud += wv * (av * (1.f - av) * zd[L - 1][ks]);
    }
    u += __shfl_xor(u, 32, 64);
    ud += __shfl_xor(ud, 32, 64);
    u += bo;

    // ---------------------------------------------------------------- weak-form epilogue
    float ubar = 0.f, udbar = 0.f;
    if (interior) {
      // per-row tables (non-uniform supports, VarNetUtility.py:506-523) override the periodic ones
      const float dnt = !A.time_dependent ? 0.f : (A.dNtrow ? (valid ? A.dNtrow[row] : 0.f) : tab_dnt);
      const float wq = tab_w;
      float t = ud - dnt * u;                                           // TFModel.py:653-655
      if (A.src) t -= (valid ? A.src[row] : 0.f) * (A.Nrow ? (valid ? A.Nrow[row] : 0.f) : tab_N);           // :657
      t *= wq;                                                          // :660
      if (!valid) t = 0.f;
      // R_k = sum over the test function's q quadrature points: xor-shuffle tree inside the wave
      // (segments of min(q,32) lanes), then q/32 wave partials through LDS when q > 32.
      const int seg = q < 32 ? q : 32;
      if (qtree) {
        for (int o = 1; o < seg; o <<= 1) t += __shfl_xor(t, o, 64);
        if (lc.g == 0 && (lc.c % seg) == 0) sInt[pt / seg] = t;
      } else if (lc.g == 0) {
        sInt[pt] = t;                                                 // q does not divide the tile: serial sum
      }
      __syncthreads();
      if (tid < TT) {
        float R = 0.f;
        if (qtree) {
          const int per = q / seg;                                    // partials per test function
          for (int j = 0; j < per; ++j) R += sInt[tid * per + j];     // :661
        } else {
          for (int p = 0; p < q; ++p) R += sInt[tid * q + p];
        }
        const long k = r0 / q + tid;
        float s = 0.f;
        if (k < A.n_k) {
          const float dj = A.detJv ? A.detJv[k] : A.detJ;
          const float lv = dj * R * R;                                  // detJ once: :571-577, :664-668
          loss_var += lv;
          if (A.lossVec) A.lossVec[k] = lv;
          s = 2.f * A.w2 * dj * R;
        }
        sR[tid] = s;
      }
      __syncthreads();
      const float s = (pt < TPTS ? sR[pt / q] : 0.f) * wq;
      udbar = s;
      ubar = -dnt * s;
    } else {
      if (valid) {
        const float e = u - A.label[row];
        const bool isbc = row < A.bDof;
        if (lc.g == 0) {
          const float e2 = A.biDimVal * e * e;                          // :643
          if (isbc) loss_bc += e2; else loss_ic += e2;
        }
        ubar = (isbc ? cb : ci) * e;
      }
    }

    STAMP(2);   // output + weak-form epilogue (2 barriers)
    // ---------------------------------------------------------------- backward
    float zb[KS], zdb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float wv = WO[2 * ks + lc.g];
      const float av = opaque(a[L - 1][ks]);
      const float sp = av * (1.f - av);
      const float ab = ubar * wv, adb = udbar * wv;
      zdb[ks] = adb * sp;
      zb[ks] = ab * sp + adb * sp * (1.f - 2.f * av) * zd[L - 1][ks];
    }
    // output layer gradient: G[f][0] = sum_pt ubar*a_L + udbar*adot_L ; bias = sum ubar
    {
      float sv[1], st[1];
      sv[0] = (lc.g == 0) ? ubar : 0.f;
      st[0] = (lc.g == 0) ? udbar : 0.f;
      STAMP(3);  // zbar_L
      wgrad_layer<KS, 1, false>(a[L - 1], zd[L - 1], sv, st, TA, TB, lc, wave, wacc[L]);
      STAMP(4);  // output-layer wgrad
    }
#pragma unroll
    for (int l = L; l >= 2; --l) {
      if (l == 2 && L > 2) {                                 // bring layer-1 activations back
        float xr[KS0], gr[KS0];
#pragma unroll
        for (int s = 0; s < KS0; ++s) { xr[s] = opaque(xin[s]); gr[s] = opaque(gin[s]); }
        layer1(xr, gr);
      }
      wgrad_layer<KS, KS, false>(a[l - 2], zd[l - 2], zb, zdb, TA, TB, lc, wave, wacc[l - 1]);
      STAMP(5);  // hidden wgrad
      // input gradient of layer l: Abar_{l-1}[in pos][pt] = sum_out W_l[in][out] zbar_l[out][pt]
      const float* Wl = WH + (l - 2) * LY::HPWS;
      f32x16 accv[MT], acct[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accv[m][i] = 0.f; acct[m][i] = 0.f; }
      float wf[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) wf[m] = Wl[lc.offB[m] + vpos(0, 0)];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wn[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) wn[m] = (ks + 1 < KS) ? Wl[lc.offB[m] + vpos(ks + 1, 0)] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          accv[m] = mfma32(wf[m], zb[ks], accv[m]);
          acct[m] = mfma32(wf[m], zdb[ks], acct[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) wf[m] = wn[m];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float av = opaque(a[l - 2][ks]);
        const float sp = av * (1.f - av);
        const float ab = accv[ks >> 4][ks & 15], adb = acct[ks >> 4][ks & 15];
        zdb[ks] = adb * sp;
        zb[ks] = ab * sp + adb * sp * (1.f - 2.f * av) * zd[l - 2][ks];
      }
      STAMP(6);  // input-gradient GEMM + zbar
    }
    wgrad_layer<KS0, KS, true>(xin, gin, zb, zdb, TA, TB, lc, wave, wacc[0]);
    STAMP(7);    // layer-1 wgrad
  }

  // ------------------------------------------------------------------ epilogue
#ifdef VN_STAMPS
  if (A.stamps && blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) A.stamps[i] = stamp_acc[i];
#endif
  __syncthreads();
  for (int i = tid; i < LY::T_SZ; i += NTHREADS) lds[LY::T_OFF + i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {                      // fixed order: bitwise reproducible sums
    if (wave == w) {
      wgrad_flush<KS0, KS, LY::HP>(wacc[0], Gacc, lc, wave);
#pragma unroll
      for (int l = 2; l <= L; ++l)
        wgrad_flush<KS, KS, LY::HP>(wacc[l - 1], Gacc + LY::G1_SZ + (l - 2) * LY::GH_SZ, lc, wave);
      wgrad_flush<KS, 1, 1>(wacc[L], Gacc + LY::GO_OFF, lc, wave);
    }
    __syncthreads();
  }
  float* out = A.partial + (long)blockIdx.x * P;
  // gradient image -> flat parameter layout ([Hin+1][Hout] per layer: kernel rows then the bias row)
#pragma unroll
  for (int l = 1; l <= L + 1; ++l) {
    const int Hin = net.H[l - 1], Hout = net.H[l];
    const int gs = (l == L + 1) ? 1 : LY::HP;
    const int brow = (l == 1) ? 2 * KS0 : LY::HP;
    const float* Gl = Gacc + ((l == 1) ? 0 : (l == L + 1) ? LY::GO_OFF : LY::G1_SZ + (l - 2) * LY::GH_SZ);
    for (int i = tid; i < (Hin + 1) * Hout; i += NTHREADS) {
      const int r = i / Hout, cc = i % Hout;
      out[net.woff[l] + i] = Gl[(r < Hin ? r : brow) * gs + cc];
    }
  }
  // loss partials: (var, bc, ic)
  float v0 = loss_var, v1 = loss_bc, v2 = loss_ic;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v0 += __shfl_down(v0, o, 64);
    v1 += __shfl_down(v1, o, 64);
    v2 += __shfl_down(v2, o, 64);
  }
  __syncthreads();
  if (lane == 0) { sInt[wave * 3 + 0] = v0; sInt[wave * 3 + 1] = v1; sInt[wave * 3 + 2] = v2; }
  __syncthreads();
  if (tid < 3) A.losspart[blockIdx.x * 3 + tid] = sInt[tid] + sInt[3 + tid] + sInt[6 + tid] + sInt[9 + tid];
}

template <int L, int KS>
hipError_t launch_one(const VnFusedArgsD& a, int grid, hipStream_t s) {
  using LY = Lay<L, KS>;
  const size_t bytes = (size_t)LY::TOTAL * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)vn_fused_kernel<L, KS>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((vn_fused_kernel<L, KS>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

template <int L, int KS>
size_t lds_one(int P) {
  (void)P;
  return (size_t)Lay<L, KS>::TOTAL * sizeof(float);
}

int pick_ks(int hmax) {
  if (hmax <= 20) return 10;
  if (hmax <= 32) return 16;
  if (hmax <= 50) return 25;
  return 0;
}

}  // namespace

#define VN_FUSED_CASES(X) \
  X(1, 10) X(2, 10) X(3, 10) X(4, 10) \
  X(2, 16) X(3, 16) X(4, 16)          \
  X(3, 25) X(4, 25) X(5, 25)

size_t vn_fused_lds_bytes(const VnNet& net) {
  const int ks = pick_ks(net.hmax);
#define X(LL, KK) if (net.L == LL && ks == KK) return lds_one<LL, KK>(net.P);
  VN_FUSED_CASES(X)
#undef X
  return 0;
}

bool vn_fused_supported(const VnNet& net, int integ_num) {
  if (net.act != VN_ACT_SIGMOID) return false;            // tanh: 8-wave kernel or generic path
  if (net.d_in > 2 * KS0) return false;
  if (integ_num < 1 || integ_num > TILE) return false;   // whole test functions must fit a tile
  const size_t b = vn_fused_lds_bytes(net);
  return b != 0 && b <= 160 * 1024;
}

hipError_t vn_fused_launch(const VnFusedArgs& h, int grid, hipStream_t s) {
  VnFusedArgsD a;
  a.net = h.net; a.theta = h.theta; a.X = h.X; a.G = h.G; a.src = h.src; a.nT = h.nT; a.n_k = h.n_k;
  a.integ_num = h.integ_num; a.feN = h.feN; a.fedNt = h.fedNt; a.feW = h.feW; a.Nrow = h.Nrow; a.dNtrow = h.dNtrow; a.detJv = h.detJv;
  a.detJ = h.detJ; a.time_dependent = h.time_dependent; a.lossVec = h.lossVec; a.Xb = h.Xb;
  a.label = h.label; a.nB = h.nB; a.bDof = h.bDof; a.biDimVal = h.biDimVal; a.w0 = h.w0; a.w1 = h.w1;
  a.w2 = h.w2; a.partial = h.partial; a.losspart = h.losspart; a.stamps = h.stamps;
  const int ks = pick_ks(h.net.hmax);
#define X(LL, KK) if (h.net.L == LL && ks == KK) return launch_one<LL, KK>(a, grid, s);
  VN_FUSED_CASES(X)
#undef X
  return hipErrorInvalidValue;
}
