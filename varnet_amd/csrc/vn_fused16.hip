// Fused gfx950 kernel, 8-wave geometry: the same algorithm as vn_fused.hip (forward with one
// tangent, weak-form epilogue, full reverse pass in one persistent launch), two waves per SIMD.
//
// Cost model that shaped it (measured, tools/micro/*.hip): on gfx950 the f32 MFMA and the
// f32 VALU share one datapath -- kernel time ~ sum of MFMA cycles + sum of vector-instruction issue
// cycles + exposed waits, whatever the interleaving; a second wave per SIMD only hides latencies.  So the
// kernel minimises matrix cycles (padding), vector instructions and barrier phases, not "overlap".
//
// Geometry: workgroup = 8 waves (512 threads), 128-point tiles, 16 points per wave,
// v_mfma_f32_16x16x4_f32.  Feature f lives in k-step ks = f/4, lane group g = f%4 (lane = 16g+c,
// c = point), accumulator row ("position") pos(ks,g) = 16(ks>>2) + 4g + (ks&3).  Layers are chained in
// registers (the accumulator tile of layer l is the B operand of layer l+1); weights sit in LDS images
// with row stride 65.
//   * H = 50 (KS = 13): rows 48, 49 of every GEMM are accumulated on the VALU instead of a 4th MFMA tile.
//   * Weight gradients: operands are transposed through LDS images; 50-wide hidden layers use lane-major
//     images written with ds_write_addtid_b32, 3x3 core tiles + two v_mfma_f32_4x4x1 border jobs (H13);
//     output and (d_in <= 3) input layer are contracted per wave with 4x4x1 MFMAs, no workgroup barrier;
//     other shapes use the generic cooperative tiles (wgrad_layer).
//   * Accumulators persist across tiles in registers, those of up to three hidden layers in an LDS stash
//     (registers are short while the forward pass stores 2*KS*L activation values; scratch spills would go
//     through L2 to HBM).  Fixed summation order everywhere: results are bitwise reproducible.
// Register budget: 256 VGPRs per wave (stored activations 2*KS*L = 130 at 5x50; 18 spilled at 5x50).
#include "vn_internal.h"
#include "vn_fused16_common.h"

#include <atomic>


namespace {
using namespace vn16;

constexpr int TSW = 132;      // transposition image row stride
// Rows of a transposition image.  Positions are accumulator rows 16(ks>>2)+4g+(ks&3); a 50-wide layer
// (KS == 13) uses positions 0..47, 48 (feature 48), 52 (feature 49), 49 (bias row) and one all-zero row,
// so its images are packed into 52 rows (position 52 -> row 51, every unused position -> the zero row 50):
// the 12 KB this frees hold another layer's accumulators (Lay::NST).
// Nets up to 32 wide (KS <= 8) use positions 0..31 only: their images keep 32 rows + one all-zero row (33).
__host__ __device__ constexpr int t_rows(int KS) { return KS == 13 ? 52 : KS <= 8 ? 33 : 64; }
template <int KS>
__device__ __forceinline__ int trow(int pos) {
  if (KS == 13) return pos < 50 ? pos : pos == 52 ? 51 : 50;
  if (KS <= 8) return pos < 32 ? pos : 32;
  return pos;
}
// Position of the constant-one row that carries the bias gradient: the first position past the layer's k-steps.
// KS == 16 (widths 51..64) has no position to spare, so the row of feature 63 is used where the layer does not have
// that feature; behind a 64-wide layer the bias gradient is summed by thin_bias instead (ones_row = false).
// KS == 8 (widths 21..32) likewise: the first spare position would be 32, i.e. a third 16-row tile -- 4 x 2 weight-gradient
// tiles for a gradient that fits 2 x 2 (the [10,20,30] net of Operator_1DtMOR.py:189 spent a quarter of its matrix work
// there) -- so the bias row rides at the position of feature 31 unless the layer is exactly 32 wide.
__host__ __device__ constexpr bool fullpos(int KS) { return KS == 16 || KS == 8; }      // 4*KS fills whole 16-row tiles
__host__ __device__ constexpr int vones(int KS) {
  if (fullpos(KS)) return 4 * KS - 1;
  for (int p = 0; p < 64; ++p)
    if (vks(p) >= KS) return p;
  return -1;
}
static_assert(vfeat(63) == 63 && vfeat(31) == 31 && vpos(7, 3) == 31, "position 4*KS-1 is feature 4*KS-1");
#ifndef VN_MERGED_ROUNDS
#define VN_MERGED_ROUNDS 1
#endif
__host__ __device__ constexpr bool merged_rounds(int L, int KS) { return VN_MERGED_ROUNDS && KS <= 8 && L >= 2; }   // one publish/contract round per hidden layer

template <int L, int KS>
struct Lay {
  static constexpr int HP = 4 * KS;
  static constexpr int HPWS = al4(HP * WS);
  static constexpr int W1_OFF = 0;                          // [8][WS]
  static constexpr int WH_OFF = al4(8 * WS);                // [L-1][HP][WS]
  static constexpr int BI_OFF = WH_OFF + (L - 1) * HPWS;    // [L][64] biases in (tile, g, i) order
  static constexpr int WO_OFF = BI_OFF + L * 64;            // [4*KS]
  static constexpr int MISC_OFF = WO_OFF + al4(4 * KS);     // sInt[128] (+128 spare)
  static constexpr int T_OFF = MISC_OFF + 256;              // TA | TB
  static constexpr int G1_SZ = (4 * KS0 + 1) * HP;
  static constexpr int GH_SZ = (HP + 1) * HP;
  static constexpr int GO_OFF = G1_SZ + (L - 1) * GH_SZ;
  static constexpr int G_SZ = al4(GO_OFF + HP + 1);
  static constexpr bool MERGE = merged_rounds(L, KS);
  static constexpr int T_IMG = (MERGE ? 2 : 1) * 2 * t_rows(KS) * TSW;     // TA | TB ( | TA' | TB' : wgrad_layer)
  static constexpr int T_H13 = (KS == 13) ? 27 * (NW * 64 + 4) : 0;         // lane-major images of the 50-wide path (H13)
  static constexpr int T_MAX = T_IMG > T_H13 ? T_IMG : T_H13;
  // The gradient image of the final flush normally shares the transposition region.  Where it is larger than that
  // region AND the layout would then exceed the 160 KB (six 64-wide layers), it is laid over the weight images
  // instead, which are dead once the tile loop has ended (G_LOW; it must end below sInt, which the flush still uses).
  static constexpr int T_SZ_SHARED = al4(T_MAX > G_SZ ? T_MAX : G_SZ);
  static constexpr bool G_LOW = (T_OFF + T_SZ_SHARED > 160 * 256) && G_SZ <= MISC_OFF;
  static constexpr int T_SZ = G_LOW ? al4(T_MAX) : T_SZ_SHARED;
  static constexpr int G_OFF = G_LOW ? 0 : T_OFF;
  // Weight-gradient accumulators of the first NST hidden layers live in LDS between their uses
  // (2 x f32x4 per lane and layer): registers are short while the forward pass stores activations,
  // and what the compiler spills instead goes to scratch, i.e. through L2 to HBM.
  // Final flush: the thin layers' partial sums (input layer with d_in <= 3, output layer: 2 x f32x4 per lane and wave) are
  // parked in one slot per wave and added in wave order by the store phase -- no serial round per wave.  The slots lie in
  // the transposition region behind the gradient image where that leaves room, else over the weight images (dead once
  // the tile loop has ended; they end below sInt, which the flush still uses).
  static constexpr int SLOT_SZ = NW * 2 * 4 * 64;
  static constexpr bool SLOT_IN_T = G_LOW || (T_SZ - al4(G_SZ) >= SLOT_SZ);
  static constexpr int SLOT_OFF = G_LOW ? T_OFF : (SLOT_IN_T ? T_OFF + al4(G_SZ) : 0);
  static_assert(SLOT_IN_T || SLOT_SZ <= MISC_OFF, "thin-layer slots fit neither behind the gradient image nor over the weight images");
  static constexpr int ST_OFF = T_OFF + T_SZ;
  static constexpr int ST_LAYER = NW * 2 * 64 * 4;
  static constexpr int ST_FIT = (160 * 256 - ST_OFF) / ST_LAYER;
  static constexpr int NST = ((KS != 13 && KS != 16) || L < 3) ? 0 : (ST_FIT < L - 1 ? ST_FIT : L - 1);
  static constexpr int TOTAL = ST_OFF + NST * ST_LAYER;
};


struct LaneC {
  int g, c;
  int offF;        // forward A-fragment lane offset: g*WS + c            (+ 16*m + 4*ks*WS)
  int offB0;       // backward A-fragment lane offset of row tile 0: fin*WS + 4g   (+ 16m*WS + vpos(ks,0)): the
                   // in-features of row tile m are fin(c) + 16m, and 16*MTM <= HP in every instantiation
  int twr;         // transposition write offset: 4g*TSW + wave*16 + c     (+ vpos(ks,0)*TSW)
};

template <int KSA, int KSB>
struct WG {
  static constexpr int ONES = vones(KSA);
  static constexpr int MTA0 = (mtiles(KSA) > (ONES >> 4) + 1) ? mtiles(KSA) : (ONES >> 4) + 1;
  static constexpr int MTA = MTA0 == 3 ? 4 : MTA0;             // 1, 2 or 4 row tiles
  static constexpr int NTB0 = mtiles(KSB);
  static constexpr int NTB = NTB0 == 3 ? 4 : NTB0;
  static constexpr int NT = MTA * NTB;                         // output tiles (16x16): 1..16
  static constexpr int TPW = (NT >= NW) ? NT / NW : 1;         // tiles per wave (same row tile m)
  static constexpr int NS = (NT >= NW) ? 1 : NW / NT;          // point splits
  static constexpr int PTS = TILE / NS;                        // points contracted by one wave
  static constexpr int PG = PTS / 4;                           // ... by one lane group
  static constexpr int ones_m = ONES >> 4;
  static constexpr int ones_i = ONES & 3;
  static constexpr int ones_g = (ONES >> 2) & 3;
  static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8 || NT == 16, "tile count must divide the waves");
  static_assert(NT < NW || NTB % TPW == 0, "a wave's tiles must share a row tile");
};



// store of one published value: lane (c, g) owns row vpos(ks,0)+4g, column wave*16+c
template <int KS>
__device__ __forceinline__ void t_write(float* T, const LaneC& lc, int ks, float v) {
  if (KS == 13 && ks == 12) {                       // features 48, 49 -> rows 48, 51; 50, 51 are padding
    if (lc.g < 2) T[lc.twr + (48 - lc.g) * TSW] = v;
  } else {
    T[lc.twr + vpos(ks, 0) * TSW] = v;
  }
}

// Cooperative weight gradient (see vn_fused.hip): all waves publish their 16 point-columns of
// the transposed operands, then each wave contracts its output tile(s) over its share of the
// 128 points into persistent accumulators.
// ones_row: the layer's input side has a position to spare for the constant-one row that carries the bias gradient
// (always, except for a 64-wide input side at KS == 16: then the bias gradient comes from thin_bias below).
template <int KSA, int KSB, bool RAWA, bool TANH, bool MERGE, int NACC, class AV, class BV>
__device__ __forceinline__ void wgrad_layer(const AV& av, const AV& azd, const BV& bv, const BV& bt, float* TA,
                                            float* TB, const LaneC& lc, int wave, f32x4 (&acc)[NACC], bool ones_row) {
  using W = WG<KSA, KSB>;
  static_assert(NACC == W::TPW, "accumulator count");
  const int t0 = (W::NT >= NW) ? wave * W::TPW : wave % W::NT;
  const int sidx = (W::NT >= NW) ? 0 : wave / W::NT;
  const int m = t0 / W::NTB, n0 = t0 % W::NTB;
  // ds_read_b128 is served in 16-lane groups that mix two neighbouring lane groups g over 64
  // banks: with the row stride == 4 (mod 64) the 16 rows of a group are conflict-free only if
  // the point offsets of g and g^1 differ by a multiple of 64 -> g = 0,1,2,3 start at 0,64,32,96.
  const int goff = (W::PG == 32) ? 64 * (lc.g & 1) + 32 * (lc.g >> 1) : lc.g * W::PG;
  const int rdA = trow<KSA>(16 * m + lc.c) * TSW + sidx * W::PTS + goff;
  int rdBt[W::TPW];
#pragma unroll
  for (int t = 0; t < W::TPW; ++t) rdBt[t] = trow<KSB>(16 * (n0 + t) + lc.c) * TSW + sidx * W::PTS + goff;
  // Nets up to 32 wide (KS <= 8) have LDS to spare: value and tangent operands are published side by side (a second
  // image pair T2 floats on) and contracted behind ONE barrier pair per layer instead of two -- a small net's tile is a chain
  // of short phases between barriers, not matrix work (profiles/r3_small_stamps.txt).
  constexpr int T2 = MERGE ? 2 * t_rows(KSB) * TSW : 0;
  auto contract = [&](const float* TAh, const float* TBh) {
    f32x4 a4 = *reinterpret_cast<const f32x4a*>(&TAh[rdA]);
    f32x4 b4[W::TPW];
#pragma unroll
    for (int t = 0; t < W::TPW; ++t) b4[t] = *reinterpret_cast<const f32x4a*>(&TBh[rdBt[t]]);
#pragma unroll
    for (int j = 0; j < W::PG / 4; ++j) {
      f32x4 an = a4, bn[W::TPW];
#pragma unroll
      for (int t = 0; t < W::TPW; ++t) bn[t] = b4[t];
      if (j + 1 < W::PG / 4) {
        an = *reinterpret_cast<const f32x4a*>(&TAh[rdA + 4 * (j + 1)]);
#pragma unroll
        for (int t = 0; t < W::TPW; ++t)
          bn[t] = *reinterpret_cast<const f32x4a*>(&TBh[rdBt[t] + 4 * (j + 1)]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < W::TPW; ++t) acc[t] = mfma16(a4[e], b4[t][e], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
      a4 = an;
#pragma unroll
      for (int t = 0; t < W::TPW; ++t) b4[t] = bn[t];
    }
  };
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float* TAh = TA + half * T2;
    float* TBh = TB + half * T2;
    if constexpr (RAWA) {
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) t_write<KSA>(TAh, lc, ks, half == 0 ? av[ks] : azd[ks]);
    } else {
#pragma unroll
      for (int j = 0; j < PA<KSA>::NP; ++j) {            // sigma'(a) * zdot for two k-steps per packed instruction
        f32x2 v2 = av.p[j];
        if (half == 1) v2 = act_d1_2<TANH>(opaque2(av.p[j])) * azd.p[j];
        t_write<KSA>(TAh, lc, 2 * j, v2[0]);
        if (2 * j + 1 < KSA) t_write<KSA>(TAh, lc, 2 * j + 1, v2[1]);
      }
    }
    if (ones_row && lc.g == W::ones_g) TAh[lc.twr - 4 * lc.g * TSW + W::ONES * TSW] = (half == 0) ? 1.f : 0.f;
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) t_write<KSB>(TBh, lc, ks, (half == 0) ? bv[ks] : bt[ks]);
    if (MERGE && half == 0) continue;
    __syncthreads();
    if (MERGE) contract(TA, TB);
    contract(TAh, TBh);
    __syncthreads();
  }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// ---- thin layers: output layer (one column) and input layer with d_in <= 3 (rows x0..x2 + bias) ---
// Few rows (or columns) against all 64 positions is one 4x4x1 MFMA per point, so there is nothing to
// share between waves: every wave transposes its own 16 points through its own 16 columns of the
// images and contracts them at once -- no workgroup barrier, 16 short MFMAs per round instead of 16
// long ones, and the partial sums of the 8 waves meet in the fixed-order flush.
__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave complete in order; the fences stop the compiler from moving a
  // lane's loads across other lanes' stores (per-thread alias reasoning)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A side: 64 positions (lane = row), B side: rows brow(j) for j = lane & 3;  TRANSPOSED swaps the roles
template <int KS, bool A_IS_LANE>
__device__ __forceinline__ void thin_contract(const float* TA, const float* TB, int rowsel, int wave, int lane, f32x4& acc) {
  const int rdA = (A_IS_LANE ? trow<KS>(lane) : rowsel) * TSW + wave * CW;
  const int rdB = (A_IS_LANE ? rowsel : trow<KS>(lane)) * TSW + wave * CW;
  f32x4 a4[CW / 4], b4[CW / 4];
#pragma unroll
  for (int q = 0; q < CW / 4; ++q) {
    a4[q] = *reinterpret_cast<const f32x4a*>(&TA[rdA + 4 * q]);
    b4[q] = *reinterpret_cast<const f32x4a*>(&TB[rdB + 4 * q]);
  }
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < CW / 4; ++q) {
    acc = mfma4(a4[q][0], b4[q][0], acc);
    t = mfma4(a4[q][1], b4[q][1], t);
    acc = mfma4(a4[q][2], b4[q][2], acc);
    t = mfma4(a4[q][3], b4[q][3], t);
  }
  acc += t;
}

// output layer: d w_o[f] = sum_p (a[f][p] ubar[p] + adot[f][p] udbar[p]).  Both seeds are per-point scalars, so the two products
// are combined per element IN REGISTERS, c = a ubar + (sigma'(a) zdot) udbar, and ONE transposition pass contracts c against a
// column of ones (the bias row carries ubar itself: d b_o = sum_p ubar[p]) -- half the LDS round trips and 4x4x1 MFMAs of the
// two-pass form, on a stretch of the tile where the matrix pipe is idle anyway.  Lanes g == 0 publish the ones into row 0
// of TB, the other lane groups zeros into rows 4, 8, 12.
template <int KS, bool TANH, class AV>
__device__ __forceinline__ void thin_wgrad_out(const AV& av, const AV& azd, float ubar, float udbar,
                                               float* TA, float* TB, const LaneC& lc, int wave, int lane, f32x4& acc,
                                               bool ones_row) {
  using W = WG<KS, 1>;
  const f32x2 ub2 = {ubar, ubar}, ud2 = {udbar, udbar};
#pragma unroll
  for (int j = 0; j < PA<KS>::NP; ++j) {
    const f32x2 a2 = opaque2(av.p[j]);
    const f32x2 c2 = a2 * ub2 + (act_d1_2<TANH>(a2) * azd.p[j]) * ud2;
    t_write<KS>(TA, lc, 2 * j, c2[0]);
    if (2 * j + 1 < KS) t_write<KS>(TA, lc, 2 * j + 1, c2[1]);
  }
  if (ones_row && lc.g == W::ones_g) TA[lc.twr - 4 * lc.g * TSW + W::ONES * TSW] = ubar;
  TB[lc.twr] = (lc.g == 0) ? 1.f : 0.f;
  wave_lds_sync();
  thin_contract<KS, true>(TA, TB, (lane & 3) == 0 ? 0 : 4, wave, lane, acc);
  wave_lds_sync();
}

template <int KS>
__device__ __forceinline__ void thin_flush_out(const f32x4& acc, float* Gl, int lane, bool ones_row) {
  if ((lane & 3) != 0) return;                         // column j = 0 holds the products with ubar
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pos = lane + i;
    if (ones_row && pos == vones(KS)) Gl[4 * KS] += acc[i];
    else if (vks(pos) < KS) Gl[vfeat(pos)] += acc[i];
  }
}

// Bias gradient of a layer whose 64-wide input side leaves no position for the constant-one row (KS == 16 only):
// sum_p zbar[j][p] as a thin contraction of a ones row against the wave's own 16 columns of zbar -- one 4x4x1 MFMA
// per point, no workgroup barrier (the images are free between two cooperative rounds; every wave touches only its
// own columns).  bsum: this lane's position j = lane.
template <int KS, class BV>
__device__ __forceinline__ void thin_bias(const BV& bv, float* TA, float* TB, const LaneC& lc, int wave, int lane, float& bsum) {
  using W = WG<KS0, KS>;
  const int sel = lane & 3;
  const int rowsel = sel == 3 ? W::ONES : 4 * sel;
  TA[lc.twr] = 0.f;                                                               // rows 0, 4, 8, 12: unused
  if (lc.g == W::ones_g) TA[lc.twr - 4 * lc.g * TSW + W::ONES * TSW] = 1.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) t_write<KS>(TB, lc, ks, bv[ks]);
  wave_lds_sync();
  f32x4 t = {0.f, 0.f, 0.f, bsum};
  thin_contract<KS, false>(TA, TB, rowsel, wave, lane, t);
  bsum = t[3];
  wave_lds_sync();
}

// input layer, d_in <= 3: rows x0, x1, x2 (positions 0, 4, 8) and the bias row against all columns
template <int KS, class BV>
__device__ __forceinline__ void thin_wgrad_in(const float (&xv)[KS0], const float (&gv)[KS0], const BV& bv,
                                              const BV& bt, float* TA, float* TB, const LaneC& lc, int wave,
                                              int lane, f32x4& acc) {
  using W = WG<KS0, KS>;
  const int sel = lane & 3;
  const int rowsel = sel == 3 ? W::ONES : 4 * sel;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    TA[lc.twr] = (half == 0) ? xv[0] : gv[0];                                     // feature g at position 4g
    if (lc.g == W::ones_g) TA[lc.twr - 4 * lc.g * TSW + W::ONES * TSW] = (half == 0) ? 1.f : 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) t_write<KS>(TB, lc, ks, (half == 0) ? bv[ks] : bt[ks]);
    wave_lds_sync();
    thin_contract<KS, false>(TA, TB, rowsel, wave, lane, acc);
    wave_lds_sync();
  }
}

template <int KS, int GS>
__device__ __forceinline__ void thin_flush_in(const f32x4& acc, float* Gl, int lane) {
  const int col = vfeat(lane);                         // lane = 4b + j = column position
  if (vks(lane) < KS && col < GS) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Gl[i * GS + col] += acc[i];
    Gl[4 * KS0 * GS + col] += acc[3];                  // bias row of the input layer's gradient image
  }
}

// ---- 50-wide hidden layers (KS == 13) -----------------------------------------------------------
// A 51 x 50 gradient (50 inputs + bias row, 50 outputs) padded to 4 x 4 tiles of 16 x 16 wastes 38 %
// of the matrix work.  Here the 48 x 48 core is 3 x 3 tiles of v_mfma_f32_16x16x4_f32 and the two
// thin borders -- rows {48, 49, bias} x all columns and columns {48, 49} x rows 0..47 -- are
// accumulated with v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 blocks per instruction, one point
// per instruction, so a border of 4 rows x 64 columns costs 8 cycles per point instead of 32.
// Jobs (w and w+4 share a SIMD; every SIMD gets 3 tile-times instead of 4).  The matrix pipe
// alternates between the two waves of a SIMD per instruction, so an 8-cycle 4x4x1 stream paired with a
// 32-cycle 16x16x4 stream would crawl at the partner's pace: the two border jobs share a SIMD.
//   w = 0,1,2: tile (w,0) over all 128 points + tile (w,2) over one half of the points
//   w = 4,5,6: tile (w-4,1) over all points   + tile (w-4,2) over the other half
//   w = 3: row border          w = 7: column border
// so the six tile waves run one instruction stream (48 MFMAs per round) and share the A fragment.
// D layout of the 4x4x1 MFMA: lane 4b+j, register i  =  sum_p A[lane 4b+i] * B[lane 4b+j].
// Image layout of this path ("lane-major"): element (position pos, point (w, c)) of an image sits at
//   vks(pos)*RS + w*64 + g(pos)*16 + c          (g(pos) = (pos>>2)&3, RS = 8*64 + 4 floats)
// i.e. one image row per k-step holds, per wave, the 64 lanes of that wave's register in lane order.  A
// publishing store is then `ds_write_addtid_b32` (address = M0 + offset + 4*lane, no address VGPR,
// 128 B/clk/CU instead of the 64 B/clk of ds_write_b32 -- the publish rounds are bound by the LDS store
// path), and a reader still finds 4 consecutive points of one feature in one ds_read_b128.
// TA: rows 0..12 = k-steps, row 13 = [bias ones | zeros | zeros | zeros]; TB: rows 0..12.
struct H13 {
  static constexpr int RS = NW * 64 + 4;
  static constexpr int TA_ROWS = 14, TB_ROWS = 13;
  static constexpr int P48 = vpos(12, 0), P49 = vpos(12, 1);          // positions of features 48, 49
  static constexpr int ONES = vones(13);                              // position (13, g=0): the bias row
  static constexpr int ZERO = ONES + 4;                               // position (13, g=1): always zero (TA only)
  __host__ __device__ static constexpr int off(int pos) { return vks(pos) * RS + ((pos >> 2) & 3) * 16; }
  __device__ static __forceinline__ int edge_row(int i) { return i == 0 ? P48 : i == 1 ? P49 : i == 2 ? ONES : ZERO; }
  // role 0: core tiles; 1: row border; 2: column border
  __device__ static __forceinline__ int role(int wave) { return wave == 3 ? 1 : wave == 7 ? 2 : 0; }
  __device__ static __forceinline__ int tile_m(int wave) { return wave & 3; }
  __device__ static __forceinline__ int tile_n(int wave) { return wave >= 4 ? 1 : 0; }
};
static_assert(H13::ZERO == 53 && vks(H13::ZERO) == 13, "zero position");
static_assert(Lay<5, 13>::T_H13 == (H13::TA_ROWS + H13::TB_ROWS) * H13::RS, "image size");

// ds_write_addtid_b32: LDS[M0 + OFF + 4*lane] = v.  M0 is written once per publishing round (the wait
// state is the documented SALU-writes-M0 -> add-TID hazard); nothing else in this kernel uses M0.
__device__ __forceinline__ void addtid_base(unsigned m0_bytes) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0_bytes) : "memory");
}
template <int OFF_BYTES>
__device__ __forceinline__ void addtid_store(float v) {
  static_assert(OFF_BYTES >= 0 && OFF_BYTES < 65536, "16-bit offset");
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(OFF_BYTES) : "memory");
#endif
}
__device__ __forceinline__ void addtid_drain() {     // the compiler does not count these stores in lgkmcnt
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void h13_contract(const float* TA, const float* TB, const LaneC& lc, int wave, int lane,
                                             f32x4 (&acc)[2]) {
  const int role = H13::role(wave);
  if (role == 1 || role == 2) {
    // border job: every lane walks all 128 points, 4 per ds_read_b128; two accumulators alternate
    const int sel = lane & 3;
    // offsets relative to TA (TB = TA + TBO): one LDS base pointer, integer selects only
    constexpr int TBO = H13::TA_ROWS * H13::RS;
    const int lane_off = vks(lane) * H13::RS + ((lane >> 2) & 3) * 16;        // lane = position
    const int oA = (role == 1) ? H13::off(H13::edge_row(sel)) : lane_off;
    const int oB = (role == 1) ? TBO + lane_off
                               : (sel == 0 ? TBO + H13::off(H13::P48) : sel == 1 ? TBO + H13::off(H13::P49) : H13::off(H13::ZERO));
    const float* pA = TA + oA;
    const float* pB = TA + oB;
    // 8 points per stage (wave w's columns 0..7 or 8..15), two register buffers used alternately (no
    // rotation copies: a stage's loads are waited for one stage later, behind 16 MFMAs)
    constexpr int NSTG = TILE / 8;
    f32x4 a0[2], b0[2], a1[2], b1[2];
    auto load = [&](f32x4 (&a4)[2], f32x4 (&b4)[2], int stage) {
      const int o = (stage >> 1) * 64 + (stage & 1) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a4[h] = *reinterpret_cast<const f32x4a*>(&pA[o + 4 * h]);
        b4[h] = *reinterpret_cast<const f32x4a*>(&pB[o + 4 * h]);
      }
    };
    auto compute = [&](const f32x4 (&a4)[2], const f32x4 (&b4)[2]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = mfma4(a4[0][e], b4[0][e], acc[0]);
        acc[1] = mfma4(a4[1][e], b4[1][e], acc[1]);
      }
    };
    load(a0, b0, 0);
#pragma unroll 1
    for (int q = 0; q < NSTG; q += 2) {
      load(a1, b1, q + 1);
      compute(a0, b0);
      if (q + 2 < NSTG) load(a0, b0, q + 2);
      compute(a1, b1);
    }
    return;
  }
  // core tiles: lane (c, g) reads feature position 16m+c; lane group g contracts the points of waves
  // 2g and 2g+1 (so g and g^1 are 128 floats apart: conflict-free ds_read_b128, rows 4 banks apart)
  const int m = H13::tile_m(wave), n0 = H13::tile_n(wave);
  const int fo = (lc.c & 3) * H13::RS + (lc.c >> 2) * 16 + lc.g * 128;     // + 4m*RS for tile row m
  const int rdA = 4 * m * H13::RS + fo;
  const int rdB = 4 * n0 * H13::RS + fo;
  const int rdC = 4 * 2 * H13::RS + fo;
  // the shared tile (m,2) is contracted over one of the lane group's two waves only
  const int base1 = (wave >= NW / 2) ? 64 : 0, base2 = 64 - base1;
  f32x4 a4 = *reinterpret_cast<const f32x4a*>(&TA[rdA + base1]);
  f32x4 b4 = *reinterpret_cast<const f32x4a*>(&TB[rdB + base1]);
  f32x4 c4 = *reinterpret_cast<const f32x4a*>(&TB[rdC + base1]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 an, bn, cn = c4;
    if (j + 1 < 4) {
      an = *reinterpret_cast<const f32x4a*>(&TA[rdA + base1 + 4 * (j + 1)]);
      bn = *reinterpret_cast<const f32x4a*>(&TB[rdB + base1 + 4 * (j + 1)]);
      cn = *reinterpret_cast<const f32x4a*>(&TB[rdC + base1 + 4 * (j + 1)]);
    } else {
      an = *reinterpret_cast<const f32x4a*>(&TA[rdA + base2]);
      bn = *reinterpret_cast<const f32x4a*>(&TB[rdB + base2]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = mfma16(a4[e], b4[e], acc[0]);
      acc[1] = mfma16(a4[e], c4[e], acc[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    a4 = an; b4 = bn; c4 = cn;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 an = a4, bn = b4;
    if (j + 1 < 4) {
      an = *reinterpret_cast<const f32x4a*>(&TA[rdA + base2 + 4 * (j + 1)]);
      bn = *reinterpret_cast<const f32x4a*>(&TB[rdB + base2 + 4 * (j + 1)]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0] = mfma16(a4[e], b4[e], acc[0]);
    __builtin_amdgcn_sched_barrier(0);
    a4 = an; b4 = bn;
  }
}

template <int KS_, int I, bool TANH>
struct H13Pub {      // unrolled stores with compile-time offsets (inline-asm immediates)
  static __device__ __forceinline__ void run(int half, const PA<13>& av, const PA<13>& azd, const PA<13>& bv,
                                             const PA<13>& bt) {
    // I even: this step publishes k-steps I and I+1 (one packed sigma'(a)*zdot for both)
    f32x2 v2 = av.p[I >> 1];
    if (half == 1) v2 = act_d1_2<TANH>(opaque2(av.p[I >> 1])) * azd.p[I >> 1];
    addtid_store<I * H13::RS * 4>(v2[0]);
    addtid_store<(H13::TA_ROWS + I) * H13::RS * 4>(half == 0 ? bv[I] : bt[I]);
    if constexpr (I + 1 < KS_) {
      addtid_store<(I + 1) * H13::RS * 4>(v2[1]);
      addtid_store<(H13::TA_ROWS + I + 1) * H13::RS * 4>(half == 0 ? bv[I + 1] : bt[I + 1]);
    }
    if constexpr (I + 2 < KS_) H13Pub<KS_, I + 2, TANH>::run(half, av, azd, bv, bt);
  }
};

// both rounds of a 50-wide hidden layer; t_base_bytes = byte address of this wave's 64 columns of TA
template <bool TANH>
__device__ __forceinline__ void h13_wgrad_layer(const PA<13>& av, const PA<13>& azd, const PA<13>& bv,
                                                const PA<13>& bt, float* TA, const LaneC& lc, int wave, int lane,
                                                unsigned t_base_bytes, f32x4 (&acc)[2]) {
  const float* TB = TA + H13::TA_ROWS * H13::RS;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    addtid_base(t_base_bytes);
    H13Pub<13, 0, TANH>::run(half, av, azd, bv, bt);
    addtid_store<13 * H13::RS * 4>((half == 0 && lc.g == 0) ? 1.f : 0.f);      // bias row | zeros
    addtid_drain();
    __syncthreads();
    h13_contract(TA, TB, lc, wave, lane, acc);
    __syncthreads();
  }
}

template <int GS>
__device__ __forceinline__ void h13_flush(const f32x4 (&acc)[2], float* Gl, const LaneC& lc, int wave, int lane,
                                          int slot_lo, int slot_hi) {   // tile waves: slots [slot_lo, slot_hi)
  const int role = H13::role(wave);
  if (role == 1) {                                   // rows 48, 49, bias x every column position
    const int pos = lane;                            // 4b + j
    const int col = vfeat(pos);
    if (vks(pos) < 13 && col < GS) {
      Gl[48 * GS + col] += acc[0][0] + acc[1][0];
      Gl[49 * GS + col] += acc[0][1] + acc[1][1];
      Gl[52 * GS + col] += acc[0][2] + acc[1][2];    // bias row of the gradient image (row 4*KS)
    }
    return;
  }
  if (role == 2) {                                   // rows 0..47 x columns 48, 49
    const int j = lane & 3;
    if (j < 2 && 48 + j < GS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pos = (lane & ~3) + i;             // 4b + i
        if (pos < 48) Gl[vfeat(pos) * GS + 48 + j] += acc[0][i] + acc[1][i];
      }
    }
    return;
  }
  const int m = H13::tile_m(wave), n0 = H13::tile_n(wave);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < slot_lo || t >= slot_hi) continue;
    const int col = vfeat(16 * (t ? 2 : n0) + lc.c);           // slot 1: this wave's share of tile (m,2)
    if (col < GS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) Gl[(4 * (4 * m + i) + lc.g) * GS + col] += acc[t][i];
    }
  }
}

template <int KSA, int KSB, int GS, int NACC>
__device__ __forceinline__ void wgrad_flush(const f32x4 (&acc)[NACC], float* Gl, const LaneC& lc, int wave, bool ones_row = true) {
  using W = WG<KSA, KSB>;
  const int t0 = (W::NT >= NW) ? wave * W::TPW : wave % W::NT;
  const int m = t0 / W::NTB, n0 = t0 % W::NTB;
#pragma unroll
  for (int t = 0; t < W::TPW; ++t) {
    const int cpos = 16 * (n0 + t) + lc.c;
    const bool colok = (vks(cpos) < KSB) && (vfeat(cpos) < GS);
    const int col = vfeat(cpos);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ks = 4 * m + i;
      int row = -1;
      if (ones_row && m == W::ones_m && i == W::ones_i && lc.g == W::ones_g) row = 4 * KSA;
      else if (ks < KSA) row = 4 * ks + lc.g;
      if (row >= 0 && colok) Gl[row * GS + col] += acc[t][i];
    }
  }
}




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
  // de-duplicated formulation (rows = unique quadrature points, no test-function grouping):
  int mode;                 // 0 fused step; 1 forward only -> out_u/out_ud; 2 reverse pass with external seeds
  int dir;                  // >= 0: tangent direction is the unit vector e_dir (G ignored)
  int ostride;              // element stride of out_ud / seed_ud
  float* out_u; float* out_ud;
  const float* seed_u; const float* seed_ud;   // seed_u nullptr = 0, seed_ud nullptr = 1
};

template <int L, int KS, bool TANH>
// The machine-level load/store optimizer pairs LDS reads into ds_read2_b32, whose 8-bit offsets force a
// VALU address add per pair; vector instructions share the datapath with the f32 MFMAs here, LDS issue
// does not, so pairing is switched off for this kernel (device pass only; -0.8 % kernel time).
#if defined(__HIP_DEVICE_COMPILE__)
#define VN_NO_LDS_PAIRING __attribute__((target("no-load-store-opt")))
#else
#define VN_NO_LDS_PAIRING
#endif
__global__ __launch_bounds__(NTHREADS, 2) VN_NO_LDS_PAIRING void vn_fused16_kernel(VnFusedArgsD A) {
  using LY = Lay<L, KS>;
  constexpr int MT = mtiles(KS);
  // EDGE: the last 16-row tile holds a single k-step (features 4(KS-1) .. 4(KS-1)+3, e.g. 48,49 of a
  // 50-wide layer).  Producing those few rows with an MFMA tile costs a quarter of the matrix work of
  // the layer; instead every lane accumulates its share of their dot products on the VALU (which runs
  // under the partner wave's MFMAs) and the four lane groups are summed with two shuffles.
  constexpr bool EDGE = (KS % 4) == 1 && KS > 1;
  constexpr int MTM = EDGE ? MT - 1 : MT;            // row tiles produced by MFMA in hidden layers
  constexpr int NVE = (KS == 13) ? 2 : 4;            // edge features that can be non-padding
  constexpr int EPOS = 16 * (MT - 1);                // accumulator row of edge feature 0
  // KSKIP: nets up to 32 wide are padded to 4*KS features in EVERY layer; a small net's tile is bound by the matrix pipe like
  // any other (removing MFMAs scales the step: profiles/r3_small_sensitivity.txt), so the k-steps and row tiles that hold
  // only padding (zero weights: they add +0) are branched over, wave-uniformly, on the layer's real widths -- the [10,20,30]
  // net of Operator_1DtMOR.py:189 needs 3 and 5 of its 8 forward k-steps, and one of two row tiles in its last input gradient.
  constexpr bool KSKIP = KS <= 8;
  // (the bound is made opaque at every use: left to itself the compiler hoists the loop-invariant compares out of the tile
  // loop as 64-bit lane masks, a pair of scalar registers per guard, and spills them through v_writelane)
  auto live_k = [](int ks, int& kn) {
    if (!KSKIP || ks == 0) return true;
    asm volatile("" : "+s"(kn));
    return ks < kn;
  };
  auto live_m = [](int m, int& mn) {
    if (!KSKIP || m == 0) return true;
    asm volatile("" : "+s"(mn));
    return m < mn;
  };
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const VnNet& net = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = net.P;
  float* W1 = lds + LY::W1_OFF;
  float* WH = lds + LY::WH_OFF;
  float* BI = lds + LY::BI_OFF;
  float* WO = lds + LY::WO_OFF;
  float* sInt = lds + LY::MISC_OFF;
  float* TA = lds + LY::T_OFF;
  float* TB = TA + t_rows(KS) * TSW;
  const unsigned t_base_bytes = (unsigned)((LY::T_OFF + wave * 64) * 4);      // lane-major images: this wave's columns
  float* Gacc = lds + LY::G_OFF;

  // lane constants that come from global memory are requested first, so that their latency hides under the prologue:
  // the quadrature index of this lane's point is the same in every interior tile (tiles start at whole test functions),
  // so the periodic FE table entries are lane constants
  const int q = A.integ_num;
  const int pq_l = (wave * CW + (lane & 15)) % q;
  const int tf_l = (wave * CW + (lane & 15)) / q;    // ... and so is its test function within the tile
  const float tab_dnt = A.time_dependent ? A.fedNt[pq_l] : 0.f;
  const float tab_w = A.feW ? A.feW[pq_l] : 1.f;
  const float tab_N = A.feN[pq_l];
  const float bo = A.theta[net.boff[L + 1]];
  // ------------------------------------------------------------------ prologue: LDS images
  {
    // Parameters are read in their own (row-major) order -- consecutive lanes, consecutive floats: a destination-driven
    // walk of the images gathers 4-byte words all over a layer and costs 4 us per launch on the texture path -- and
    // scattered into the images [in-feature][out-position]; every load is issued before anything waits, the zero fill of
    // the images (padding rows / columns must be exact zeros) and of the transposition region runs under their latency.
    const int d_in = net.d_in, H1 = net.H[1];
    constexpr int NSRC = (LY::HP * LY::HP + NTHREADS - 1) / NTHREADS;
    static_assert(8 * 64 <= NTHREADS, "layer 1: one parameter per thread");
    float v1 = 0.f, vh[L > 1 ? L - 1 : 1][NSRC];
    if (tid < d_in * H1) v1 = A.theta[net.woff[1] + tid];
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const int n = net.H[l - 1] * net.H[l];
      const float* src = A.theta + net.woff[l];
#pragma unroll
      for (int it = 0; it < NSRC; ++it) {
        const int j = tid + it * NTHREADS;
        vh[l - 2][it] = j < n ? src[j] : 0.f;
      }
    }
    // biases (one per thread: L * 64 <= 512) and output weights ride in the same batch of loads
    static_assert(L * 64 <= NTHREADS && 4 * KS <= NTHREADS, "one bias / output weight per thread");
    float vb = 0.f, vo = 0.f;
    if (tid < L * 64) {
      const int l = tid / 64 + 1, idx = tid % 64;
      const int mt = idx >> 4, g = (idx >> 2) & 3, r = idx & 3;       // [tile][g][i]
      const int ks = 4 * mt + r, f = 4 * ks + g;
      vb = (ks < KS && f < net.H[l]) ? A.theta[net.boff[l] + f] : 0.f;
    }
    if (tid < 4 * KS) vo = (tid < net.H[L]) ? A.theta[net.woff[L + 1] + tid] : 0.f;
    static_assert(LY::BI_OFF % 4 == 0 && LY::T_OFF % 4 == 0 && LY::T_SZ % 4 == 0, "16-byte zero fill");
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < LY::BI_OFF / 4; i += NTHREADS) reinterpret_cast<f32x4a*>(lds)[i] = z4;                    // W1 | WH
    for (int i = tid; i < LY::T_SZ / 4; i += NTHREADS) reinterpret_cast<f32x4a*>(lds + LY::T_OFF)[i] = z4;
    __syncthreads();
    if (tid < d_in * H1) {
      const int k = tid / H1, f = tid - k * H1;
      W1[k * WS + vpos(f >> 2, f & 3)] = v1;
    }
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      float* Wl = WH + (l - 2) * LY::HPWS;
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
  __syncthreads();

  LaneC lc;
  lc.g = lane >> 4;
  lc.c = lane & 15;
  lc.offF = lc.g * WS + lc.c;
  static_assert(16 * MTM <= LY::HP, "backward fragment rows stay inside the weight image");
  static_assert(vfeat(16 + 5) == 16 + vfeat(5), "row tile m holds features fin(c) + 16m");
  lc.offB0 = vfeat(lc.c) * WS + 4 * lc.g;
  lc.twr = 4 * lc.g * TSW + wave * CW + lc.c;

  // persistent weight-gradient accumulators
  using W1G = WG<KS0, KS>;
  using WHG = WG<KS, KS>;
  constexpr bool HID13 = (KS == 13);                 // 50-wide hidden layers: 3x3 core tiles + 4x4x1 borders
  constexpr int NHACC = HID13 ? 2 : WHG::TPW;
  f32x4a* stash = reinterpret_cast<f32x4a*>(lds + LY::ST_OFF) + wave * 2 * 64 + lane;   // [layer][wave][slot][lane]
  constexpr int ST_L = LY::ST_LAYER / 4;
  f32x4 wacc1[W1G::TPW], wacch[L > 1 ? L - 1 : 1][NHACC];
  float woacc = 0.f, boacc = 0.f;                    // output layer: lane (g, c) holds d w_o[4c + g]; d b_o in every lane
#pragma unroll
  for (int t = 0; t < W1G::TPW; ++t) wacc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < (L > 1 ? L - 1 : 1); ++l)
#pragma unroll
    for (int t = 0; t < NHACC; ++t) {
      wacch[l][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (l < LY::NST) stash[l * ST_L + t * 64] = wacch[l][t];
    }

  const bool thin_in = net.d_in <= 3;                // input-layer weight gradient without workgroup barriers
  // KS == 16 with a 64-wide layer: its output side has no position left for the constant-one row of the NEXT layer's
  // weight gradient, whose bias gradient is then summed by thin_bias (hidden) / a per-lane scalar (output layer)
  bool ones_h[L > 1 ? L - 1 : 1];                    // layer l = 2..L: input side H[l-1] < 64
#pragma unroll
  for (int l = 2; l <= L; ++l) ones_h[l - 2] = !fullpos(KS) || net.H[l - 1] < 4 * KS;
  float bsum_h[L > 1 ? L - 1 : 1];
#pragma unroll
  for (int l = 0; l < (L > 1 ? L - 1 : 1); ++l) bsum_h[l] = 0.f;
  const int TT = TILE / q;                                   // whole test functions per tile
  const int TPTS = TT * q;                                   // points used in an interior tile (<= TILE)
  const bool qtree = (TILE % q) == 0;                        // q divides the tile: shuffle-tree R_k
  const long ntiles_i = A.mode ? (A.nT + TILE - 1) / TILE : (A.n_k + TT - 1) / TT;
  const long ntiles = ntiles_i + (A.mode == 1 ? 0 : (A.nB + TILE - 1) / TILE);
  float loss_var = 0.f, loss_bc = 0.f, loss_ic = 0.f;
  const long nI = A.nB - A.bDof;
  const float cb = A.bDof > 0 ? 2.f * A.w0 * A.biDimVal / (float)A.bDof : 0.f;
  const float ci = nI > 0 ? 2.f * A.w1 * A.biDimVal / (float)nI : 0.f;

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");                 // keep LDS fragment loads inside the loop
    const bool interior = tile < ntiles_i;
    const long r0 = (interior && !A.mode) ? tile * TPTS : (interior ? tile : tile - ntiles_i) * TILE;
    const long nrows = interior ? A.nT : A.nB;
    const int pt = wave * CW + lc.c;
    const long row = r0 + pt;
    const bool valid = row < nrows && (!interior || A.mode || pt < TPTS);

    float xin[KS0], gin[KS0];
    {
      const float* Xp = interior ? A.X : A.Xb;
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        const int f = 4 * s + lc.g;
        xin[s] = (valid && f < net.d_in) ? Xp[row * net.d_in + f] : 0.f;
        if (A.dir >= 0) gin[s] = (valid && interior && f == A.dir) ? 1.f : 0.f;
        else gin[s] = (valid && interior && f < net.dim) ? A.G[row * net.dim + f] : 0.f;
      }
    }

    PA<KS> a[L], zd[L];

    // ---------------------------------------------------------------- layer 1 (also recomputed late)
    auto layer1_raw = [&](const float (&xi)[KS0], const float (&gi)[KS0], f32x4 (&ov)[MT], f32x4 (&ot)[MT]) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ov[m] = *reinterpret_cast<const f32x4a*>(&BI[m * 16 + lc.g * 4]);
        ot[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        if (4 * s < net.d_in) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float wf = W1[4 * s * WS + lc.offF + 16 * m];
            ov[m] = mfma16(wf, xi[s], ov[m]);
            ot[m] = mfma16(wf, gi[s], ot[m]);
          }
        }
      }
    };

    // ---------------------------------------------------------------- forward
    f32x4 pv[MT], ptn[MT];
    layer1_raw(xin, gin, pv, ptn);
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const float* Wl = WH + (l - 2) * LY::HPWS;
      int k_in = (net.H[l - 1] + 3) >> 2, m_out = (net.H[l] + 15) >> 4;      // KSKIP: live k-steps / row tiles of this layer (scalar)
      f32x4 nv[MT], nt[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        nv[m] = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + m * 16 + lc.g * 4]);
        nt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // Activation of the previous layer, pipelined under this layer's MFMAs in PAIRS of k-steps (rows 2j, 2j+1 of an
      // accumulator tile are a register pair): stage A = packed scale + 2 v_exp, stage B = packed 1+e + 2 v_rcp,
      // stage C = packed sigma' * zdot.  Pair j is consumed by k-steps 2j and 2j+1; A(j+3) is issued after the first
      // of them, B(j+2) and C(j+1) after the second, so no transcendental chain is longer than one stage.
      // (The edge rows stay scalar FMAs: packing them -- across the edge features with splat activations, or along k
      // with weight pairs -- was measured 2.4-3.8 % SLOWER: the extra register pairs push 25 -> 48 spilled VGPRs.)
      constexpr int NP = PA<KS>::NP;
      auto zin2 = [&](int j) { return f32x2{pv[(2 * j) >> 2][(2 * j) & 3], pv[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
      auto zdin2 = [&](int j) { return f32x2{ptn[(2 * j) >> 2][(2 * j) & 3], ptn[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
      float wf[MTM], we[NVE], ev[NVE], et[NVE];
#pragma unroll
      for (int m = 0; m < MTM; ++m) wf[m] = Wl[lc.offF + 16 * m];
#pragma unroll
      for (int v = 0; v < NVE; ++v) {
        we[v] = EDGE ? Wl[lc.offF - lc.c + EPOS + 4 * v] : 0.f;
        ev[v] = 0.f;
        et[v] = 0.f;
      }
      f32x2 cs2 = act_fin2<TANH>(act_exp2<TANH>(zin2(0)));
      f32x2 cq2 = act_d1_2<TANH>(cs2) * zdin2(0);
      a[l - 2].p[0] = cs2;
      zd[l - 2].p[0] = zdin2(0);
      f32x2 s1 = (NP > 1) ? act_fin2<TANH>(act_exp2<TANH>(zin2(1))) : f32x2{0.f, 0.f};
      f32x2 e2 = (NP > 2) ? act_exp2<TANH>(zin2(2)) : f32x2{0.f, 0.f};
      f32x2 e3 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int j = ks >> 1;
        float wn[MTM], wen[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + lc.offF + 16 * m] : 0.f;
#pragma unroll
        for (int v = 0; v < NVE; ++v)
          wen[v] = (EDGE && ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + lc.offF - lc.c + EPOS + 4 * v] : 0.f;
        const float cs = cs2[ks & 1], cq = cq2[ks & 1];
        __builtin_amdgcn_sched_barrier(0);
        if (live_k(ks, k_in)) {
#pragma unroll
          for (int m = 0; m < MTM; ++m) {
            if (!live_m(m, m_out)) continue;
            nv[m] = mfma16(wf[m], cs, nv[m]);
            nt[m] = mfma16(wf[m], cq, nt[m]);
          }
        }
        if (EDGE) {
#pragma unroll
          for (int v = 0; v < NVE; ++v) { ev[v] += we[v] * cs; et[v] += we[v] * cq; }
        }
        if ((ks & 1) == 0) {
          if (j + 3 < NP) e3 = act_exp2<TANH>(zin2(j + 3));                  // stage A of pair j+3
          __builtin_amdgcn_sched_barrier(0);
        } else {
          f32x2 s2 = s1, q1 = cq2;
          if (j + 2 < NP) s2 = act_fin2<TANH>(e2);                            // stage B of pair j+2
          if (j + 1 < NP) {                                                    // stage C of pair j+1
            const f32x2 zz = zdin2(j + 1);
            q1 = act_d1_2<TANH>(s1) * zz;
            a[l - 2].p[j + 1] = s1;
            zd[l - 2].p[j + 1] = zz;
          }
          __builtin_amdgcn_sched_barrier(0);
          cs2 = s1; cq2 = q1; s1 = s2; e2 = e3;
        }
#pragma unroll
        for (int m = 0; m < MTM; ++m) wf[m] = wn[m];
#pragma unroll
        for (int v = 0; v < NVE; ++v) we[v] = wen[v];
      }
      if (EDGE) {
        // sum the four lane groups' shares; group g keeps edge feature g
        nv[MT - 1][0] += edge_reduce_scatter<NVE>(ev, lc.g);         // bias was loaded above
        nt[MT - 1][0] = edge_reduce_scatter<NVE>(et, lc.g);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) { pv[m] = nv[m]; ptn[m] = nt[m]; }
    }
    // rows (2j, 2j+1) of an accumulator tile are a register pair: pairs of k-steps per packed instruction (the
    // second half of a last, odd pair is a padding row: computed, never used)
    auto pairOf = [](const f32x4 (&t)[MT], int j) { return f32x2{t[(2 * j) >> 2][(2 * j) & 3], t[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
#pragma unroll
    for (int j = 0; j < PA<KS>::NP; ++j) {
      a[L - 1].p[j] = act_fin2<TANH>(act_exp2<TANH>(pairOf(pv, j)));
      zd[L - 1].p[j] = pairOf(ptn, j);
    }
    // output layer (VALU)
    float u = 0.f, ud = 0.f;
    {
      f32x2 u2 = {0.f, 0.f}, ud2 = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < PA<KS>::NP; ++j) {
        const bool full = 2 * j + 1 < KS;
        const f32x2 wv = {WO[4 * (2 * j) + lc.g], full ? WO[4 * (2 * j + 1) + lc.g] : 0.f};
        const f32x2 av = a[L - 1].p[j];
        u2 += wv * av;
        ud2 += wv * (act_d1_2<TANH>(av) * zd[L - 1].p[j]);
      }
      u = u2[0] + u2[1];
      ud = ud2[0] + ud2[1];
    }
    // sum over the four lane groups: row swaps (v_permlane16_swap / v_permlane32_swap, tools/micro/rowsum4_check.hip) instead of
    // two ds_bpermute round trips per value -- this stretch of the tile has no matrix work to hide an LDS latency behind;
    // same pairing as xor 16 then xor 32, so the same bits
    u = rowsum4(u);  ud = rowsum4(ud);
    u += bo;

    // ---------------------------------------------------------------- weak-form epilogue
    if (A.mode == 1) {                                       // forward only: model value and directional derivative
      if (valid && lc.g == 0) {
        if (A.out_u) A.out_u[row] = u;
        if (A.out_ud) A.out_ud[row * A.ostride] = ud;
      }
      continue;
    }
    float ubar = 0.f, udbar = 0.f;
    bool epi_barrier = false;                                // this tile's epilogue ran a workgroup barrier (uniform)
    if (interior && A.mode == 2) {                           // seeds were assembled per unique point
      if (valid) {
        ubar = A.seed_u ? A.seed_u[row] : 0.f;
        udbar = A.seed_ud ? A.seed_ud[row * A.ostride] : 1.f;   // no array: tangent seed 1 (the direction G carries the seeds)
      }
    } else if (interior) {
      // per-row tables (non-uniform supports, VarNetUtility.py:506-523) override the periodic ones
      const float dnt = !A.time_dependent ? 0.f : (A.dNtrow ? (valid ? A.dNtrow[row] : 0.f) : tab_dnt);
      const float wq = tab_w;
      float t = ud - dnt * u;
      if (A.src) t -= (valid ? A.src[row] : 0.f) * (A.Nrow ? (valid ? A.Nrow[row] : 0.f) : tab_N);
      t *= wq;
      if (!valid) t = 0.f;
      const int seg = q < CW ? q : CW;
      // A test function of up to 16 points (1D+t, two-point Gauss: q = 16) lies inside one wave: the butterfly below leaves
      // its sum R_k in every lane of the group (a + b == b + a bit for bit, so all lanes agree) -- no LDS, no workgroup barrier.
      const bool rk_in_wave = qtree && q <= CW;
      if (qtree) {
        // butterfly over the seg <= 16 lanes of a test function's points, inside a 16-lane row: DPP moves (no LDS round
        // trip on the critical path of the epilogue).  After the two quad steps all lanes of a quad hold the same value, so
        // mirroring within 8 and within 16 lanes pairs the same sums as xor 4 / xor 8 would (bit-identical results).
        if (seg > 1) t += dpp_f32<0xB1>(t);        // quad_perm [1,0,3,2]
        if (seg > 2) t += dpp_f32<0x4E>(t);        // quad_perm [2,3,0,1]
        if (seg > 4) t += dpp_f32<0x141>(t);       // row_half_mirror
        if (seg > 8) t += dpp_f32<0x140>(t);       // row_mirror
        if (!rk_in_wave && lc.g == 0 && (lc.c % seg) == 0) sInt[pt / seg] = t;
      } else if (lc.g == 0) {
        sInt[pt] = t;                                                 // q does not divide the tile: serial sum
      }
      if (!rk_in_wave) { __syncthreads(); epi_barrier = true; }
      // every lane sums the partials of its own test function (same order in all lanes, so all
      // agree bit for bit): no second barrier and no serial section
      float R = 0.f;
      if (rk_in_wave) {
        R = t;
      } else if (qtree) {
        const int per = q / seg;                                      // partials per test function (a power of two <= 8)
        if (per == 4) {
          // integNum 64 (2D+t, two-point Gauss): the four partials with one 16-byte read instead of four dependent round
          // trips right behind the barrier, where every wave of the workgroup waits for them; same order of additions
          const f32x4 s4 = *reinterpret_cast<const f32x4a*>(&sInt[tf_l * 4]);
          R = ((s4[0] + s4[1]) + s4[2]) + s4[3];
        } else {
          for (int j = 0; j < per; ++j) R += sInt[tf_l * per + j];    // :661
        }
      } else {
        for (int p = 0; p < q; ++p) R += sInt[tf_l * q + p];
      }
      const long k = tile * TT + tf_l;                                // = r0 / q + tf_l
      float s = 0.f;
      if (pt < TPTS && k < A.n_k) {
        const float dj = A.detJv ? A.detJv[k] : A.detJ;
        if (lc.g == 0 && pq_l == 0) {                                 // one lane per test function
          const float lv = dj * R * R;
          loss_var += lv;
          if (A.lossVec) A.lossVec[k] = lv;
        }
        s = 2.f * A.w2 * dj * R * wq;
      }
      udbar = s;
      ubar = -dnt * s;
    } else {
      if (valid) {
        const float e = u - A.label[row];
        const bool isbc = row < A.bDof;
        if (lc.g == 0) {
          const float e2 = A.biDimVal * e * e;
          if (isbc) loss_bc += e2; else loss_ic += e2;
        }
        ubar = (isbc ? cb : ci) * e;
      }
    }

    // ---------------------------------------------------------------- backward
    PA<KS> zb, zdb;
#pragma unroll
    for (int j = 0; j < PA<KS>::NP; ++j) {
      const bool full = 2 * j + 1 < KS;
      const f32x2 wv = {WO[4 * (2 * j) + lc.g], full ? WO[4 * (2 * j + 1) + lc.g] : 0.f};
      const f32x2 av = opaque2(a[L - 1].p[j]);
      const f32x2 sp = act_d1_2<TANH>(av);
      const f32x2 ab = wv * f32x2{ubar, ubar}, adb = wv * f32x2{udbar, udbar};
      const f32x2 zq = adb * sp;
      zdb.p[j] = zq;
      zb.p[j] = ab * sp + zq * act_d2r_2<TANH>(av) * zd[L - 1].p[j];
      // Output-layer weight gradient, no LDS, no MFMA, no barrier: d w_o[f] = sum_p (a[f][p] ubar[p] + adot[f][p] udbar[p]).
      // Both seeds are per-point scalars, so the products are combined per element in registers and summed over the 16
      // lanes of a row (= the wave's 16 points of feature 4 ks + g) with DPP moves; lane c == ks of each row keeps the sum
      // of k-step ks.  This stretch of the tile (epilogue -> first publish round) has no matrix work to hide an LDS round
      // trip behind, and the images stay untouched until the first publish.
      const f32x2 c2 = av * f32x2{ubar, ubar} + (sp * zd[L - 1].p[j]) * f32x2{udbar, udbar};
      const float s0 = rowsum16(c2[0]);
      woacc += (lc.c == 2 * j) ? s0 : 0.f;
      if (2 * j + 1 < KS) {
        const float s1 = rowsum16(c2[1]);
        woacc += (lc.c == 2 * j + 1) ? s1 : 0.f;
      }
    }
    boacc += rowsum16(ubar);                 // d loss / d b_o = sum_p ubar_p (every lane of the wave holds the wave's sum)
    // The lane-major images of the hidden layers overlap other waves' columns of the per-wave (input-layer) transposition
    // at the end of the previous tile: a tile whose epilogue had its own workgroup barrier is already past it.
    if constexpr (HID13) { if (!epi_barrier) __syncthreads(); }
#pragma unroll
    for (int l = L; l >= 2; --l) {
      if (l == 2 && L > 2) {                                 // bring layer-1 activations back
        float xr[KS0], gr[KS0];
#pragma unroll
        for (int s = 0; s < KS0; ++s) { xr[s] = opaque(xin[s]); gr[s] = opaque(gin[s]); }
        f32x4 rv[MT], rt[MT];
        layer1_raw(xr, gr, rv, rt);
#pragma unroll
        for (int j = 0; j < PA<KS>::NP; ++j) {
          a[0].p[j] = act_fin2<TANH>(act_exp2<TANH>(pairOf(rv, j)));
          zd[0].p[j] = pairOf(rt, j);
        }
      }
      if constexpr (HID13) {
        if (l - 2 < LY::NST) {
          f32x4 acc2[2] = {stash[(l - 2) * ST_L], stash[(l - 2) * ST_L + 64]};
          h13_wgrad_layer<TANH>(a[l - 2], zd[l - 2], zb, zdb, TA, lc, wave, lane, t_base_bytes, acc2);
          stash[(l - 2) * ST_L] = acc2[0];
          stash[(l - 2) * ST_L + 64] = acc2[1];
        } else {
          h13_wgrad_layer<TANH>(a[l - 2], zd[l - 2], zb, zdb, TA, lc, wave, lane, t_base_bytes, wacch[l - 2]);
        }
      }
      else if constexpr (NHACC == 2) {
        if (l - 2 < LY::NST) {               // accumulators of this layer live in the LDS stash between tiles
          f32x4 acc2[2] = {stash[(l - 2) * ST_L], stash[(l - 2) * ST_L + 64]};
          wgrad_layer<KS, KS, false, TANH, LY::MERGE>(a[l - 2], zd[l - 2], zb, zdb, TA, TB, lc, wave, acc2, ones_h[l - 2]);
          stash[(l - 2) * ST_L] = acc2[0];
          stash[(l - 2) * ST_L + 64] = acc2[1];
        } else {
          wgrad_layer<KS, KS, false, TANH, LY::MERGE>(a[l - 2], zd[l - 2], zb, zdb, TA, TB, lc, wave, wacch[l - 2], ones_h[l - 2]);
        }
      }
      else wgrad_layer<KS, KS, false, TANH, LY::MERGE>(a[l - 2], zd[l - 2], zb, zdb, TA, TB, lc, wave, wacch[l - 2], ones_h[l - 2]);
      if constexpr (fullpos(KS)) {
        if (!ones_h[l - 2]) thin_bias<KS>(zb, TA, TB, lc, wave, lane, bsum_h[l - 2]);
      }
      const float* Wl = WH + (l - 2) * LY::HPWS;
      int k_out = (net.H[l] + 3) >> 2, m_in = (net.H[l - 1] + 15) >> 4;
      f32x4 accv[MT], acct[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) { accv[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acct[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      float wf[MTM], we[NVE], ev[NVE], et[NVE];
#pragma unroll
      for (int m = 0; m < MTM; ++m) wf[m] = Wl[lc.offB0 + 16 * m * WS + vpos(0, 0)];
#pragma unroll
      for (int v = 0; v < NVE; ++v) {
        we[v] = EDGE ? Wl[(4 * (KS - 1) + v) * WS + 4 * lc.g + vpos(0, 0)] : 0.f;
        ev[v] = 0.f;
        et[v] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wn[MTM], wen[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[lc.offB0 + 16 * m * WS + vpos(ks + 1, 0)] : 0.f;
#pragma unroll
        for (int v = 0; v < NVE; ++v)
          wen[v] = (EDGE && ks + 1 < KS) ? Wl[(4 * (KS - 1) + v) * WS + 4 * lc.g + vpos(ks + 1, 0)] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        if (live_k(ks, k_out)) {
#pragma unroll
          for (int m = 0; m < MTM; ++m) {
            if (!live_m(m, m_in)) continue;
            accv[m] = mfma16(wf[m], zb[ks], accv[m]);
            acct[m] = mfma16(wf[m], zdb[ks], acct[m]);
          }
        }
        if (EDGE) {
#pragma unroll
          for (int v = 0; v < NVE; ++v) { ev[v] += we[v] * zb[ks]; et[v] += we[v] * zdb[ks]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTM; ++m) wf[m] = wn[m];
#pragma unroll
        for (int v = 0; v < NVE; ++v) we[v] = wen[v];
      }
      if (EDGE) {
        accv[MT - 1][0] = edge_reduce_scatter<NVE>(ev, lc.g);
        acct[MT - 1][0] = edge_reduce_scatter<NVE>(et, lc.g);
      }
      // zbar of layer l-1, two k-steps per packed instruction (accumulator rows ks, ks+1 are a register pair)
#pragma unroll
      for (int j = 0; j < PA<KS>::NP; ++j) {
        const int ks = 2 * j;
        if (ks + 1 < KS) {
          const f32x2 av = opaque2(a[l - 2].p[j]);
          const f32x2 sp = act_d1_2<TANH>(av);
          const f32x2 ab = {accv[ks >> 2][ks & 3], accv[ks >> 2][(ks & 3) + 1]};
          const f32x2 adb = {acct[ks >> 2][ks & 3], acct[ks >> 2][(ks & 3) + 1]};
          const f32x2 zq = adb * sp;
          zdb.p[j] = zq;
          zb.p[j] = ab * sp + zq * act_d2r_2<TANH>(av) * zd[l - 2].p[j];
        } else {
          const float av = opaque(a[l - 2][ks]);
          const float sp = act_d1<TANH>(av);
          const float ab = accv[ks >> 2][ks & 3], adb = acct[ks >> 2][ks & 3];
          zdb.set(ks, adb * sp);
          zb.set(ks, ab * sp + adb * sp * act_d2r<TANH>(av) * zd[l - 2][ks]);
        }
      }
    }
    if (thin_in) thin_wgrad_in<KS>(xin, gin, zb, zdb, TA, TB, lc, wave, lane, wacc1[0]);
    else wgrad_layer<KS0, KS, true, TANH, LY::MERGE>(xin, gin, zb, zdb, TA, TB, lc, wave, wacc1, true);
  }

  // ------------------------------------------------------------------ epilogue
  __syncthreads();
  for (int i = tid; i < (LY::G_LOW ? LY::G_SZ : LY::T_SZ); i += NTHREADS) lds[LY::G_OFF + i] = 0.f;
  __syncthreads();
  // Accumulators -> LDS gradient image, every sum in a fixed order (bitwise reproducible).
  // Thin layers first: all eight waves hold partial sums of the same elements; each parks its registers in its own slot
  // ([wave][input | output layer][register][lane]) and the store phase below adds the slots in wave order.
  float* SL = lds + LY::SLOT_OFF;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    SL[((wave * 2 + 0) * 4 + i) * 64 + lane] = thin_in ? wacc1[0][i] : 0.f;
    SL[((wave * 2 + 1) * 4 + i) * 64 + lane] = (i == 0) ? woacc : boacc;
  }
  if constexpr (HID13) {
    // 50-wide hidden layers: a tile wave's first slot and the two border jobs own their image elements
    // exclusively (one parallel phase); the shared tile (m,2) gets its two halves in two phases.
    auto flush_hidden = [&](int slot_lo, int slot_hi) {
#pragma unroll
      for (int l = 2; l <= L; ++l) {
        f32x4 acc2[2];
        if (l - 2 < LY::NST) { acc2[0] = stash[(l - 2) * ST_L]; acc2[1] = stash[(l - 2) * ST_L + 64]; }
        else { acc2[0] = wacch[l - 2][0]; acc2[1] = wacch[l - 2][1]; }
        h13_flush<LY::HP>(acc2, Gacc + LY::G1_SZ + (l - 2) * LY::GH_SZ, lc, wave, lane, slot_lo, slot_hi);
      }
    };
    const bool border = H13::role(wave) != 0;
    flush_hidden(0, border ? 2 : 1);
    __syncthreads();
    if (!border && wave < NW / 2) flush_hidden(1, 2);
    __syncthreads();
    if (!border && wave >= NW / 2) flush_hidden(1, 2);
    if (!thin_in) {                                  // d_in > 3: generic cooperative tiles of the input layer, one round per point split
      using W1 = WG<KS0, KS>;
      const int s1 = (W1::NT >= NW) ? 0 : wave / W1::NT;
      for (int r = 0; r < ((W1::NT >= NW) ? 1 : W1::NS); ++r) {
        __syncthreads();
        if (s1 == r) wgrad_flush<KS0, KS, LY::HP>(wacc1, Gacc, lc, wave);
      }
    }
    __syncthreads();
  } else {
    // generic tiles: waves with the same point-split index own distinct tiles, so round r serves all waves
    // of split r at once (one round when a layer has >= 8 tiles)
    using W1 = WG<KS0, KS>;
    const int s1 = (W1::NT >= NW) ? 0 : wave / W1::NT;
    const int sh = (WHG::NT >= NW) ? 0 : wave / WHG::NT;
    constexpr int R1 = (W1::NT >= NW) ? 1 : W1::NS;
    constexpr int RH = (L > 1) ? ((WHG::NT >= NW) ? 1 : WHG::NS) : 0;
    // bias gradients that did not ride in a constant-one row (KS == 16 behind a 64-wide layer) are per-wave partial sums
    // added in wave order: those rare shapes keep one round per wave
    bool serial_bias = false;
    if constexpr (fullpos(KS)) {
#pragma unroll
      for (int l = 2; l <= L; ++l) serial_bias = serial_bias || !ones_h[l - 2];
    }
    const int r1 = thin_in ? 0 : R1;
    const int nr = serial_bias ? NW : (r1 > RH ? r1 : RH);
    for (int r = 0; r < nr; ++r) {
      if (!thin_in && s1 == r && r < R1) wgrad_flush<KS0, KS, LY::HP>(wacc1, Gacc, lc, wave);
      if (sh == r && r < RH) {
#pragma unroll
        for (int l = 2; l <= L; ++l) {
          if constexpr (NHACC == 2) {
            if (l - 2 < LY::NST) {
              const f32x4 acc2[2] = {stash[(l - 2) * ST_L], stash[(l - 2) * ST_L + 64]};
              wgrad_flush<KS, KS, LY::HP>(acc2, Gacc + LY::G1_SZ + (l - 2) * LY::GH_SZ, lc, wave, ones_h[l - 2]);
              continue;
            }
          }
          wgrad_flush<KS, KS, LY::HP>(wacch[l - 2], Gacc + LY::G1_SZ + (l - 2) * LY::GH_SZ, lc, wave, ones_h[l - 2]);
        }
      }
      if constexpr (fullpos(KS)) {
        if (serial_bias && wave == r) {
          // hidden: lane = position of the output feature; output layer: sum over the wave's 16 points
#pragma unroll
          for (int l = 2; l <= L; ++l)
            if (!ones_h[l - 2] && vfeat(lane) < LY::HP)
              Gacc[LY::G1_SZ + (l - 2) * LY::GH_SZ + LY::HP * LY::HP + vfeat(lane)] += bsum_h[l - 2];
        }
      }
      __syncthreads();
    }
    if (nr == 0) __syncthreads();                    // the slots must be visible to the store phase
  }
  float* out = A.partial + (long)blockIdx.x * P;
#pragma unroll
  for (int l = 1; l <= L + 1; ++l) {
    const int Hin = net.H[l - 1], Hout = net.H[l];
    const int gs = (l == L + 1) ? 1 : LY::HP;
    const int brow = (l == 1) ? 4 * KS0 : LY::HP;
    const float* Gl = Gacc + ((l == 1) ? 0 : (l == L + 1) ? LY::GO_OFF : LY::G1_SZ + (l - 2) * LY::GH_SZ);
    // 64 consecutive threads write one row of the [Hin+1, Hout] block (coalesced, no integer division)
    const int cc = tid & 63;
    if (cc < Hout) {
      for (int r = tid >> 6; r <= Hin; r += NTHREADS / 64) {
        float v = Gl[(r < Hin ? r : brow) * gs + cc];
        if (l == 1 && thin_in) {
          // thin_wgrad_in: register i = input row (3 = bias row), lane = column position of feature cc
          const float* sp = SL + (r < Hin ? r : 3) * 64 + vpos(cc >> 2, cc & 3);
#pragma unroll
          for (int w = 0; w < NW; ++w) v += sp[(w * 2 + 0) * 4 * 64];
        } else if (l == L + 1) {
          // output layer: slot register 0 = woacc (lane 16 g + c holds feature 4 c + g), register 1 = boacc (any lane)
          const float* sp = SL + (r < Hin ? 4 * 64 + 16 * (r & 3) + (r >> 2) : 5 * 64);
#pragma unroll
          for (int w = 0; w < NW; ++w) v += sp[(w * 2) * 4 * 64];
        }
        out[net.woff[l] + r * Hout + cc] = v;
      }
    }
  }
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
  if (tid < 3) {
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += sInt[w * 3 + tid];
    A.losspart[blockIdx.x * 3 + tid] = s;
  }
}

template <int L, int KS, bool TANH>
hipError_t launch_one(const VnFusedArgsD& a, int grid, hipStream_t s) {
  using LY = Lay<L, KS>;
  const size_t bytes = (size_t)LY::TOTAL * sizeof(float);
  // the attribute is per device and sticky: set it once per device (bit mask; engines on different devices may
  // be driven from different threads)
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_fused16_kernel<L, KS, TANH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((vn_fused16_kernel<L, KS, TANH>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

template <int L, int KS>
size_t lds_one() {
  return (size_t)Lay<L, KS>::TOTAL * sizeof(float);
}

int pick_ks(int hmax) {
  if (hmax <= 20) return 5;
  if (hmax <= 32) return 8;
  if (hmax <= 50) return 13;      // NVE == 2: the edge k-step carries features 48, 49 only
  if (hmax <= 64) return 16;      // <= 63: the bias gradient rides in the row of (absent) feature 63; a 64-wide layer's
                                  // successor gets it from thin_bias
  return 0;
}

}  // namespace

#define VN_FUSED16_CASES(X) \
  X(1, 5) X(2, 5) X(3, 5) X(4, 5) X(5, 5) X(6, 5) X(7, 5) X(8, 5)  \
  X(1, 8) X(2, 8) X(3, 8) X(4, 8) X(5, 8) X(6, 8) X(7, 8) X(8, 8)  \
  X(1, 13) X(2, 13) X(3, 13) X(4, 13) X(5, 13) X(6, 13) X(7, 13) X(8, 13)  \
  X(1, 16) X(2, 16) X(3, 16) X(4, 16) X(5, 16) X(6, 16)

int vn_fused16_ks(const VnNet& net) { return pick_ks(net.hmax); }

size_t vn_fused16_lds_bytes(const VnNet& net) {
  const int ks = pick_ks(net.hmax);
#define X(LL, KK) if (net.L == LL && ks == KK) return lds_one<LL, KK>();
  VN_FUSED16_CASES(X)
#undef X
  return 0;
}

bool vn_fused16_net_supported(const VnNet& net) {
  if (net.d_in > 4 * KS0) return false;
  const size_t b = vn_fused16_lds_bytes(net);
  return b != 0 && b <= 160 * 1024;
}

bool vn_fused16_supported(const VnNet& net, int integ_num) {
  if (integ_num < 1 || integ_num > TILE) return false;   // whole test functions must fit a tile
  return vn_fused16_net_supported(net);
}

hipError_t vn_fused16_launch(const VnFusedArgs& h, int grid, hipStream_t s) {
  VnFusedArgsD a;
  a.net = h.net; a.theta = h.theta; a.X = h.X; a.G = h.G; a.src = h.src; a.nT = h.nT; a.n_k = h.n_k;
  a.integ_num = h.integ_num; a.feN = h.feN; a.fedNt = h.fedNt; a.feW = h.feW; a.Nrow = h.Nrow; a.dNtrow = h.dNtrow; a.detJv = h.detJv;
  a.detJ = h.detJ; a.time_dependent = h.time_dependent; a.lossVec = h.lossVec; a.Xb = h.Xb;
  a.label = h.label; a.nB = h.nB; a.bDof = h.bDof; a.biDimVal = h.biDimVal; a.w0 = h.w0; a.w1 = h.w1;
  a.w2 = h.w2; a.partial = h.partial; a.losspart = h.losspart; a.stamps = h.stamps;
  a.mode = h.mode; a.dir = h.mode ? h.dir : -1; a.ostride = h.ostride; a.out_u = h.out_u; a.out_ud = h.out_ud;
  a.seed_u = h.seed_u; a.seed_ud = h.seed_ud;
  const int ks = pick_ks(h.net.hmax);
#define X(LL, KK)                                                                              \
  if (h.net.L == LL && ks == KK)                                                                \
    return h.net.act == VN_ACT_TANH ? launch_one<LL, KK, true>(a, grid, s) : launch_one<LL, KK, false>(a, grid, s);
  VN_FUSED16_CASES(X)
#undef X
  return hipErrorInvalidValue;
}
