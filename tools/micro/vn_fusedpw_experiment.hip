// EXPERIMENT, NOT PART OF THE PRODUCT LIBRARY (round 4, review item 4; result: profiles/r4_pw_perf.txt).
// Built once, parity green, 17.5 % SLOWER than vn_fused16.hip on the bench workload (9.009 vs 7.667 ms): removed by the stop rule.
// To rebuild it: copy this file to varnet_amd/csrc/vn_fusedpw.hip, add it to SRCS in the Makefile, declare
// vn_fusedpw_supported / vn_fusedpw_tile / vn_fusedpw_launch in vn_internal.h and route vn_grad's fused launch to it
// (tile = 64 points); tools/micro/pw_parity_cases.py and tools/micro/pw_perf.py are the checks that were run.
//
// Fused gfx950 kernel, ONE WAVE PER SIMD geometry ("pw": per-wave weight gradient).  Same algorithm and the same
// forward / epilogue / input-gradient code as vn_fused16.hip (forward with one tangent, weak-form epilogue, full reverse
// pass in one persistent launch); what differs is where the weight gradient is contracted:
//
//   vn_fused16: 8 waves x 256 registers.  A wave cannot hold a whole layer's gradient, so the waves PUBLISH their 16 points
//               into workgroup-wide LDS images, barrier, and each contracts its share of the output tiles over all 128 points:
//               two workgroup barriers per round, eight rounds per tile, accumulators of three layers in an LDS stash.
//   here:       4 waves x 512 registers (__launch_bounds__(256, 1)), 16 points per wave, 64-point tiles.  Every wave keeps
//               the accumulators of the WHOLE gradient in registers (44 per 50-wide hidden layer: 3 x 3 core tiles of
//               v_mfma_f32_16x16x4_f32 + one row-border and one column-border accumulator of v_mfma_f32_4x4x1_16B_f32) and
//               contracts its OWN 16 points: the transposition goes through a private LDS image of the wave, there is no
//               publish barrier, no release barrier, no stash.  The only workgroup barrier left in the tile loop is the one
//               that sums R_k of a test function spread over several waves (integNum > 16).
// Matrix work per point is unchanged (the same MFMAs, issued by the wave that owns the point); what goes away is the
// serialisation publish -> barrier -> contract -> barrier and the waits of one SIMD partner for the other.  Latencies are
// hidden by software pipelining inside the wave (weight fragments one k-step ahead, both operand sets of a layer published
// into two image buffers before either is contracted, the next tile's inputs requested a tile ahead), not by a second wave.
//
// Serves: hidden width 33..50 (KS = 13), 1..5 hidden layers, d_in <= 3, integNum dividing 64, the fused training step
// (mode 0).  Everything else stays on vn_fused16.hip.
#include "vn_internal.h"

#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4a __attribute__((may_alias));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NW = 4;
constexpr int NTHREADS = 64 * NW;
constexpr int TILE = 64;
constexpr int CW = 16;        // points per wave
constexpr int WS = 65;        // weight image row stride
constexpr int KS0 = 2;        // input layer k-steps
constexpr int KS = 13;        // hidden k-steps: widths 33..50

__host__ __device__ constexpr int al4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int vpos(int ks, int g) { return 16 * (ks >> 2) + 4 * g + (ks & 3); }
__host__ __device__ constexpr int vks(int pos) { return 4 * (pos >> 4) + (pos & 3); }
__host__ __device__ constexpr int vfeat(int pos) { return 4 * vks(pos) + ((pos >> 2) & 3); }

// Per-wave transposition image ("lane-major", as the H13 images of vn_fused16.hip): element (position pos, point c) sits at
//   vks(pos) * RSW + g(pos) * 16 + c,    g(pos) = (pos >> 2) & 3
// i.e. one row per k-step holds the wave's 64 lanes in lane order, so a publishing store is ds_write_addtid_b32 (no address
// register), and a reader finds 4 consecutive points of one position in one ds_read_b128.
// TA: rows 0..12 = k-steps of the layer's input side, row 13 = [ones | zeros | - | -] (bias row, zero row); TB: rows 0..12.
struct PWI {
  static constexpr int RSW = 64 + 4;
  static constexpr int TA_ROWS = KS + 1, TB_ROWS = KS;
  static constexpr int BUF = (TA_ROWS + TB_ROWS) * RSW;          // one buffer of one wave
  static constexpr int P48 = vpos(12, 0), P49 = vpos(12, 1);     // positions of features 48, 49
  static constexpr int ONES = vpos(13, 0), ZERO = vpos(13, 1);   // positions (k-step 13, g = 0 / 1)
  __host__ __device__ static constexpr int off(int pos) { return vks(pos) * RSW + ((pos >> 2) & 3) * 16; }
  __device__ static __forceinline__ int edge_row(int i) { return i == 0 ? P48 : i == 1 ? P49 : i == 2 ? ONES : ZERO; }
};
static_assert(vks(PWI::ONES) == 13 && vks(PWI::ZERO) == 13 && PWI::P48 == 48 && PWI::P49 == 52, "edge positions");

template <int L>
struct Lay {
  static constexpr int HP = 4 * KS;
  static constexpr int HPWS = al4(HP * WS);
  static constexpr int W1_OFF = 0;                          // [8][WS]
  static constexpr int WH_OFF = al4(8 * WS);                // [L-1][HP][WS]
  static constexpr int BI_OFF = WH_OFF + (L - 1) * HPWS;    // [L][64] biases in (tile, g, i) order
  static constexpr int WO_OFF = BI_OFF + L * 64;            // [4*KS]
  static constexpr int MISC_OFF = WO_OFF + al4(4 * KS);     // sInt[2][64] (double-buffered R_k partials), loss partials
  static constexpr int T_OFF = MISC_OFF + 256;
  static constexpr int T_IMG = NW * 2 * PWI::BUF + 4 * PWI::RSW;          // two buffers per wave (+ rows the border reads overrun)
  static constexpr int G1_SZ = (4 * KS0 + 1) * HP;
  static constexpr int GH_SZ = (HP + 1) * HP;
  static constexpr int GO_OFF = G1_SZ + (L - 1) * GH_SZ;
  static constexpr int G_SZ = al4(GO_OFF + HP + 1);
  static constexpr int SLOT_SZ = NW * 2 * 4 * 64;                          // thin layers: one slot per wave
  static constexpr int T_SZ = al4((T_IMG > G_SZ + SLOT_SZ) ? T_IMG : G_SZ + SLOT_SZ);
  static constexpr int G_OFF = T_OFF;
  static constexpr int SLOT_OFF = T_OFF + al4(G_SZ);
  static constexpr int TOTAL = T_OFF + T_SZ;
  static_assert(TOTAL * 4 <= 160 * 1024, "LDS");
};

__device__ __forceinline__ float opaque(float x) {
  asm("" : "+v"(x));
  return x;
}
__device__ __forceinline__ f32x2 opaque2(f32x2 x) {
  asm("" : "+v"(x));
  return x;
}

// Activation: sigmoid, or tanh = 2*sigmoid(2z) - 1 (VarNet.py:97); everything else is a function of the stored activation a.
template <bool TANH>
__device__ __forceinline__ f32x2 act_exp2(f32x2 z) {
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
template <bool TANH>
__device__ __forceinline__ float act_d1(float a) { return TANH ? __builtin_fmaf(-a, a, 1.f) : a * (1.f - a); }
template <bool TANH>
__device__ __forceinline__ float act_d2r(float a) { return TANH ? -2.f * a : 1.f - 2.f * a; }

// Per-lane value arrays indexed by k-step, kept as even-aligned register pairs (v_pk_* on two k-steps per instruction).
template <int N>
struct PA {
  static constexpr int NP = (N + 1) / 2;
  f32x2 p[NP];
  __device__ __forceinline__ float operator[](int i) const { return p[i >> 1][i & 1]; }
  __device__ __forceinline__ void set(int i, float v) { p[i >> 1][i & 1] = v; }
};

// x summed over the four 16-lane rows of the wave, in every lane: (r0 + r1) + (r2 + r3)
__device__ __forceinline__ float rowsum4(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float s = a + b;
  float c = s, d = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  return c + d;
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rowsum16(float x) {
  x += dpp_f32<0xB1>(x);
  x += dpp_f32<0x4E>(x);
  x += dpp_f32<0x141>(x);
  x += dpp_f32<0x140>(x);
  return x;
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

struct LaneC {
  int g, c;
  int offF;        // forward A-fragment lane offset: g*WS + c            (+ 16*m + 4*ks*WS)
  int offB0;       // backward A-fragment lane offset of row tile 0: fin*WS + 4g   (+ 16m*WS + vpos(ks,0))
};

// Edge rows (features 48, 49): every lane group holds a partial sum of each; group g must end up with the total of feature g.
__device__ __forceinline__ float edge_reduce_scatter(const float (&e)[2], int g) {
  const float keep = (g & 1) ? e[1] : e[0], give = (g & 1) ? e[0] : e[1];
  float t = keep + __shfl_xor(give, 16, 64);
  t += __shfl_xor(t, 32, 64);
  return g < 2 ? t : 0.f;
}

// ---- publish: 14 + 13 ds_write_addtid_b32 of one operand set into one buffer of this wave's image --------------------------
// One asm statement per side: M0 (the store's base) is written inside the statement that uses it.
#define PW_ST(i) "ds_write_addtid_b32 %" #i " offset:%c[o" #i "]\n\t"
__device__ __forceinline__ void pw_store_ta(unsigned base_bytes, const float (&v)[14]) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = PWI::RSW * 4;
  asm volatile("s_mov_b32 m0, %[b]\n\ts_nop 0\n\t" PW_ST(0) PW_ST(1) PW_ST(2) PW_ST(3) PW_ST(4) PW_ST(5) PW_ST(6) PW_ST(7) PW_ST(8)
               PW_ST(9) PW_ST(10) PW_ST(11) PW_ST(12) PW_ST(13)
               :
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                 "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), [b] "s"(base_bytes), [o0] "n"(0 * R), [o1] "n"(1 * R),
                 [o2] "n"(2 * R), [o3] "n"(3 * R), [o4] "n"(4 * R), [o5] "n"(5 * R), [o6] "n"(6 * R), [o7] "n"(7 * R),
                 [o8] "n"(8 * R), [o9] "n"(9 * R), [o10] "n"(10 * R), [o11] "n"(11 * R), [o12] "n"(12 * R), [o13] "n"(13 * R)
               : "memory");
#endif
}
__device__ __forceinline__ void pw_store_tb(unsigned base_bytes, const float (&v)[13]) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = PWI::RSW * 4;
  asm volatile("s_mov_b32 m0, %[b]\n\ts_nop 0\n\t" PW_ST(0) PW_ST(1) PW_ST(2) PW_ST(3) PW_ST(4) PW_ST(5) PW_ST(6) PW_ST(7) PW_ST(8)
               PW_ST(9) PW_ST(10) PW_ST(11) PW_ST(12)
               :
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                 "v"(v[10]), "v"(v[11]), "v"(v[12]), [b] "s"(base_bytes), [o0] "n"(0 * R), [o1] "n"(1 * R), [o2] "n"(2 * R),
                 [o3] "n"(3 * R), [o4] "n"(4 * R), [o5] "n"(5 * R), [o6] "n"(6 * R), [o7] "n"(7 * R), [o8] "n"(8 * R),
                 [o9] "n"(9 * R), [o10] "n"(10 * R), [o11] "n"(11 * R), [o12] "n"(12 * R)
               : "memory");
#endif
}
__device__ __forceinline__ void pw_store_row(unsigned base_bytes, float v) {      // one row at `base_bytes`
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(v), "s"(base_bytes) : "memory");
#endif
}
__device__ __forceinline__ void pw_drain() {      // the compiler does not count these stores in lgkmcnt
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Hidden layer l: dW_l += [a; 1]^T zbar + [adot; 0]^T zdbar over this wave's 16 points.  Both operand sets go into the wave's two
// buffers first, then both are contracted: one LDS write -> read latency per layer, behind no barrier.
struct HAcc {
  f32x4 core[3][3];      // 48 x 48 block: tile (m, n) = input positions 16m.., output positions 16n..
  f32x4 rb;              // rows {48, 49, bias} x every output position (lane = position 4b + j, register = row)
  f32x4 cb;              // input positions 0..47 x columns {48, 49}   (lane 4b + j: column j, register i: position 4b + i)
};

__device__ __forceinline__ void pw_contract(const float* T, const LaneC& lc, int lane, HAcc& h) {
  const float* TA = T;
  const float* TB = T + PWI::TA_ROWS * PWI::RSW;
  const int fo = (lc.c & 3) * PWI::RSW + (lc.c >> 2) * 16 + 4 * lc.g;      // position 16m + c at points 4g..4g+3  (+ 4m rows)
  f32x4 a4[3], b4[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    a4[m] = *reinterpret_cast<const f32x4a*>(&TA[4 * m * PWI::RSW + fo]);
    b4[m] = *reinterpret_cast<const f32x4a*>(&TB[4 * m * PWI::RSW + fo]);
  }
  // border jobs: lane = position
  const int sel = lane & 3;
  const int lane_off = vks(lane) * PWI::RSW + ((lane >> 2) & 3) * 16;
  const float* rA = TA + PWI::off(PWI::edge_row(sel));
  const float* rB = TB + lane_off;
  const float* cA = TA + lane_off;
  const float* cB = (sel == 0) ? TB + PWI::off(PWI::P48) : (sel == 1) ? TB + PWI::off(PWI::P49) : TA + PWI::off(PWI::ZERO);
  f32x4 ra[4], rbv[4], ca[4], cbv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ra[q] = *reinterpret_cast<const f32x4a*>(&rA[4 * q]);
    rbv[q] = *reinterpret_cast<const f32x4a*>(&rB[4 * q]);
    ca[q] = *reinterpret_cast<const f32x4a*>(&cA[4 * q]);
    cbv[q] = *reinterpret_cast<const f32x4a*>(&cB[4 * q]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) h.core[m][n] = mfma16(a4[m][e], b4[n][e], h.core[m][n]);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h.rb = mfma4(ra[q][e], rbv[q][e], h.rb);
      h.cb = mfma4(ca[q][e], cbv[q][e], h.cb);
    }
}

template <bool TANH>
__device__ __forceinline__ void pw_wgrad_hidden(const PA<KS>& av, const PA<KS>& azd, const PA<KS>& bv, const PA<KS>& bt,
                                                float* T, unsigned t_bytes, const LaneC& lc, int lane, HAcc& h) {
  float ta[14], tb[13];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { ta[ks] = av[ks]; tb[ks] = bv[ks]; }
  ta[13] = (lc.g == 0) ? 1.f : 0.f;                                   // bias row | zero row
  pw_store_ta(t_bytes, ta);
  pw_store_tb(t_bytes + PWI::TA_ROWS * PWI::RSW * 4, tb);
#pragma unroll
  for (int j = 0; j < PA<KS>::NP; ++j) {                              // sigma'(a) * zdot, two k-steps per packed instruction
    const f32x2 v2 = act_d1_2<TANH>(opaque2(av.p[j])) * azd.p[j];
    ta[2 * j] = v2[0];
    if (2 * j + 1 < KS) ta[2 * j + 1] = v2[1];
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) tb[ks] = bt[ks];
  ta[13] = 0.f;
  pw_store_ta(t_bytes + PWI::BUF * 4, ta);
  pw_store_tb(t_bytes + (PWI::BUF + PWI::TA_ROWS * PWI::RSW) * 4, tb);
  pw_drain();
  pw_contract(T, lc, lane, h);
  pw_contract(T + PWI::BUF, lc, lane, h);
}

// Input layer (d_in <= 3): rows x0, x1, x2 and the bias row against every output position -- one row-border job per operand set.
// register i of lane 4b + j: i = 0..2 input feature, 3 = bias.
template <class BV>
__device__ __forceinline__ void pw_wgrad_in(float x0, float g0, const BV& bv, const BV& bt, float* T, unsigned t_bytes,
                                            const LaneC& lc, int lane, f32x4& acc) {
  float tb[13];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) tb[ks] = bv[ks];
  // TA row 0: feature g of the input at lane block g (features >= d_in are zero inputs); row 13: ones | zeros
  pw_store_row(t_bytes, x0);
  pw_store_row(t_bytes + 13 * PWI::RSW * 4, (lc.g == 0) ? 1.f : 0.f);
  pw_store_tb(t_bytes + PWI::TA_ROWS * PWI::RSW * 4, tb);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) tb[ks] = bt[ks];
  pw_store_row(t_bytes + PWI::BUF * 4, g0);
  pw_store_row(t_bytes + (PWI::BUF + 13 * PWI::RSW) * 4, 0.f);
  pw_store_tb(t_bytes + (PWI::BUF + PWI::TA_ROWS * PWI::RSW) * 4, tb);
  pw_drain();
  const int sel = lane & 3;
  const int lane_off = vks(lane) * PWI::RSW + ((lane >> 2) & 3) * 16;
  const int a_off = (sel == 3) ? PWI::off(PWI::ONES) : 16 * sel;                 // input feature sel: row 0, lane block sel
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const float* TA = T + half * PWI::BUF;
    const float* TB = TA + PWI::TA_ROWS * PWI::RSW;
    f32x4 a4[4], b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a4[q] = *reinterpret_cast<const f32x4a*>(&TA[a_off + 4 * q]);
      b4[q] = *reinterpret_cast<const f32x4a*>(&TB[lane_off + 4 * q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma4(a4[q][e], b4[q][e], acc);
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
};

#if defined(__HIP_DEVICE_COMPILE__)
#define VN_NO_LDS_PAIRING __attribute__((target("no-load-store-opt")))
#else
#define VN_NO_LDS_PAIRING
#endif

template <int L, bool TANH>
__global__ __launch_bounds__(NTHREADS, 1) VN_NO_LDS_PAIRING void vn_fusedpw_kernel(VnFusedArgsD A) {
  using LY = Lay<L>;
  constexpr int MT = 4;            // row tiles of a 52-position layer; the 4th holds the edge k-step only
  constexpr int MTM = 3;           // row tiles produced by MFMA (features 48, 49 on the VALU)
  constexpr int NVE = 2;
  constexpr int EPOS = 16 * (MT - 1);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const VnNet& net = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = net.P;
  float* W1 = lds + LY::W1_OFF;
  float* WH = lds + LY::WH_OFF;
  float* BI = lds + LY::BI_OFF;
  float* WO = lds + LY::WO_OFF;
  float* sInt = lds + LY::MISC_OFF;
  float* Tw = lds + LY::T_OFF + wave * 2 * PWI::BUF;                       // this wave's two image buffers
  const unsigned t_bytes = (unsigned)((LY::T_OFF + wave * 2 * PWI::BUF) * 4);
  float* Gacc = lds + LY::G_OFF;

  const int q = A.integ_num;
  const int pq_l = (wave * CW + (lane & 15)) % q;
  const int tf_l = (wave * CW + (lane & 15)) / q;
  const float tab_dnt = A.time_dependent ? A.fedNt[pq_l] : 0.f;
  const float tab_w = A.feW ? A.feW[pq_l] : 1.f;
  const float tab_N = A.feN[pq_l];
  const float bo = A.theta[net.boff[L + 1]];
  // ------------------------------------------------------------------ prologue: LDS images
  {
    const int d_in = net.d_in, H1 = net.H[1];
    constexpr int NSRC = (LY::HP * LY::HP + NTHREADS - 1) / NTHREADS;
    float v1 = 0.f, vh[L > 1 ? L - 1 : 1][NSRC];
    if (tid < d_in * H1) v1 = A.theta[net.woff[1] + tid];                  // d_in <= 3, H1 <= 50: one per thread
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
    constexpr int NBI = (L * 64 + NTHREADS - 1) / NTHREADS;
    float vb[NBI], vo = 0.f;
#pragma unroll
    for (int it = 0; it < NBI; ++it) {
      const int t = tid + it * NTHREADS;
      vb[it] = 0.f;
      if (t < L * 64) {
        const int l = t / 64 + 1, idx = t % 64;
        const int mt = idx >> 4, g = (idx >> 2) & 3, r = idx & 3;       // [tile][g][i]
        const int ks = 4 * mt + r, f = 4 * ks + g;
        vb[it] = (ks < KS && f < net.H[l]) ? A.theta[net.boff[l] + f] : 0.f;
      }
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
      const int dq = NTHREADS / Hout, dr = NTHREADS - dq * Hout;
      int k = tid / Hout, f = tid - k * Hout;
#pragma unroll
      for (int it = 0; it < NSRC; ++it) {
        if (tid + it * NTHREADS < n) Wl[k * WS + vpos(f >> 2, f & 3)] = vh[l - 2][it];
        f += dr; k += dq;
        if (f >= Hout) { f -= Hout; ++k; }
      }
    }
#pragma unroll
    for (int it = 0; it < NBI; ++it)
      if (tid + it * NTHREADS < L * 64) BI[tid + it * NTHREADS] = vb[it];
    if (tid < 4 * KS) WO[tid] = vo;
  }
  __syncthreads();

  LaneC lc;
  lc.g = lane >> 4;
  lc.c = lane & 15;
  lc.offF = lc.g * WS + lc.c;
  lc.offB0 = vfeat(lc.c) * WS + 4 * lc.g;

  // persistent weight-gradient accumulators: the whole gradient, in this wave's registers
  HAcc hacc[L > 1 ? L - 1 : 1];
#pragma unroll
  for (int l = 0; l < (L > 1 ? L - 1 : 1); ++l) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) hacc[l].core[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    hacc[l].rb = f32x4{0.f, 0.f, 0.f, 0.f};
    hacc[l].cb = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wacc1 = {0.f, 0.f, 0.f, 0.f};
  float woacc = 0.f, boacc = 0.f;

  const int TT = TILE / q;                                   // whole test functions per tile (q divides 64)
  const long ntiles_i = (A.n_k + TT - 1) / TT;
  const long ntiles = ntiles_i + (A.nB + TILE - 1) / TILE;
  float loss_var = 0.f, loss_bc = 0.f, loss_ic = 0.f;
  const long nI = A.nB - A.bDof;
  const float cb = A.bDof > 0 ? 2.f * A.w0 * A.biDimVal / (float)A.bDof : 0.f;
  const float ci = nI > 0 ? 2.f * A.w1 * A.biDimVal / (float)nI : 0.f;

  // inputs of a tile: feature 4s + g of this lane's point (k-step 0 only matters: d_in <= 3), requested one tile ahead
  auto fetch = [&](long tile, float& x0, float& g0, float& srcv, float& dntv, float& nrv, float& labv) {
    const bool interior = tile < ntiles_i;
    const long r0 = (interior ? tile : tile - ntiles_i) * TILE;
    const long nrows = interior ? A.nT : A.nB;
    const long row = r0 + wave * CW + lc.c;
    const bool valid = tile < ntiles && row < nrows;
    const float* Xp = interior ? A.X : A.Xb;
    x0 = (valid && lc.g < net.d_in) ? Xp[row * net.d_in + lc.g] : 0.f;
    g0 = (valid && interior && lc.g < net.dim) ? A.G[row * net.dim + lc.g] : 0.f;
    srcv = (valid && interior && A.src) ? A.src[row] : 0.f;
    dntv = (valid && interior && A.dNtrow) ? A.dNtrow[row] : 0.f;
    nrv = (valid && interior && A.Nrow) ? A.Nrow[row] : 0.f;
    labv = (valid && !interior) ? A.label[row] : 0.f;
  };
  float nx0, ng0, nsrc, ndnt, nnr, nlab;
  fetch(blockIdx.x, nx0, ng0, nsrc, ndnt, nnr, nlab);

  int par = 0;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    asm volatile("" ::: "memory");                 // keep LDS fragment loads inside the loop
    const bool interior = tile < ntiles_i;
    const long r0 = (interior ? tile : tile - ntiles_i) * TILE;
    const long nrows = interior ? A.nT : A.nB;
    const int pt = wave * CW + lc.c;
    const long row = r0 + pt;
    const bool valid = row < nrows;
    const float x0 = nx0, g0 = ng0, srow = nsrc, dntrow = ndnt, nrow_v = nnr, lab = nlab;
    fetch(tile + gridDim.x, nx0, ng0, nsrc, ndnt, nnr, nlab);

    PA<KS> a[L], zd[L];

    // ---------------------------------------------------------------- layer 1 (also recomputed late)
    auto layer1_raw = [&](float xi, float gi, f32x4 (&ov)[MT], f32x4 (&ot)[MT]) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ov[m] = *reinterpret_cast<const f32x4a*>(&BI[m * 16 + lc.g * 4]);
        ot[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float wf = W1[lc.offF + 16 * m];
        ov[m] = mfma16(wf, xi, ov[m]);
        ot[m] = mfma16(wf, gi, ot[m]);
      }
    };

    // ---------------------------------------------------------------- forward
    f32x4 pv[MT], ptn[MT];
    layer1_raw(x0, g0, pv, ptn);
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const float* Wl = WH + (l - 2) * LY::HPWS;
      f32x4 nv[MT], nt[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        nv[m] = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + m * 16 + lc.g * 4]);
        nt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      constexpr int NP = PA<KS>::NP;
      auto zin2 = [&](int j) { return f32x2{pv[(2 * j) >> 2][(2 * j) & 3], pv[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
      auto zdin2 = [&](int j) { return f32x2{ptn[(2 * j) >> 2][(2 * j) & 3], ptn[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
      float wf[MTM], we[NVE], ev[NVE], et[NVE];
#pragma unroll
      for (int m = 0; m < MTM; ++m) wf[m] = Wl[lc.offF + 16 * m];
#pragma unroll
      for (int v = 0; v < NVE; ++v) {
        we[v] = Wl[lc.offF - lc.c + EPOS + 4 * v];
        ev[v] = 0.f;
        et[v] = 0.f;
      }
      f32x2 cs2 = act_fin2<TANH>(act_exp2<TANH>(zin2(0)));
      f32x2 cq2 = act_d1_2<TANH>(cs2) * zdin2(0);
      a[l - 2].p[0] = cs2;
      zd[l - 2].p[0] = zdin2(0);
      f32x2 s1 = act_fin2<TANH>(act_exp2<TANH>(zin2(1)));
      f32x2 e2 = act_exp2<TANH>(zin2(2));
      f32x2 e3 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int j = ks >> 1;
        float wn[MTM], wen[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + lc.offF + 16 * m] : 0.f;
#pragma unroll
        for (int v = 0; v < NVE; ++v) wen[v] = (ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + lc.offF - lc.c + EPOS + 4 * v] : 0.f;
        const float cs = cs2[ks & 1], cq = cq2[ks & 1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTM; ++m) {
          nv[m] = mfma16(wf[m], cs, nv[m]);
          nt[m] = mfma16(wf[m], cq, nt[m]);
        }
#pragma unroll
        for (int v = 0; v < NVE; ++v) { ev[v] += we[v] * cs; et[v] += we[v] * cq; }
        if ((ks & 1) == 0) {
          if (j + 3 < NP) e3 = act_exp2<TANH>(zin2(j + 3));
          __builtin_amdgcn_sched_barrier(0);
        } else {
          f32x2 s2 = s1, q1 = cq2;
          if (j + 2 < NP) s2 = act_fin2<TANH>(e2);
          if (j + 1 < NP) {
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
      nv[MT - 1][0] += edge_reduce_scatter(ev, lc.g);         // bias was loaded above
      nt[MT - 1][0] = edge_reduce_scatter(et, lc.g);
#pragma unroll
      for (int m = 0; m < MT; ++m) { pv[m] = nv[m]; ptn[m] = nt[m]; }
    }
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
    u = rowsum4(u);  ud = rowsum4(ud);
    u += bo;

    // ---------------------------------------------------------------- weak-form epilogue
    float ubar = 0.f, udbar = 0.f;
    if (interior) {
      const float dnt = !A.time_dependent ? 0.f : (A.dNtrow ? dntrow : tab_dnt);
      const float wq = tab_w;
      float t = ud - dnt * u;
      if (A.src) t -= srow * (A.Nrow ? nrow_v : tab_N);
      t *= wq;
      if (!valid) t = 0.f;
      const int seg = q < CW ? q : CW;
      const bool rk_in_wave = q <= CW;
      if (seg > 1) t += dpp_f32<0xB1>(t);
      if (seg > 2) t += dpp_f32<0x4E>(t);
      if (seg > 4) t += dpp_f32<0x141>(t);
      if (seg > 8) t += dpp_f32<0x140>(t);
      float* sI = sInt + par * 64;                           // double-buffered: a wave can be one barrier ahead of another
      float R = t;
      if (!rk_in_wave) {
        if (lc.g == 0 && lc.c == 0) sI[wave] = t;            // one partial per wave (seg == 16)
        __syncthreads();
        const int per = q / CW;                              // 2 or 4 waves per test function
        R = 0.f;
        for (int j = 0; j < per; ++j) R += sI[tf_l * per + j];
      }
      const long k = tile * TT + tf_l;
      float s = 0.f;
      if (k < A.n_k) {
        const float dj = A.detJv ? A.detJv[k] : A.detJ;
        if (lc.g == 0 && pq_l == 0) {
          const float lv = dj * R * R;
          loss_var += lv;
          if (A.lossVec) A.lossVec[k] = lv;
        }
        s = 2.f * A.w2 * dj * R * wq;
      }
      udbar = s;
      ubar = -dnt * s;
    } else if (valid) {
      const float e = u - lab;
      const bool isbc = row < A.bDof;
      if (lc.g == 0) {
        const float e2 = A.biDimVal * e * e;
        if (isbc) loss_bc += e2; else loss_ic += e2;
      }
      ubar = (isbc ? cb : ci) * e;
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
      // output-layer weight gradient in registers (see vn_fused16.hip): lane c == ks of each row keeps the sum of k-step ks
      const f32x2 c2 = av * f32x2{ubar, ubar} + (sp * zd[L - 1].p[j]) * f32x2{udbar, udbar};
      const float s0 = rowsum16(c2[0]);
      woacc += (lc.c == 2 * j) ? s0 : 0.f;
      if (2 * j + 1 < KS) {
        const float s1 = rowsum16(c2[1]);
        woacc += (lc.c == 2 * j + 1) ? s1 : 0.f;
      }
    }
    boacc += rowsum16(ubar);
#pragma unroll
    for (int l = L; l >= 2; --l) {
      if (l == 2 && L > 2) {                                 // bring layer-1 activations back
        f32x4 rv[MT], rt[MT];
        layer1_raw(opaque(x0), opaque(g0), rv, rt);
#pragma unroll
        for (int j = 0; j < PA<KS>::NP; ++j) {
          a[0].p[j] = act_fin2<TANH>(act_exp2<TANH>(pairOf(rv, j)));
          zd[0].p[j] = pairOf(rt, j);
        }
      }
      pw_wgrad_hidden<TANH>(a[l - 2], zd[l - 2], zb, zdb, Tw, t_bytes, lc, lane, hacc[l - 2]);
      const float* Wl = WH + (l - 2) * LY::HPWS;
      f32x4 accv[MT], acct[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) { accv[m] = f32x4{0.f, 0.f, 0.f, 0.f}; acct[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      float wf[MTM], we[NVE], ev[NVE], et[NVE];
#pragma unroll
      for (int m = 0; m < MTM; ++m) wf[m] = Wl[lc.offB0 + 16 * m * WS + vpos(0, 0)];
#pragma unroll
      for (int v = 0; v < NVE; ++v) {
        we[v] = Wl[(4 * (KS - 1) + v) * WS + 4 * lc.g + vpos(0, 0)];
        ev[v] = 0.f;
        et[v] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wn[MTM], wen[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[lc.offB0 + 16 * m * WS + vpos(ks + 1, 0)] : 0.f;
#pragma unroll
        for (int v = 0; v < NVE; ++v) wen[v] = (ks + 1 < KS) ? Wl[(4 * (KS - 1) + v) * WS + 4 * lc.g + vpos(ks + 1, 0)] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTM; ++m) {
          accv[m] = mfma16(wf[m], zb[ks], accv[m]);
          acct[m] = mfma16(wf[m], zdb[ks], acct[m]);
        }
#pragma unroll
        for (int v = 0; v < NVE; ++v) { ev[v] += we[v] * zb[ks]; et[v] += we[v] * zdb[ks]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTM; ++m) wf[m] = wn[m];
#pragma unroll
        for (int v = 0; v < NVE; ++v) we[v] = wen[v];
      }
      accv[MT - 1][0] = edge_reduce_scatter(ev, lc.g);
      acct[MT - 1][0] = edge_reduce_scatter(et, lc.g);
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
    pw_wgrad_in(x0, g0, zb, zdb, Tw, t_bytes, lc, lane, wacc1);
  }

  // ------------------------------------------------------------------ flush: accumulators -> LDS gradient image, in wave order
  __syncthreads();
  for (int i = tid; i < LY::T_SZ; i += NTHREADS) lds[LY::T_OFF + i] = 0.f;
  __syncthreads();
  float* SL = lds + LY::SLOT_OFF;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    SL[((wave * 2 + 0) * 4 + i) * 64 + lane] = wacc1[i];
    SL[((wave * 2 + 1) * 4 + i) * 64 + lane] = (i == 0) ? woacc : boacc;
  }
  constexpr int GS = LY::HP;
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int l = 2; l <= L; ++l) {
        float* Gl = Gacc + LY::G1_SZ + (l - 2) * LY::GH_SZ;
        const HAcc& h = hacc[l - 2];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) {
            const int col = vfeat(16 * n + lc.c);
#pragma unroll
            for (int i = 0; i < 4; ++i) Gl[(4 * (4 * m + i) + lc.g) * GS + col] += h.core[m][n][i];
          }
        {                                                     // rows 48, 49, bias x every output position (lane = position)
          const int col = vfeat(lane);
          if (vks(lane) < KS && col < GS) {
            Gl[48 * GS + col] += h.rb[0];
            Gl[49 * GS + col] += h.rb[1];
            Gl[52 * GS + col] += h.rb[2];                    // bias row of the gradient image (row 4*KS)
          }
        }
        {                                                     // input positions 0..47 x columns 48, 49
          const int j = lane & 3;
          if (j < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int pos = (lane & ~3) + i;
              if (pos < 48) Gl[vfeat(pos) * GS + 48 + j] += h.cb[i];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  float* out = A.partial + (long)blockIdx.x * P;
#pragma unroll
  for (int l = 1; l <= L + 1; ++l) {
    const int Hin = net.H[l - 1], Hout = net.H[l];
    const int gs = (l == L + 1) ? 1 : LY::HP;
    const int brow = (l == 1) ? 4 * KS0 : LY::HP;
    const float* Gl = Gacc + ((l == 1) ? 0 : (l == L + 1) ? LY::GO_OFF : LY::G1_SZ + (l - 2) * LY::GH_SZ);
    const int cc = tid & 63;
    if (cc < Hout) {
      for (int r = tid >> 6; r <= Hin; r += NTHREADS / 64) {
        float v = Gl[(r < Hin ? r : brow) * gs + cc];
        if (l == 1) {
          // pw_wgrad_in: register i = input row (3 = bias row), lane = column position of feature cc
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
  float* sL = sInt + 128;
  if (lane == 0) { sL[wave * 3 + 0] = v0; sL[wave * 3 + 1] = v1; sL[wave * 3 + 2] = v2; }
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += sL[w * 3 + tid];
    A.losspart[blockIdx.x * 3 + tid] = s;
  }
}

template <int L, bool TANH>
hipError_t launch_one(const VnFusedArgsD& a, int grid, hipStream_t s) {
  const size_t bytes = (size_t)Lay<L>::TOTAL * sizeof(float);
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_fusedpw_kernel<L, TANH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((vn_fusedpw_kernel<L, TANH>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

}  // namespace

bool vn_fusedpw_supported(const VnNet& net, int integ_num) {
  if (net.act == VN_ACT_PER_LAYER) return false;
  if (net.L < 1 || net.L > 5 || net.d_in > 3 || net.hmax < 33 || net.hmax > 50) return false;
  return integ_num >= 1 && integ_num <= TILE && (TILE % integ_num) == 0;
}

int vn_fusedpw_tile() { return TILE; }

hipError_t vn_fusedpw_launch(const VnFusedArgs& h, int grid, hipStream_t s) {
  if (h.mode != 0) return hipErrorInvalidValue;
  VnFusedArgsD a;
  a.net = h.net; a.theta = h.theta; a.X = h.X; a.G = h.G; a.src = h.src; a.nT = h.nT; a.n_k = h.n_k;
  a.integ_num = h.integ_num; a.feN = h.feN; a.fedNt = h.fedNt; a.feW = h.feW; a.Nrow = h.Nrow; a.dNtrow = h.dNtrow; a.detJv = h.detJv;
  a.detJ = h.detJ; a.time_dependent = h.time_dependent; a.lossVec = h.lossVec; a.Xb = h.Xb;
  a.label = h.label; a.nB = h.nB; a.bDof = h.bDof; a.biDimVal = h.biDimVal; a.w0 = h.w0; a.w1 = h.w1;
  a.w2 = h.w2; a.partial = h.partial; a.losspart = h.losspart;
  const bool th = h.net.act == VN_ACT_TANH;
  switch (h.net.L) {
    case 1: return th ? launch_one<1, true>(a, grid, s) : launch_one<1, false>(a, grid, s);
    case 2: return th ? launch_one<2, true>(a, grid, s) : launch_one<2, false>(a, grid, s);
    case 3: return th ? launch_one<3, true>(a, grid, s) : launch_one<3, false>(a, grid, s);
    case 4: return th ? launch_one<4, true>(a, grid, s) : launch_one<4, false>(a, grid, s);
    case 5: return th ? launch_one<5, true>(a, grid, s) : launch_one<5, false>(a, grid, s);
  }
  return hipErrorInvalidValue;
}
