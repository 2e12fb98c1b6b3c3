// Value and input gradient of the network at a set of points in ONE pass: value forward, then the value-adjoint
// sweep back to the inputs with seed 1 -- the reference's own tf.gradients(model(Input), Input) (TFModel.py:536-541):
// 2 F_pt per point whatever the number of coordinates, where one forward-tangent pass per coordinate costs dim * 2 F_pt.
// Serves the de-duplicated formulation (vn_dedup.hip), which needs (u, du/dx_d) once per unique quadrature point
// before the weak-form assembly over rows, vn_forward_grad, and -- without the adjoint sweep (out_g == nullptr) -- vn_forward:
// the value alone at F_pt per point (the fused kernel's forward-only mode carries a tangent stream of zeros through every
// layer for such calls: twice the matrix work).
//
// Same geometry and data layout as vn_fused16.hip (8 waves, 16 points per wave, v_mfma_f32_16x16x4_f32, feature f in
// k-step f/4 / lane group f%4, layers chained in registers, weight images [in-feature][out-position] with row stride 65
// in LDS: forward fragments and the transposed fragments of the sweep come from the one copy), with ONE stream per
// direction instead of two, no weight gradient, hence no transposition images, no workgroup barrier after the prologue:
// every wave walks its own 16-point chunks.  The activations of all layers stay in registers between the two sweeps
// (KS * L values per lane: 65 at 5x50).
#include "vn_points16.h"
#include "vn_pgrad16.h"

#include <atomic>

namespace {
using namespace vn16;

struct VnPgradArgsD {
  VnNet net;
  const float* theta;
  const float* X;            // [n, d_in]
  long n;
  float* out_u;              // [n]
  float* out_g;              // [n, dim]: du/dx_d, d < dim (the leading coordinates of X)
  float* out_pack;           // [n, 4] = (u, du/dx_0, du/dx_1, du/dx_2) or nullptr
};

template <int L, int KS, bool TANH>
#if defined(__HIP_DEVICE_COMPILE__)
#define VN_NO_LDS_PAIRING __attribute__((target("no-load-store-opt")))      // see vn_fused16.hip
#else
#define VN_NO_LDS_PAIRING
#endif
__global__ __launch_bounds__(NTHREADS, 2) VN_NO_LDS_PAIRING void vn_pgrad16_kernel(VnPgradArgsD A) {
  using LY = PLay<L, KS>;
  constexpr int MT = mtiles(KS);
  constexpr bool EDGE = (KS % 4) == 1 && KS > 1;     // the last row tile holds one k-step: its rows run on the VALU (vn_fused16.hip)
  constexpr int MTM = EDGE ? MT - 1 : MT;
  constexpr int NVE = (KS == 13) ? 2 : 4;
  constexpr int EPOS = 16 * (MT - 1);
  constexpr bool KSKIP = KS <= 8;                    // k-steps / row tiles that hold only padding are branched over
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
  float* W1 = lds + LY::W1_OFF;
  float* WH = lds + LY::WH_OFF;
  float* BI = lds + LY::BI_OFF;
  float* WO = lds + LY::WO_OFF;
  const float bo = A.theta[net.boff[L + 1]];
  stage_weight_images<L, KS>(net, A.theta, lds, tid);      // vn_points16.h
  __syncthreads();

  const int g = lane >> 4, c = lane & 15;
  const int offF = g * WS + c;                       // forward A fragment: in-feature 4ks+g, out-position 16m+c
  static_assert(16 * MTM <= LY::HP, "backward fragment rows stay inside the weight image");
  const int offB0 = vfeat(c) * WS + 4 * g;           // transposed A fragment of row tile 0: in-feature vfeat(c), out-position vpos(ks, g)

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

    PA<KS> a[L];
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
      const float* Wl = WH + (l - 2) * LY::HPWS;
      int k_in = (net.H[l - 1] + 3) >> 2, m_out = (net.H[l] + 15) >> 4;
      f32x4 nv[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) nv[m] = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + m * 16 + g * 4]);
      // activation of the previous layer pipelined under this layer's MFMAs in pairs of k-steps: stage A (packed scale + 2 v_exp)
      // of pair j+3 after the first k-step of pair j, stage B (packed 1+e + 2 v_rcp) of pair j+2 after the second
      constexpr int NP = PA<KS>::NP;
      auto zin2 = [&](int j) { return f32x2{pv[(2 * j) >> 2][(2 * j) & 3], pv[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
      float wf[MTM > 0 ? MTM : 1], we[NVE], ev[NVE];
#pragma unroll
      for (int m = 0; m < MTM; ++m) wf[m] = Wl[offF + 16 * m];
#pragma unroll
      for (int v = 0; v < NVE; ++v) {
        we[v] = EDGE ? Wl[offF - c + EPOS + 4 * v] : 0.f;
        ev[v] = 0.f;
      }
      f32x2 cs2 = act_fin2<TANH>(act_exp2<TANH>(zin2(0)));
      a[l - 2].p[0] = cs2;
      f32x2 s1 = (NP > 1) ? act_fin2<TANH>(act_exp2<TANH>(zin2(1))) : f32x2{0.f, 0.f};
      f32x2 e2 = (NP > 2) ? act_exp2<TANH>(zin2(2)) : f32x2{0.f, 0.f};
      f32x2 e3 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int j = ks >> 1;
        float wn[MTM > 0 ? MTM : 1], wen[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + offF + 16 * m] : 0.f;
#pragma unroll
        for (int v = 0; v < NVE; ++v) wen[v] = (EDGE && ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + offF - c + EPOS + 4 * v] : 0.f;
        const float cs = cs2[ks & 1];
        __builtin_amdgcn_sched_barrier(0);
        if (live_k(ks, k_in)) {
#pragma unroll
          for (int m = 0; m < MTM; ++m) {
            if (!live_m(m, m_out)) continue;
            nv[m] = mfma16(wf[m], cs, nv[m]);
          }
        }
        if (EDGE) {
#pragma unroll
          for (int v = 0; v < NVE; ++v) ev[v] += we[v] * cs;
        }
        if ((ks & 1) == 0) {
          if (j + 3 < NP) e3 = act_exp2<TANH>(zin2(j + 3));
          __builtin_amdgcn_sched_barrier(0);
        } else {
          f32x2 s2 = s1;
          if (j + 2 < NP) s2 = act_fin2<TANH>(e2);
          if (j + 1 < NP) a[l - 2].p[j + 1] = s1;
          __builtin_amdgcn_sched_barrier(0);
          cs2 = s1; s1 = s2; e2 = e3;
        }
#pragma unroll
        for (int m = 0; m < MTM; ++m) wf[m] = wn[m];
#pragma unroll
        for (int v = 0; v < NVE; ++v) we[v] = wen[v];
      }
      if (EDGE) nv[MT - 1][0] += edge_reduce_scatter<NVE>(ev, g);       // bias was loaded above
#pragma unroll
      for (int m = 0; m < MT; ++m) pv[m] = nv[m];
    }
    auto pairOf = [](const f32x4 (&t)[MT], int j) { return f32x2{t[(2 * j) >> 2][(2 * j) & 3], t[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
#pragma unroll
    for (int j = 0; j < PA<KS>::NP; ++j) a[L - 1].p[j] = act_fin2<TANH>(act_exp2<TANH>(pairOf(pv, j)));

    // ---------------------------------------------------------------- output layer and its adjoint (seed 1)
    PA<KS> zb;
    float u = 0.f;
    {
      f32x2 u2 = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < PA<KS>::NP; ++j) {
        const bool full = 2 * j + 1 < KS;
        const f32x2 wv = {WO[4 * (2 * j) + g], full ? WO[4 * (2 * j + 1) + g] : 0.f};
        const f32x2 av = a[L - 1].p[j];
        u2 += wv * av;
        zb.p[j] = wv * act_d1_2<TANH>(av);             // d u / d z_L
      }
      u = rowsum4(u2[0] + u2[1]) + bo;
    }
    if (A.out_g == nullptr && A.out_pack == nullptr) {   // value only (vn_forward): F_pt per point, no adjoint sweep
      if (valid && g == 3 && A.out_u) A.out_u[row] = u;
      continue;
    }

    // ---------------------------------------------------------------- value-adjoint sweep to the inputs
#pragma unroll
    for (int l = L; l >= 2; --l) {
      const float* Wl = WH + (l - 2) * LY::HPWS;
      int k_out = (net.H[l] + 3) >> 2, m_in = (net.H[l - 1] + 15) >> 4;
      f32x4 accv[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) accv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      float wf[MTM > 0 ? MTM : 1], we[NVE], ev[NVE];
#pragma unroll
      for (int m = 0; m < MTM; ++m) wf[m] = Wl[offB0 + 16 * m * WS + vpos(0, 0)];
#pragma unroll
      for (int v = 0; v < NVE; ++v) {
        we[v] = EDGE ? Wl[(4 * (KS - 1) + v) * WS + 4 * g + vpos(0, 0)] : 0.f;
        ev[v] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wn[MTM > 0 ? MTM : 1], wen[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[offB0 + 16 * m * WS + vpos(ks + 1, 0)] : 0.f;
#pragma unroll
        for (int v = 0; v < NVE; ++v) wen[v] = (EDGE && ks + 1 < KS) ? Wl[(4 * (KS - 1) + v) * WS + 4 * g + vpos(ks + 1, 0)] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        if (live_k(ks, k_out)) {
#pragma unroll
          for (int m = 0; m < MTM; ++m) {
            if (!live_m(m, m_in)) continue;
            accv[m] = mfma16(wf[m], zb[ks], accv[m]);
          }
        }
        if (EDGE) {
#pragma unroll
          for (int v = 0; v < NVE; ++v) ev[v] += we[v] * zb[ks];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTM; ++m) wf[m] = wn[m];
#pragma unroll
        for (int v = 0; v < NVE; ++v) we[v] = wen[v];
      }
      if (EDGE) accv[MT - 1][0] = edge_reduce_scatter<NVE>(ev, g);
#pragma unroll
      for (int j = 0; j < PA<KS>::NP; ++j) {
        const int ks = 2 * j;
        if (ks + 1 < KS) {
          const f32x2 ab = {accv[ks >> 2][ks & 3], accv[ks >> 2][(ks & 3) + 1]};
          zb.p[j] = ab * act_d1_2<TANH>(a[l - 2].p[j]);
        } else {
          zb.set(ks, accv[ks >> 2][ks & 3] * act_d1<TANH>(a[l - 2][ks]));
        }
      }
    }
    // input layer: du/dx_d = sum_f W_1[d][f] zbar_1[f]; this lane holds feature 4ks+g of every k-step, the four lane
    // groups are summed by row swaps (same pairing in every lane: all agree bit for bit)
    float xg[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d < net.dim) {
        float t = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) t += W1[d * WS + vpos(ks, 0) + 4 * g] * zb[ks];
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

template <int L, int KS, bool TANH>
hipError_t launch_one(const VnPgradArgsD& a, int ncu, int wgs_per_cu, hipStream_t s) {
  const size_t bytes = (size_t)PLay<L, KS>::TOTAL * sizeof(float);
  // the attribute is per device and sticky: set it once per device (bit mask; engines on different devices may be
  // driven from different threads)
  static std::atomic<unsigned long long> attr_done{0};
  static std::atomic<int> occ{0};                    // workgroups of this instantiation a CU holds (registers, LDS)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_pgrad16_kernel<L, KS, TANH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    int nb = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)vn_pgrad16_kernel<L, KS, TANH>, NTHREADS, bytes) != hipSuccess) {
      (void)hipGetLastError();
      nb = 1;
    }
    occ.store(nb < 1 ? 1 : nb, std::memory_order_relaxed);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  // Waves are independent here (no workgroup barrier in the chunk loop), so a second resident workgroup per CU -- where
  // the instantiation's registers (<= 128) and weight images (<= 80 KB) allow it -- hides more of the LDS / transcendental
  // latencies under the partner's MFMAs: 395 -> 382 us on the bench network (profiles/r5_dedup_ab.txt).
  int per_cu = wgs_per_cu > 0 ? wgs_per_cu : (occ.load(std::memory_order_relaxed) >= 2 ? 2 : 1);
  const long wgs = ((a.n + CW - 1) / CW + NW - 1) / NW;
  const long cap = (long)ncu * per_cu;
  const int grid = (int)(wgs < cap ? wgs : cap);
  hipLaunchKernelGGL((vn_pgrad16_kernel<L, KS, TANH>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

}  // namespace

// the instantiations (vn_points16.h): in the product library the networks the bf16-piece kernels do NOT serve
#define VN_PGRAD16_CASES(X) VN_POINT16_F32_CASES(X)

hipError_t vn_pgrad16_launch(const VnNet& net, const float* theta, const float* X, long n, float* out_u, float* out_g,
                             float* out_pack, int ncu, int wgs_per_cu, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (((out_g || out_pack) && net.dim > 3) || net.d_in > 4 * KS0) return hipErrorInvalidValue;
  VnPgradArgsD a;
  a.net = net; a.theta = theta; a.X = X; a.n = n; a.out_u = out_u; a.out_g = out_g; a.out_pack = out_pack;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK)                                                                              \
  if (net.L == LL && ks == KK)                                                                  \
    return net.act == VN_ACT_TANH ? launch_one<LL, KK, true>(a, ncu, wgs_per_cu, s) : launch_one<LL, KK, false>(a, ncu, wgs_per_cu, s);
  VN_PGRAD16_CASES(X)
#undef X
  return hipErrorInvalidValue;
}
