// Strong PDE residual  res = -u_t + kappa Lap(u) - (v - grad kappa) . grad(u) + s  (TFModel.py:543-545, 743-754) at a set
// of points, on the matrix pipe: second-order forward ("Taylor") mode.  The reference takes tf.gradients twice; the
// per-point kernel of vn_pointwise.hip carries (value, d first, dim second) derivative arrays per thread and runs at 1 TFLOP/s
// (106 ms for 10^6 points at 5x50: every monitor of the training loop, VarNet.py:1363, and every residual-driven
// re-sampling, VarNet.py:1696-1868, waits on it).  Here a wave carries 16 points through the network once per coordinate
// direction e_d with THREE chained streams per layer
//     z   = W^T a + b,            a'   = sigma(z)
//     z.  = W^T a.                a.'  = sigma'(z) z.
//     z.. = W^T a..               a..' = sigma''(z) z.^2 + sigma'(z) z..
// so that one pass yields u, du/dx_d and d2u/dx_d^2 (the time direction needs the first two only: its third stream is
// branched over).  3 dim + 2 streams of F_pt per point in all (the value stream is recomputed per direction: carrying all
// 2 dim + 2 streams through one sweep needs 192 accumulator registers).  Geometry, layout and weight images of vn_pgrad16.hip
// (8 waves, 16 points per wave, v_mfma_f32_16x16x4_f32, layers chained in registers, no workgroup barrier after the
// prologue); nothing is stored between layers, so the kernel is lean in registers.
#include "vn_points16.h"
#include "vn_taylor16.h"

#include <atomic>

namespace {
using namespace vn16;

struct VnTaylorArgsD {
  VnNet net;
  const float* theta;
  const float* X;            // [n, d_in]
  const float* diff;         // [n]
  const float* vel;          // [n, dim]
  const float* src;          // [n] or nullptr
  const float* ddx;          // [n, dim] or nullptr (grad kappa)
  int td;                    // time dependent: the coordinate behind the spatial ones is t
  long n;
  float* u;                  // [n] or nullptr
  float* res;                // [n]
};

template <int L, int KS, bool TANH>
#if defined(__HIP_DEVICE_COMPILE__)
#define VN_NO_LDS_PAIRING __attribute__((target("no-load-store-opt")))      // see vn_fused16.hip
#else
#define VN_NO_LDS_PAIRING
#endif
__global__ __launch_bounds__(NTHREADS, 2) VN_NO_LDS_PAIRING void vn_taylor16_kernel(VnTaylorArgsD A) {
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
  const int dim = net.dim, nd1 = dim + (A.td ? 1 : 0);

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
    for (int d = 0; d < nd1; ++d) {                  // one pass per coordinate direction e_d (d == dim: time)
      asm volatile("" ::: "memory");                 // keep LDS fragment loads inside the loop
      const bool second = d < dim;                   // wave-uniform: the time direction needs no second derivative
      float gin[KS0];
#pragma unroll
      for (int s = 0; s < KS0; ++s) gin[s] = (4 * s + g == d) ? 1.f : 0.f;
      // ---------------------------------------------------------------- layer 1: z.. = 0 (an affine map has no curvature)
      f32x4 pv[MT], pt[MT], p2[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        pv[m] = *reinterpret_cast<const f32x4a*>(&BI[m * 16 + g * 4]);
        pt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        p2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        if (4 * s < net.d_in) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float wf = W1[4 * s * WS + offF + 16 * m];
            pv[m] = mfma16(wf, xin[s], pv[m]);
            pt[m] = mfma16(wf, gin[s], pt[m]);
          }
        }
      }
      // ---------------------------------------------------------------- hidden layers
#pragma unroll
      for (int l = 2; l <= L; ++l) {
        const float* Wl = WH + (l - 2) * LY::HPWS;
        int k_in = (net.H[l - 1] + 3) >> 2, m_out = (net.H[l] + 15) >> 4;
        f32x4 nv[MT], nt[MT], n2[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          nv[m] = *reinterpret_cast<const f32x4a*>(&BI[(l - 1) * 64 + m * 16 + g * 4]);
          nt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
          n2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // activation of the previous layer pipelined under this layer's MFMAs in pairs of k-steps (vn_fused16.hip): stage A
        // (packed scale + 2 v_exp) of pair j+3 after the first k-step of pair j, stage B (packed 1+e + 2 v_rcp) of pair j+2 and
        // stage C (the two derivative streams of pair j+1) after the second
        constexpr int NP = PA<KS>::NP;
        auto pr = [](const f32x4 (&t)[MT], int j) { return f32x2{t[(2 * j) >> 2][(2 * j) & 3], t[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
        auto second_of = [&](f32x2 s, f32x2 zd, f32x2 z2) {          // a..' = sigma'(z) ((sigma''/sigma')(z) z.^2 + z..)
          return act_d1_2<TANH>(s) * (act_d2r_2<TANH>(s) * zd * zd + z2);
        };
        float wf[MTM > 0 ? MTM : 1], we[NVE], ev[NVE], et[NVE], e2v[NVE];
#pragma unroll
        for (int m = 0; m < MTM; ++m) wf[m] = Wl[offF + 16 * m];
#pragma unroll
        for (int v = 0; v < NVE; ++v) {
          we[v] = EDGE ? Wl[offF - c + EPOS + 4 * v] : 0.f;
          ev[v] = 0.f; et[v] = 0.f; e2v[v] = 0.f;
        }
        f32x2 cs2 = act_fin2<TANH>(act_exp2<TANH>(pr(pv, 0)));
        f32x2 cq2 = act_d1_2<TANH>(cs2) * pr(pt, 0);
        f32x2 cw2 = second_of(cs2, pr(pt, 0), pr(p2, 0));
        f32x2 s1 = (NP > 1) ? act_fin2<TANH>(act_exp2<TANH>(pr(pv, 1))) : f32x2{0.f, 0.f};
        f32x2 e2 = (NP > 2) ? act_exp2<TANH>(pr(pv, 2)) : f32x2{0.f, 0.f};
        f32x2 e3 = {0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int j = ks >> 1;
          float wn[MTM > 0 ? MTM : 1], wen[NVE];
#pragma unroll
          for (int m = 0; m < MTM; ++m) wn[m] = (ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + offF + 16 * m] : 0.f;
#pragma unroll
          for (int v = 0; v < NVE; ++v) wen[v] = (EDGE && ks + 1 < KS) ? Wl[4 * (ks + 1) * WS + offF - c + EPOS + 4 * v] : 0.f;
          const float cs = cs2[ks & 1], cq = cq2[ks & 1], cw = cw2[ks & 1];
          __builtin_amdgcn_sched_barrier(0);
          if (live_k(ks, k_in)) {
#pragma unroll
            for (int m = 0; m < MTM; ++m) {
              if (!live_m(m, m_out)) continue;
              nv[m] = mfma16(wf[m], cs, nv[m]);
              nt[m] = mfma16(wf[m], cq, nt[m]);
              if (second) n2[m] = mfma16(wf[m], cw, n2[m]);
            }
          }
          if (EDGE) {
#pragma unroll
            for (int v = 0; v < NVE; ++v) { ev[v] += we[v] * cs; et[v] += we[v] * cq; e2v[v] += we[v] * cw; }
          }
          if ((ks & 1) == 0) {
            if (j + 3 < NP) e3 = act_exp2<TANH>(pr(pv, j + 3));
            __builtin_amdgcn_sched_barrier(0);
          } else {
            f32x2 s2 = s1, q1 = cq2, w1 = cw2;
            if (j + 2 < NP) s2 = act_fin2<TANH>(e2);
            if (j + 1 < NP) {
              const f32x2 zz = pr(pt, j + 1);
              q1 = act_d1_2<TANH>(s1) * zz;
              w1 = second_of(s1, zz, pr(p2, j + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            cs2 = s1; cq2 = q1; cw2 = w1; s1 = s2; e2 = e3;
          }
#pragma unroll
          for (int m = 0; m < MTM; ++m) wf[m] = wn[m];
#pragma unroll
          for (int v = 0; v < NVE; ++v) we[v] = wen[v];
        }
        if (EDGE) {                                   // sum the four lane groups' shares; group g keeps edge feature g
          nv[MT - 1][0] += edge_reduce_scatter<NVE>(ev, g);          // bias was loaded above
          nt[MT - 1][0] = edge_reduce_scatter<NVE>(et, g);
          n2[MT - 1][0] = edge_reduce_scatter<NVE>(e2v, g);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) { pv[m] = nv[m]; pt[m] = nt[m]; p2[m] = n2[m]; }
      }
      // ---------------------------------------------------------------- last activation + output layer (VALU)
      auto pairOf = [](const f32x4 (&t)[MT], int j) { return f32x2{t[(2 * j) >> 2][(2 * j) & 3], t[(2 * j) >> 2][((2 * j) & 3) + 1]}; };
      f32x2 u2 = {0.f, 0.f}, ud2 = {0.f, 0.f}, uw2 = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < PA<KS>::NP; ++j) {
        const bool full = 2 * j + 1 < KS;
        const f32x2 wv = {WO[4 * (2 * j) + g], full ? WO[4 * (2 * j + 1) + g] : 0.f};
        const f32x2 av = act_fin2<TANH>(act_exp2<TANH>(pairOf(pv, j)));
        const f32x2 zd = pairOf(pt, j);
        const f32x2 sp = act_d1_2<TANH>(av);
        u2 += wv * av;
        ud2 += wv * (sp * zd);
        uw2 += wv * (sp * (act_d2r_2<TANH>(av) * zd * zd + pairOf(p2, j)));
      }
      const float u = rowsum4(u2[0] + u2[1]) + bo;
      const float ud = rowsum4(ud2[0] + ud2[1]);
      const float uw = rowsum4(uw2[0] + uw2[1]);
      uval = u;
      if (second) {                                  // TFModel.py:750-754
        lap += uw;
        float vd = 0.f;
        if (valid) {
          vd = A.vel[row * dim + d];
          if (A.ddx) vd -= A.ddx[row * dim + d];
        }
        adv += vd * ud;
      } else {
        ut = ud;
      }
    }
    if (valid && g == 0) {
      float out = A.td ? -ut : 0.f;
      out += A.diff[row] * lap;
      out -= adv;
      if (A.src) out += A.src[row];
      if (A.u) A.u[row] = uval;
      A.res[row] = out;
    }
  }
}

template <int L, int KS, bool TANH>
hipError_t launch_one(const VnTaylorArgsD& a, int ncu, hipStream_t s) {
  const size_t bytes = (size_t)PLay<L, KS>::TOTAL * sizeof(float);
  // the attribute is per device and sticky: set it once per device (bit mask; engines on different devices may be
  // driven from different threads)
  static std::atomic<unsigned long long> attr_done{0};
  static std::atomic<int> occ{0};                    // workgroups of this instantiation a CU holds (registers, LDS)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_taylor16_kernel<L, KS, TANH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    int nb = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)vn_taylor16_kernel<L, KS, TANH>, NTHREADS, bytes) != hipSuccess) {
      (void)hipGetLastError();
      nb = 1;
    }
    occ.store(nb < 1 ? 1 : nb, std::memory_order_relaxed);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int per_cu = occ.load(std::memory_order_relaxed) >= 2 ? 2 : 1;       // waves are independent: see vn_pgrad16.hip
  const long wgs = ((a.n + CW - 1) / CW + NW - 1) / NW;
  const long cap = (long)ncu * per_cu;
  const int grid = (int)(wgs < cap ? wgs : cap);
  hipLaunchKernelGGL((vn_taylor16_kernel<L, KS, TANH>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

}  // namespace

// the instantiations (vn_points16.h): in the product library the networks the bf16-piece kernels do NOT serve
#define VN_TAYLOR16_CASES(X) VN_POINT16_F32_CASES(X)

hipError_t vn_taylor16_residual(const VnNet& net, const float* theta, const float* X, const float* diff, const float* vel,
                                const float* src, const float* ddx, int td, long n, float* u, float* res, int ncu, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  static_assert(4 * KS0 == 8, "vn_taylor16_supported states the input width");
  if (!vn_taylor16_supported(net, td)) return hipErrorInvalidValue;
  VnTaylorArgsD a;
  a.net = net; a.theta = theta; a.X = X; a.diff = diff; a.vel = vel; a.src = src; a.ddx = ddx; a.td = td; a.n = n; a.u = u; a.res = res;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK)                                                                              \
  if (net.L == LL && ks == KK)                                                                  \
    return net.act == VN_ACT_TANH ? launch_one<LL, KK, true>(a, ncu, s) : launch_one<LL, KK, false>(a, ncu, s);
  VN_TAYLOR16_CASES(X)
#undef X
  return hipErrorInvalidValue;
}
