// fp64 forms of the point kernels on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64): the model value (vn_forward_f64) and the
// strong residual  -u_t + kappa Lap(u) - (v - grad kappa) . grad(u) + s  (vn_residual_f64; TFModel.py:543-545, 743-754) in
// second-order forward mode -- the algorithm of vn_taylor16.hip in double precision.  BASELINE config 5 checks its fp64 residual
// against the fp64 oracle (<= 1e-10); the per-thread kernels of vn_pointwise.hip that served it carry their derivative arrays in
// scratch (253 ms for 10^6 points at 5x50, 36 ms for the value alone).
//
// Kept deliberately plain (this is a checking path, not a training path): layers chain in registers as in the fp32 kernels --
// feature f in k-step f/4, lane group f%4 -- but the fp64 MFMA interleaves the rows of a lane group (register i of lane group g
// holds row 4i + g of the tile, where the fp32 16x16x4 holds row 4g + i), so here the accumulator row of a feature is the
// feature index itself and the images need no column permutation; weight images in LDS as doubles (networks whose images
// exceed the 160 KB fall back to the per-thread kernels), every row tile on the matrix pipe (no edge rows on the VALU, no
// padding branches, no software pipeline), libm exp / tanh and a true division for the activation.
#include "vn_points16.h"
#include "vn_taylor16.h"

#include <atomic>

namespace {
using namespace vn16;

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef f64x4 f64x4a __attribute__((may_alias));

template <int L, int KS>
struct DLay {                                               // offsets in doubles
  static constexpr int HP = 4 * KS;
  static constexpr int HPWS = al4(HP * WS);
  static constexpr int W1_OFF = 0;                          // [8][WS]
  static constexpr int WH_OFF = al4(8 * WS);                // [L-1][HP][WS]
  static constexpr int BI_OFF = WH_OFF + (L - 1) * HPWS;    // [L][64] biases by feature
  static constexpr int WO_OFF = BI_OFF + L * 64;            // [4*KS]
  static constexpr int TOTAL = WO_OFF + al4(4 * KS);
  static constexpr size_t BYTES = (size_t)TOTAL * sizeof(double);
};

struct VnTaylorDArgs {
  VnNet net;
  const double* theta;
  const double* X;           // [n, d_in]
  const double* diff;        // [n]         (residual only)
  const double* vel;         // [n, dim]
  const double* src;         // [n] or nullptr
  const double* ddx;         // [n, dim] or nullptr
  int td;
  long n;
  double* u;                 // [n] or nullptr
  double* res;               // [n], or nullptr: value only (vn_forward_f64)
};

template <bool TANH>
__device__ __forceinline__ double actd(double z) { return TANH ? tanh(z) : 1.0 / (1.0 + exp(-z)); }
template <bool TANH>
__device__ __forceinline__ double actd_d1(double a) { return TANH ? 1.0 - a * a : a * (1.0 - a); }
template <bool TANH>
__device__ __forceinline__ double actd_d2r(double a) { return TANH ? -2.0 * a : 1.0 - 2.0 * a; }      // sigma'' / sigma'

__device__ __forceinline__ f64x4 mfma16d(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double rowsum4d(double x) {       // over the four 16-lane rows of the wave, in every lane
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

template <int L, int KS, bool TANH>
__global__ __launch_bounds__(NTHREADS, 2) void vn_taylor16d_kernel(VnTaylorDArgs A) {
  using LY = DLay<L, KS>;
  constexpr int MT = mtiles(KS);
  extern __shared__ __attribute__((aligned(16))) double ldsd[];
  const VnNet& net = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* W1 = ldsd + LY::W1_OFF;
  double* WH = ldsd + LY::WH_OFF;
  double* BI = ldsd + LY::BI_OFF;
  double* WO = ldsd + LY::WO_OFF;
  // ---- prologue: weight images [in-feature][out-position] (row stride WS), padding exact zeros
  for (int i = tid; i < LY::TOTAL; i += NTHREADS) ldsd[i] = 0.0;
  __syncthreads();
  {
    const int H1 = net.H[1];
    for (int j = tid; j < net.d_in * H1; j += NTHREADS) {
      const int k = j / H1, f = j - k * H1;
      W1[k * WS + f] = A.theta[net.woff[1] + j];
    }
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      double* Wl = WH + (l - 2) * LY::HPWS;
      const int Hout = net.H[l], n = net.H[l - 1] * Hout;
      for (int j = tid; j < n; j += NTHREADS) {
        const int k = j / Hout, f = j - k * Hout;
        Wl[k * WS + f] = A.theta[net.woff[l] + j];
      }
    }
    for (int j = tid; j < L * 64; j += NTHREADS) {               // biases by feature: [L][64]
      const int l = j / 64 + 1, f = j % 64;
      if (f < net.H[l]) BI[j] = A.theta[net.boff[l] + f];
    }
    for (int j = tid; j < net.H[L]; j += NTHREADS) WO[j] = A.theta[net.woff[L + 1] + j];
  }
  const double bo = A.theta[net.boff[L + 1]];
  __syncthreads();

  const int g = lane >> 4, c = lane & 15;
  const int offF = g * WS + c;
  const int dim = net.dim;
  const bool value_only = A.res == nullptr;
  const int npass = value_only ? 1 : dim + (A.td ? 1 : 0);
  const long nchunks = (A.n + CW - 1) / CW;
  for (long chunk = (long)blockIdx.x * NW + wave; chunk < nchunks; chunk += (long)gridDim.x * NW) {
    const long row = chunk * CW + c;
    const bool valid = row < A.n;
    double xin[KS0];
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int f = 4 * s + g;
      xin[s] = (valid && f < net.d_in) ? A.X[row * net.d_in + f] : 0.0;
    }
    double uval = 0.0, lap = 0.0, adv = 0.0, ut = 0.0;
#pragma unroll 1
    for (int d = 0; d < npass; ++d) {                // one pass per coordinate direction e_d (d == dim: time)
      asm volatile("" ::: "memory");
      const bool first = !value_only;                // wave-uniform: derivative streams wanted at all
      const bool second = first && d < dim;          // ... and the second derivative (not for the time direction)
      double gin[KS0];
#pragma unroll
      for (int s = 0; s < KS0; ++s) gin[s] = (first && 4 * s + g == d) ? 1.0 : 0.0;
      f64x4 pv[MT], pt[MT], p2[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        pv[m] = f64x4{BI[m * 16 + g], BI[m * 16 + 4 + g], BI[m * 16 + 8 + g], BI[m * 16 + 12 + g]};     // register i: feature 16m + 4i + g
        pt[m] = f64x4{0.0, 0.0, 0.0, 0.0};
        p2[m] = f64x4{0.0, 0.0, 0.0, 0.0};
      }
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        if (4 * s < net.d_in) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const double wf = W1[4 * s * WS + offF + 16 * m];
            pv[m] = mfma16d(wf, xin[s], pv[m]);
            if (first) pt[m] = mfma16d(wf, gin[s], pt[m]);
          }
        }
      }
#pragma unroll
      for (int l = 2; l <= L; ++l) {
        const double* Wl = WH + (l - 2) * LY::HPWS;
        f64x4 nv[MT], nt[MT], n2[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const double* bl = &BI[(l - 1) * 64 + m * 16 + g];
          nv[m] = f64x4{bl[0], bl[4], bl[8], bl[12]};
          nt[m] = f64x4{0.0, 0.0, 0.0, 0.0};
          n2[m] = f64x4{0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const double z = pv[ks >> 2][ks & 3], zd = pt[ks >> 2][ks & 3], z2 = p2[ks >> 2][ks & 3];
          const double a = actd<TANH>(z);
          const double s1 = actd_d1<TANH>(a);
          const double ad = s1 * zd;
          const double a2 = s1 * (actd_d2r<TANH>(a) * zd * zd + z2);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const double wf = Wl[4 * ks * WS + offF + 16 * m];
            nv[m] = mfma16d(wf, a, nv[m]);
            if (first) nt[m] = mfma16d(wf, ad, nt[m]);
            if (second) n2[m] = mfma16d(wf, a2, n2[m]);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) { pv[m] = nv[m]; pt[m] = nt[m]; p2[m] = n2[m]; }
      }
      double u = 0.0, ud = 0.0, uw = 0.0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double wv = WO[4 * ks + g];
        const double z = pv[ks >> 2][ks & 3], zd = pt[ks >> 2][ks & 3], z2 = p2[ks >> 2][ks & 3];
        const double a = actd<TANH>(z);
        const double s1 = actd_d1<TANH>(a);
        u += wv * a;
        ud += wv * (s1 * zd);
        uw += wv * (s1 * (actd_d2r<TANH>(a) * zd * zd + z2));
      }
      u = rowsum4d(u) + bo;
      ud = rowsum4d(ud);
      uw = rowsum4d(uw);
      uval = u;
      if (second) {                                  // TFModel.py:750-754
        lap += uw;
        double vd = 0.0;
        if (valid) {
          vd = A.vel[row * dim + d];
          if (A.ddx) vd -= A.ddx[row * dim + d];
        }
        adv += vd * ud;
      } else if (first) {
        ut = ud;
      }
    }
    if (valid && g == 0) {
      if (A.u) A.u[row] = uval;
      if (!value_only) {
        double out = A.td ? -ut : 0.0;
        out += A.diff[row] * lap;
        out -= adv;
        if (A.src) out += A.src[row];
        A.res[row] = out;
      }
    }
  }
}

template <int L, int KS, bool TANH>
hipError_t launch_one(const VnTaylorDArgs& a, int ncu, hipStream_t s) {
  constexpr size_t bytes = DLay<L, KS>::BYTES;
  if (bytes > 160 * 1024) return hipErrorInvalidValue;       // images beyond the LDS: the caller falls back to the per-thread kernels
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)vn_taylor16d_kernel<L, KS, TANH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const long wgs = ((a.n + CW - 1) / CW + NW - 1) / NW;
  const int grid = (int)(wgs < ncu ? wgs : ncu);
  hipLaunchKernelGGL((vn_taylor16d_kernel<L, KS, TANH>), dim3(grid), dim3(NTHREADS), bytes, s, a);
  return hipGetLastError();
}

template <int L, int KS>
constexpr bool fits() { return DLay<L, KS>::BYTES <= 160 * 1024; }

}  // namespace

// every network of the 8-wave family whose double-precision images fit the LDS
#define VN_TAYLOR16D_CASES(X) \
  X(1, 5) X(2, 5) X(3, 5) X(4, 5) X(5, 5) X(6, 5) X(7, 5) X(8, 5)  \
  X(1, 8) X(2, 8) X(3, 8) X(4, 8) X(5, 8) X(6, 8) X(7, 8) X(8, 8)  \
  X(1, 13) X(2, 13) X(3, 13) X(4, 13) X(5, 13) X(6, 13)  \
  X(1, 16) X(2, 16) X(3, 16) X(4, 16) X(5, 16)

bool vn_taylor16d_supported(const VnNet& net) {
  if (!vn_fused16_net_supported(net) || net.dim > 3 || net.d_in > 4 * KS0) return false;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK) if (net.L == LL && ks == KK) return fits<LL, KK>();
  VN_TAYLOR16D_CASES(X)
#undef X
  return false;
}

hipError_t vn_taylor16d_launch(const VnNet& net, const double* theta, const double* X, const double* diff, const double* vel,
                               const double* src, const double* ddx, int td, long n, double* u, double* res, int ncu, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (!vn_taylor16d_supported(net) || (res && net.dim + (td ? 1 : 0) > net.d_in)) return hipErrorInvalidValue;
  VnTaylorDArgs a;
  a.net = net; a.theta = theta; a.X = X; a.diff = diff; a.vel = vel; a.src = src; a.ddx = ddx; a.td = td; a.n = n; a.u = u; a.res = res;
  const int ks = vn_fused16_ks(net);
#define X(LL, KK)                                                                              \
  if (net.L == LL && ks == KK)                                                                  \
    return net.act == VN_ACT_TANH ? launch_one<LL, KK, true>(a, ncu, s) : launch_one<LL, KK, false>(a, ncu, s);
  VN_TAYLOR16D_CASES(X)
#undef X
  return hipErrorInvalidValue;
}
