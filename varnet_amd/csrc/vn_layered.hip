// Layer-by-layer route for networks outside the range of the fused / generic kernels (more than VN_KMAX_LAYERS hidden
// layers, hidden widths above VN_KMAX_WIDTH, more than VN_KMAX_DIN inputs).  The reference accepts any `layerWidth`
// (TFModel.py:208-221); this keeps the constructor drop-in for such networks at the price of HBM round trips.
//
// Formulation (the same (value, one tangent) recurrences as the kernels, oracle/tangent_ref.py):
//   forward   z = a_{l-1} W_l + b_l,  zd = ad_{l-1} W_l,   a_l = act(z),  ad_l = act'(z) zd
//   reverse   zbar = s1 abar + s2r ad adbar,  zdbar = s1 adbar          (s1 = act'(z), s2r = act''/act', both functions of a;
//             act''(z) zd = s2r s1 zd = s2r ad, so the reverse pass needs only the stored (a, ad), not zd)
//             abar_{l-1} = zbar W_l^T,  adbar_{l-1} = zdbar W_l^T
//             dW_l = a_{l-1}^T zbar + ad_{l-1}^T zdbar,   db_l = sum_rows zbar
// Data layout: the streams of a chunk of rows are STACKED along the row axis -- buffer [S][n][H], read by the GEMMs as
// one (S n) x H row-major matrix -- so every layer is ONE GEMM forward and TWO in the reverse pass, whatever S is
// (S = 1: BC/IC rows and vn_forward; S = 2: interior rows; S = 1 + nd1 + dim: the strong residual with its first and
// second derivative streams).  The GEMMs are the MFMA kernels of vn_gemm.hip (fp32 with the forward epilogue fused, fp64 for
// the fp64 entry points): no vendor library is linked, loaded or named here (tests/test_layered_gpu.py checks the products
// against library GEMMs from the outside).  Everything between the GEMMs is hand-written below; all reductions have a fixed
// order, results are run-to-run reproducible.
// Rows are processed in chunks sized to a fixed workspace.  The seeds need R_k of whole test functions first, so a gradient
// evaluation is forward (all rows) -> seed kernel -> reverse (all rows): the forward KEEPS the activations of all rows in
// HBM when they fit half of the free memory (6.4 M rows of a 3 x 256 net: 39 GB of the 288) and the reverse pass reads
// them -- 6 F_pt per interior point; otherwise the reverse pass recomputes them per chunk (8 F_pt).
// Networks up to 256 wide do not run their training passes here: vn_wide.hip carries tiles of 32 points through all layers
// with the activations in LDS (3x faster; VN_LAYERED_NOWIDE=1 keeps them on the GEMMs, which also remain the fallback
// when the stored activations do not fit, and serve the residual and fp64 entry points).
#include "vn_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

int lfail(char* err, size_t n, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (err && n) vsnprintf(err, n, fmt, ap);
  va_end(ap);
  return 1;
}

#define LHIP(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return lfail(err, errlen, "%s: %s", #expr, hipGetErrorString(e_));    \
  } while (0)
// VN_LAYERED_TRACE=1: stage markers on stderr with a stream sync in front of each (diagnosis of a fault: which stage)
static bool g_trace = [] { const char* t = getenv("VN_LAYERED_TRACE"); return t && *t && *t != '0'; }();
#define LTRACE(s_, ...)                                                        \
  do {                                                                         \
    if (g_trace) {                                                             \
      hipError_t te_ = hipStreamSynchronize(s_);                               \
      fprintf(stderr, "[vn_layered] sync=%d ", (int)te_);                      \
      fprintf(stderr, __VA_ARGS__);                                            \
      fprintf(stderr, "\n");                                                   \
      fflush(stderr);                                                          \
    }                                                                          \
  } while (0)

// ---- elementwise kernels -------------------------------------------------------------------------------------------
constexpr int EB = 256;

template <typename T> __device__ __forceinline__ T act_f(T z, int act);
template <> __device__ __forceinline__ float act_f<float>(float z, int act) {
  return act == VN_ACT_TANH ? tanhf(z) : 1.0f / (1.0f + expf(-z));
}
template <> __device__ __forceinline__ double act_f<double>(double z, int act) {
  return act == VN_ACT_TANH ? tanh(z) : 1.0 / (1.0 + exp(-z));
}
template <typename T> __device__ __forceinline__ T act_s1(T a, int act) { return act == VN_ACT_TANH ? T(1) - a * a : a * (T(1) - a); }
template <typename T> __device__ __forceinline__ T act_s2r(T a, int act) { return act == VN_ACT_TANH ? T(-2) * a : T(1) - T(2) * a; }

// layer-0 streams of the training pass: stream 0 = X rows, stream 1 = the tangent direction (G in the first `dim`
// inputs, zero elsewhere: the derivative is taken along the spatial coordinates only, TFModel.py:536-545)
__global__ __launch_bounds__(EB) void k_pack_train(const float* __restrict__ X, const float* __restrict__ G, long n, int d_in,
                                                  int dim, int S, float* __restrict__ out) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i >= n * d_in) return;
  const long r = i / d_in;
  const int k = (int)(i % d_in);
  out[i] = X[i];
  if (S == 2) out[n * d_in + i] = (k < dim) ? G[r * dim + k] : 0.f;
}

// adjoints of the last hidden layer: abar = ubar w_o^T, adbar = udbar w_o^T
__global__ __launch_bounds__(EB) void k_seed_outer(const float* __restrict__ ubar, const float* __restrict__ udbar,
                                                  const float* __restrict__ wo, long n, int H, int S, float* __restrict__ out) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i >= n * H) return;
  const long r = i / H;
  const float w = wo[i % H];
  out[i] = ubar[r] * w;
  if (S == 2) out[n * H + i] = udbar[r] * w;
}

// both at once for the last hidden layer: (zbar, zdbar)_L straight from the seeds and the stored (a, ad)_L -- the adjoints
// (abar, adbar) = (ubar, udbar) w_o^T are never written
__global__ __launch_bounds__(EB) void k_seed_act_bwd(const float* __restrict__ ubar, const float* __restrict__ udbar,
                                                    const float* __restrict__ wo, const float* __restrict__ A, long n, int H, int S,
                                                    int act, float* __restrict__ out) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i >= n * H) return;
  const long r = i / H;
  const float w = wo[i % H];
  const float a = A[i];
  const float s1 = act_s1<float>(a, act);
  float zbar = s1 * (ubar[r] * w);
  if (S == 2) {
    const float adbar = udbar[r] * w;
    zbar += act_s2r<float>(a, act) * A[n * H + i] * adbar;
    out[n * H + i] = s1 * adbar;
  }
  out[i] = zbar;
}

// (abar, adbar) -> (zbar, zdbar) in place; A = the layer's stored (a, ad)
__global__ __launch_bounds__(EB) void k_act_bwd(float* __restrict__ B, const float* __restrict__ A, long n, int H, int S, int act) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i >= n * H) return;
  const float a = A[i];
  const float s1 = act_s1<float>(a, act);
  float zbar = s1 * B[i];
  if (S == 2) {
    const float adbar = B[n * H + i];
    zbar += act_s2r<float>(a, act) * A[n * H + i] * adbar;
    B[n * H + i] = s1 * adbar;
  }
  B[i] = zbar;
}

// out[r] = (bias ? *bias : 0) + sum_c A[r][c] w[c]: one wave per row, fixed-order butterfly (model value / directional
// derivative from the last hidden layer; the library gemv runs this shape at 0.5 TB/s)
__global__ __launch_bounds__(EB) void k_rowdot(const float* __restrict__ A, const float* __restrict__ wv, const float* __restrict__ bias,
                                              long n, int H, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (EB / 64) + (threadIdx.x >> 6), nw = (long)gridDim.x * (EB / 64);
  for (long r = wave; r < n; r += nw) {
    float acc = 0.f;
    for (int c = lane; c < H; c += 64) acc += A[r * H + c] * wv[c];
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[r] = acc + (bias ? *bias : 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(EB) void k_fill(T* __restrict__ p, long n, T v) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i < n) p[i] = v;
}
// fill from a device scalar (the output bias)
template <typename T>
__global__ __launch_bounds__(EB) void k_fill_from(T* __restrict__ p, long n, const T* __restrict__ v) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i < n) p[i] = *v;
}

// ---- strong residual: streams 0 = value, 1..nd1 = first derivatives, nd1+1..nd1+dim = second derivatives --------------
template <typename T>
__global__ __launch_bounds__(EB) void k_pack_res(const T* __restrict__ X, long n, int d_in, int nd1, int S, T* __restrict__ out) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i >= n * d_in) return;
  const int k = (int)(i % d_in);
  out[i] = X[i];
  for (int s = 1; s < S; ++s) out[(long)s * n * d_in + i] = (s <= nd1 && k == s - 1) ? T(1) : T(0);
}

template <typename T>
__global__ __launch_bounds__(EB) void k_act_res(T* __restrict__ A, const T* __restrict__ bias, long n, int H, int nd1, int dim,
                                               int act) {
  const long i = (long)blockIdx.x * EB + threadIdx.x;
  if (i >= n * H) return;
  const long st = n * H;
  const T s = act_f<T>(A[i] + bias[i % H], act);
  const T s1 = act_s1<T>(s, act);
  const T s2 = s1 * act_s2r<T>(s, act);
  A[i] = s;
  for (int d = 0; d < dim; ++d) {
    const T z1 = A[(1 + d) * st + i];
    A[(1 + nd1 + d) * st + i] = s2 * z1 * z1 + s1 * A[(1 + nd1 + d) * st + i];
  }
  for (int d = 0; d < nd1; ++d) A[(1 + d) * st + i] *= s1;
}

// y = [val | g_0.. | lap_0..] (S x n, from the output gemv) -> u, residual (TFModel.py:750-754)
template <typename T>
__global__ __launch_bounds__(EB) void k_res_combine(const T* __restrict__ y, const T* __restrict__ bo, const T* __restrict__ diff,
                                                   const T* __restrict__ vel, const T* __restrict__ src,
                                                   const T* __restrict__ ddx, int td, long n, int dim, int nd1,
                                                   T* __restrict__ u, T* __restrict__ res) {
  const long r = (long)blockIdx.x * EB + threadIdx.x;
  if (r >= n) return;
  T lap = T(0);
  for (int d = 0; d < dim; ++d) lap += y[(1 + nd1 + d) * n + r];
  T out = td ? -y[(1 + dim) * n + r] : T(0);
  out += diff[r] * lap;
  for (int d = 0; d < dim; ++d) {
    const T dd = ddx ? ddx[r * dim + d] : T(0);
    out -= (vel[r * dim + d] - dd) * y[(1 + d) * n + r];
  }
  if (src) out += src[r];
  if (u) u[r] = y[r] + *bo;
  res[r] = out;
}

// ---- reductions over the rows of a chunk (hand-written: a library gemv / TN gemm sees a tiny output and millions of
// rows to sum, and without atomics runs them on a handful of workgroups -- 83 % of the route's time when first measured)
constexpr int RS_ROWS = 2048;          // rows per workgroup of the column-sum kernel

// part[b][c] = sum over the rows r of block b of A[r][c] * (x ? x[r] : 1)      (A row-major n x H).  A workgroup is 64
// columns x 4 row lanes (grid.y = column groups): each wave streams 256 contiguous bytes per row, four rows of it in
// flight per lane; the four row lanes meet in LDS in a fixed order.
__global__ __launch_bounds__(EB) void k_colsum_part(const float* __restrict__ A, const float* __restrict__ x, long n, int H,
                                                   float* __restrict__ part) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + tx;
  const long r0 = (long)blockIdx.x * RS_ROWS;
  const long r1 = r0 + RS_ROWS < n ? r0 + RS_ROWS : n;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < H) {
    long r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = A[(r + 4 * k) * H + c];
        acc[k] += x ? v * x[r + 4 * k] : v;
      }
    }
    for (; r < r1; r += 4) acc[0] += x ? A[r * H + c] * x[r] : A[r * H + c];
  }
  red[ty][tx] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (ty == 0 && c < H) part[(long)blockIdx.x * H + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// the same with 16-byte loads (H a multiple of 4, rows 16-byte aligned): a workgroup is 64 column quads x 4 row lanes over
// RSV_ROWS rows -- a wave streams 1 KB per row instead of 256 B, and four times as many workgroups are in flight
constexpr int RSV_ROWS = 512;
typedef float lf32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(EB) void k_colsum_part_v4(const float* __restrict__ A, const float* __restrict__ x, long n, int H,
                                                      float* __restrict__ part) {
  __shared__ lf32x4 red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + 4 * tx;
  const long r0 = (long)blockIdx.x * RSV_ROWS;
  const long r1 = r0 + RSV_ROWS < n ? r0 + RSV_ROWS : n;
  lf32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = lf32x4{0.f, 0.f, 0.f, 0.f};
  if (c < H) {
    long r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const lf32x4 v = *(const lf32x4*)(A + (r + 4 * k) * H + c);
        acc[k] += x ? v * x[r + 4 * k] : v;
      }
    }
    for (; r < r1; r += 4) {
      const lf32x4 v = *(const lf32x4*)(A + r * H + c);
      acc[0] += x ? v * x[r] : v;
    }
  }
  red[ty][tx] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (ty == 0 && c < H) *(lf32x4*)(part + (long)blockIdx.x * H + c) = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// narrow matrices (H < 8): the column lanes of the kernel above would mostly idle; here a workgroup's threads split
// the rows, lane = row, and a fixed-order LDS tree adds them (H == 1: the output-bias gradient, sum of ubar)
__global__ __launch_bounds__(EB) void k_colsum_part_narrow(const float* __restrict__ A, const float* __restrict__ x, long n, int H,
                                                          float* __restrict__ part) {
  __shared__ float red[EB];
  const long r0 = (long)blockIdx.x * RS_ROWS;
  const long r1 = r0 + RS_ROWS < n ? r0 + RS_ROWS : n;
  for (int c = 0; c < H; ++c) {
    float acc = 0.f;
    for (long r = r0 + threadIdx.x; r < r1; r += EB) acc += x ? A[r * H + c] * x[r] : A[r * H + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = EB / 2; sft > 0; sft >>= 1) {
      if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
      __syncthreads();
    }
    if (threadIdx.x == 0) part[(long)blockIdx.x * H + c] = red[0];
    __syncthreads();
  }
}
// dst[i] += sum_b part[b][i], fixed order: 64 elements x 8 groups of partials per workgroup; group g adds the contiguous run
// of partials [g n/8, (g+1) n/8), the eight group sums meet in LDS in a fixed order (one thread per element walking all
// partials left a few workgroups with a chain of thousands of dependent adds)
constexpr int SUMG = 8;
__global__ __launch_bounds__(64 * SUMG) void k_sum_parts(const float* __restrict__ part, int nparts, long len, float* __restrict__ dst) {
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

inline unsigned blocks(long n) { return (unsigned)((n + EB - 1) / EB); }

}  // namespace

struct VnLayered {
  VnNet net{};
  int ncu = 256;
  long sumH = 0;           // sum of H[0..L]
  int hmax_all = 0;        // max of H[0..L]
  // one workspace, carved per call
  void* ws = nullptr;
  size_t ws_bytes = 0;
  float* part = nullptr;   // partial sums of the row reductions
  float* wt = nullptr;     // W_l^T for the input-gradient product (vn_gemm.hip)
  size_t wt_elems = 0;
  size_t part_elems = 0;
  // Activations of a whole row set kept from the forward of a gradient evaluation to its reverse pass (slot 0: interior
  // rows, slot 1: BC/IC rows): 288 GB of HBM hold them for every problem size the kernels' route is measured on, and
  // the reverse pass then skips its forward recompute (6 F_pt per point instead of 8).  Falls back to recompute when
  // they would not fit half of the free memory.
  struct Kept {
    float* buf = nullptr; size_t cap = 0;       // elements
    const float* X = nullptr; long n = 0, c = 0; int S = 0; bool valid = false;
  } kept[2];
  bool never_keep = false;                      // VN_LAYERED_NOKEEP=1: always recompute (tests run both ways)
  VnWide* wide = nullptr;                       // tile kernels for the training passes of nets up to 256 wide (vn_wide.hip)
};

namespace {

constexpr size_t WS_TARGET = (size_t)3 << 30;     // bytes of HBM the route may hold for its chunk buffers

int ensure_ws(VnLayered* w, size_t bytes, char* err, size_t errlen) {
  if (bytes <= w->ws_bytes) return 0;
  if (w->ws) (void)hipFree(w->ws);
  w->ws = nullptr; w->ws_bytes = 0;
  LHIP(hipMalloc(&w->ws, bytes));
  w->ws_bytes = bytes;
  return 0;
}

int ensure_part(VnLayered* w, size_t elems, char* err, size_t errlen) {
  if (elems <= w->part_elems) return 0;
  if (w->part) (void)hipFree(w->part);
  w->part = nullptr; w->part_elems = 0;
  LHIP(hipMalloc((void**)&w->part, elems * sizeof(float)));
  w->part_elems = elems;
  return 0;
}

// dst[0..H) += sum over the n rows of A (n x H, row-major) weighted by x (or 1): two launches, fixed summation order
int colsum_add(VnLayered* w, const float* A, const float* x, long n, int H, float* dst, hipStream_t s, char* err, size_t errlen) {
  const bool v4 = H >= 64 && H % 4 == 0 && ((size_t)A & 15) == 0;
  const int nb = (int)(v4 ? (n + RSV_ROWS - 1) / RSV_ROWS : (n + RS_ROWS - 1) / RS_ROWS);
  if (int rc = ensure_part(w, (size_t)nb * H, err, errlen)) return rc;
  LTRACE(s, "colsum A=%p x=%p n=%ld H=%d nb=%d part=%p dst=%p", (const void*)A, (const void*)x, n, H, nb, (void*)w->part, (void*)dst);
  if (v4) hipLaunchKernelGGL(k_colsum_part_v4, dim3(nb, (H + 255) / 256), dim3(EB), 0, s, A, x, n, H, w->part);
  else if (H >= 8) hipLaunchKernelGGL(k_colsum_part, dim3(nb, (H + 63) / 64), dim3(EB), 0, s, A, x, n, H, w->part);
  else hipLaunchKernelGGL(k_colsum_part_narrow, dim3(nb), dim3(EB), 0, s, A, x, n, H, w->part);
  LHIP(hipGetLastError());
  LTRACE(s, "colsum partial kernel done");
  hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((H + 63) / 64)), dim3(64 * SUMG), 0, s, w->part, nb, (long)H, dst);
  LHIP(hipGetLastError());
  return 0;
}

int ensure_wt(VnLayered* w, size_t elems, char* err, size_t errlen) {
  if (elems <= w->wt_elems) return 0;
  if (w->wt) (void)hipFree(w->wt);
  w->wt = nullptr; w->wt_elems = 0;
  LHIP(hipMalloc((void**)&w->wt, elems * sizeof(float)));
  w->wt_elems = elems;
  return 0;
}

#define LGEMM(expr)                                                                             \
  do {                                                                                          \
    int e_ = (expr);                                                                            \
    if (e_ != 0) return lfail(err, errlen, "%s: %s", #expr, hipGetErrorString((hipError_t)e_)); \
  } while (0)

// dW (Hin x Hout, row-major) += A^T Zbar over M stacked rows.  The rows are cut into groups; one launch of vn_gemm_tn_parts
// gives a partial product per group (parallelism = groups x output tiles instead of output tiles), a fixed-order sum
// adds them up.
int wgrad_add(VnLayered* w, const float* A, const float* Zbar, long M, int Hin, int Hout, float* dW, hipStream_t s, char* err,
              size_t errlen) {
  const long rows = vn_gemm_tn_rows(M, Hin, Hout, w->ncu);     // rows per group that fill the chip evenly
  const int G = (int)(M / rows);                      // full groups; the ragged rest is one more GEMM
  const long rest = M - (long)G * rows;
  const int np = G + (rest > 0 ? 1 : 0);
  const long len = (long)Hin * Hout;
  if (int rc = ensure_part(w, (size_t)np * len, err, errlen)) return rc;
  LGEMM(vn_gemm_tn_parts(A, Zbar, w->part, M, Hin, Hout, rows, s));          // np partial products, one launch
  hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((len + 63) / 64)), dim3(64 * SUMG), 0, s, w->part, np, len, dW);
  LHIP(hipGetLastError());
  return 0;
}

// rows per chunk for `per_row` elements of sizeof(T) each
template <typename T>
long chunk_rows(long n, long per_row) {
  long c = (long)(WS_TARGET / ((size_t)per_row * sizeof(T)));
  if (c < 1024) c = 1024;
  // the stacked GEMMs take S*c rows in a 32-bit int
  if (c > (1l << 26)) c = 1l << 26;
  return c < n ? (c & ~3l) : n;             // several chunks: a multiple of 4 rows, so that every chunk's matrices stay 16-byte aligned
}

// Z(M x Hout) = A(M x Hin) W(Hin x Hout), all row-major (vn_gemm.hip: fp32 and fp64 MFMA products)
template <typename T>
int gemm_fwd(VnLayered* w, long M, int Hin, int Hout, const T* A, const T* W, T* Z, hipStream_t s, char* err, size_t errlen) {
  if constexpr (sizeof(T) == 4) LGEMM(vn_gemm_nn(A, W, Z, M, Hout, Hin, s));
  else LGEMM(vn_dgemm_nn(A, W, Z, M, Hout, Hin, s));
  return 0;
}
// y(M) = beta y + A(M x H) w
template <typename T>
int gemv_rows(VnLayered* w, long M, int H, const T* A, const T* x, T beta, T* y, hipStream_t s, char* err, size_t errlen) {
  if constexpr (sizeof(T) == 4) LGEMM(vn_rowdot(A, x, y, M, H, beta, s));
  else LGEMM(vn_drowdot(A, x, y, M, H, beta, s));
  return 0;
}

}  // namespace

int vn_layered_create(VnLayered** out, const VnNet& net, char* err, size_t errlen) {
  *out = nullptr;
  VnLayered* w = new VnLayered();
  w->net = net;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      w->ncu = prop.multiProcessorCount;
  }
  const char* nk = getenv("VN_LAYERED_NOKEEP");
  w->never_keep = nk && *nk && *nk != '0';
  for (int l = 0; l <= net.L; ++l) {
    w->sumH += net.H[l];
    if (net.H[l] > w->hmax_all) w->hmax_all = net.H[l];
  }
  if (vn_wide_supported(net)) {
    if (int rc = vn_wide_create(&w->wide, net, err, errlen)) {
      vn_layered_destroy(w);
      return rc;
    }
  }
  *out = w;
  return 0;
}

void vn_layered_destroy(VnLayered* w) {
  if (!w) return;
  vn_wide_destroy(w->wide);
  if (w->wt) (void)hipFree(w->wt);
  if (w->ws) (void)hipFree(w->ws);
  if (w->part) (void)hipFree(w->part);
  for (auto& k : w->kept) if (k.buf) (void)hipFree(k.buf);
  delete w;
}

namespace {

// Forward of one chunk.  act[l] (l = 0..L): [S][c][H_l] stacked (a, ad).
int chunk_forward(VnLayered* w, const float* theta, const float* X, const float* G, long c, int S, float** act, hipStream_t s,
                  char* err, size_t errlen) {
  const VnNet& net = w->net;
  hipLaunchKernelGGL(k_pack_train, dim3(blocks(c * net.d_in)), dim3(EB), 0, s, X, G, c, net.d_in, net.dim, S, act[0]);
  LHIP(hipGetLastError());
  for (int l = 1; l <= net.L; ++l) {
    // product and layer epilogue in one kernel (vn_gemm.hip)
    LGEMM(vn_gemm_fwd(act[l - 1], theta + net.woff[l], theta + net.boff[l], act[l], c, S, net.H[l], net.H[l - 1], net.actl[l], s));
  }
  return 0;
}

// carve the chunk buffers out of `base` (activations) and the workspace (adjoints); returns elements used of each
long carve_act(const VnLayered* w, float* base, long c, int S, float** act) {
  const VnNet& net = w->net;
  long used = 0;
  for (int l = 0; l <= net.L; ++l) {
    act[l] = base ? base + used : nullptr;
    used = (used + (long)S * c * net.H[l] + 3) & ~3l;      // every matrix starts on a 16-byte boundary (vn_gemm.hip: b128 loads)
  }
  return used;
}
long carve(VnLayered* w, long c, int S, bool train, float** act, float** adj, bool act_elsewhere = false) {
  float* p = (float*)w->ws;
  long used = act_elsewhere ? 0 : carve_act(w, p, c, S, act);
  if (train) {
    for (int i = 0; i < 2; ++i) {
      adj[i] = p ? p + used : nullptr;
      used = (used + (long)S * c * w->hmax_all + 3) & ~3l;
    }
  }
  return used;
}

// room for the activations of all n rows in slot `slot`, cut into chunks of c rows?  (never more than half of what is free)
bool kept_reserve(VnLayered* w, int slot, long n, long c, int S) {
  if (w->never_keep) return false;
  VnLayered::Kept& k = w->kept[slot];
  // (chunks of c rows, c a multiple of 4 when there is more than one; + the 16-byte rounding of a ragged last chunk's matrices)
  const size_t need = (size_t)((n + c - 1) / c) * (size_t)S * c * w->sumH + 4 * (VN_MAX_LAYERS + 2);
  if (need > k.cap) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    if ((need - k.cap) * sizeof(float) > fr / 2) return false;
    if (k.buf) (void)hipFree(k.buf);
    k.buf = nullptr; k.cap = 0;
    if (hipMalloc((void**)&k.buf, need * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return false; }
    k.cap = need;
  }
  return true;
}

}  // namespace

int vn_layered_forward(VnLayered* w, const float* theta, const VnRows& seg, hipStream_t s, char* err, size_t errlen,
                       int keep_slot) {
  if (keep_slot >= 0) w->kept[keep_slot].valid = false;
  if (seg.n <= 0) return 0;
  if (w->wide) return vn_wide_forward(w->wide, theta, seg, keep_slot, !w->never_keep, s, err, errlen);
  const VnNet& net = w->net;
  const int S = (seg.G && seg.ud) ? 2 : 1;
  // with keep_slot >= 0 the chunking is the reverse pass's, and the activations go to the slot instead of the workspace
  long c = chunk_rows<float>(seg.n, (long)S * w->sumH);
  bool keep = false;
  if (keep_slot >= 0) {
    const long cb = chunk_rows<float>(seg.n, (long)S * w->sumH + 2l * S * w->hmax_all);
    keep = kept_reserve(w, keep_slot, seg.n, cb, S);
    if (keep) c = cb;
  }
  float *act[VN_MAX_LAYERS + 2], *adj[2];
  if (!keep) {
    if (int rc = ensure_ws(w, (size_t)carve(w, c, S, false, act, adj) * sizeof(float), err, errlen)) return rc;
  }
  const int HL = net.H[net.L];
  long k = 0;
  for (long r0 = 0; r0 < seg.n; r0 += c, ++k) {
    const long cn = (seg.n - r0 < c) ? seg.n - r0 : c;
    // the stacked layout depends on the chunk length: carve per chunk (the last one is shorter)
    if (keep) carve_act(w, w->kept[keep_slot].buf + (size_t)k * S * c * w->sumH, cn, S, act);
    else carve(w, cn, S, false, act, adj);
    if (int rc = chunk_forward(w, theta, seg.X + r0 * net.d_in, S == 2 ? seg.G + r0 * net.dim : nullptr, cn, S, act, s, err, errlen))
      return rc;
    LTRACE(s, "fwd chunk %ld rows %ld keep=%d act0=%p", k, cn, (int)keep, (void*)act[0]);
    // u = a_L w_o + b_o,  ud = ad_L w_o
    const unsigned rb = blocks(cn * 16) < 8192u ? blocks(cn * 16) : 8192u;
    hipLaunchKernelGGL(k_rowdot, dim3(rb), dim3(EB), 0, s, act[net.L], theta + net.woff[net.L + 1], theta + net.boff[net.L + 1], cn, HL,
                       seg.u + r0);
    LHIP(hipGetLastError());
    if (S == 2) {
      hipLaunchKernelGGL(k_rowdot, dim3(rb), dim3(EB), 0, s, act[net.L] + cn * HL, theta + net.woff[net.L + 1], (const float*)nullptr,
                         cn, HL, seg.ud + r0);
      LHIP(hipGetLastError());
    }
  }
  if (keep) {
    VnLayered::Kept& kp = w->kept[keep_slot];
    kp.X = seg.X; kp.n = seg.n; kp.c = c; kp.S = S; kp.valid = true;
  }
  return 0;
}

int vn_layered_backward(VnLayered* w, const float* theta, const VnRows& seg, float* grad, hipStream_t s, char* err,
                        size_t errlen, int keep_slot) {
  if (seg.n <= 0) return 0;
  // activations stored by the tile kernels' forward: their reverse kernel; else the GEMMs below recompute per chunk
  if (w->wide && vn_wide_has_kept(w->wide, keep_slot, seg)) return vn_wide_backward(w->wide, theta, seg, grad, keep_slot, s, err, errlen);
  const VnNet& net = w->net;
  const int S = (seg.G && seg.udbar) ? 2 : 1;
  const long per_row = (long)S * w->sumH + 2l * S * w->hmax_all;
  long c = chunk_rows<float>(seg.n, per_row);
  // activations kept by the forward of this gradient evaluation (same rows, same stacking): no recompute
  VnLayered::Kept* kp = keep_slot >= 0 ? &w->kept[keep_slot] : nullptr;
  const bool kept = kp && kp->valid && kp->X == seg.X && kp->n == seg.n && kp->S == S;
  if (kept) c = kp->c;
  if (kp) kp->valid = false;                        // theta moves after this step
  float *act[VN_MAX_LAYERS + 2], *adj[2];
  if (int rc = ensure_ws(w, (size_t)carve(w, c, S, true, act, adj, kept) * sizeof(float), err, errlen)) return rc;
  const int L = net.L, HL = net.H[L];
  long k = 0;
  for (long r0 = 0; r0 < seg.n; r0 += c, ++k) {
    const long cn = (seg.n - r0 < c) ? seg.n - r0 : c;
    carve(w, cn, S, true, act, adj, kept);
    if (kept) carve_act(w, kp->buf + (size_t)k * S * c * w->sumH, cn, S, act);
    else if (int rc = chunk_forward(w, theta, seg.X + r0 * net.d_in, S == 2 ? seg.G + r0 * net.dim : nullptr, cn, S, act, s, err,
                                    errlen))
      return rc;
    LTRACE(s, "bwd chunk %ld rows %ld kept=%d act0=%p actL=%p adj=%p,%p ws=%p/%zu", k, cn, (int)kept, (void*)act[0], (void*)act[L], (void*)adj[0], (void*)adj[1], w->ws, w->ws_bytes);
    const float* ubar = seg.ubar + r0;
    const float* udbar = S == 2 ? seg.udbar + r0 : nullptr;
    // output layer: dw_o += a_L^T ubar (+ ad_L^T udbar), db_o += sum ubar
    if (int rc = colsum_add(w, act[L], ubar, cn, HL, grad + net.woff[L + 1], s, err, errlen)) return rc;
    if (S == 2)
      if (int rc = colsum_add(w, act[L] + cn * HL, udbar, cn, HL, grad + net.woff[L + 1], s, err, errlen)) return rc;
    if (int rc = colsum_add(w, ubar, nullptr, cn, 1, grad + net.boff[L + 1], s, err, errlen)) return rc;
    LTRACE(s, "bwd output-layer sums done");
    float* cur = adj[0];
    float* nxt = adj[1];
    hipLaunchKernelGGL(k_seed_act_bwd, dim3(blocks(cn * HL)), dim3(EB), 0, s, ubar, udbar, theta + net.woff[L + 1], act[L], cn, HL, S,
                       net.actl[L], cur);
    LHIP(hipGetLastError());
    for (int l = L; l >= 1; --l) {
      const int Hin = net.H[l - 1], Hout = net.H[l];
      const long M = (long)S * cn;
      if (l < L) {
        hipLaunchKernelGGL(k_act_bwd, dim3(blocks(cn * Hout)), dim3(EB), 0, s, cur, act[l], cn, Hout, S, net.actl[l]);
        LHIP(hipGetLastError());
      }
      LTRACE(s, "bwd layer %d act_bwd done", l);
      // db_l += sum over the value-stream rows of zbar;  dW_l += [a; ad]^T [zbar; zdbar]
      if (int rc = colsum_add(w, cur, nullptr, cn, Hout, grad + net.boff[l], s, err, errlen)) return rc;
      if (int rc = wgrad_add(w, act[l - 1], cur, M, Hin, Hout, grad + net.woff[l], s, err, errlen)) return rc;
      LTRACE(s, "bwd layer %d colsum + wgrad done", l);
      if (l > 1) {
        // [abar; adbar]_{l-1} (M x Hin) = Zbar (M x Hout) W_l^T  ==  column-major (Hin x M) = W'(Hout x Hin)^T Zbar'(Hout x M)
        // dA = Zb W^T as a plain product with the transposed weights (a few hundred KB, rewritten per call)
        if (int rc = ensure_wt(w, (size_t)Hin * Hout, err, errlen)) return rc;
        LGEMM(vn_transpose(theta + net.woff[l], w->wt, Hin, Hout, s));
        LGEMM(vn_gemm_nn(cur, w->wt, nxt, M, Hin, Hout, s));
        // (the reverse epilogue of layer l-1 fused into this product was built and measured: its loads of the stored
        // (a | ad) in accumulator order cost what k_act_bwd costs -- 512,512: 2 128 us against 1 500 + 425 -- so it stays
        // a separate, HBM-bound elementwise kernel)
        float* t = cur; cur = nxt; nxt = t;
      }
    }
  }
  return 0;
}

namespace {

template <typename T>
int pointwise_streams(VnLayered* w, const T* theta, const T* X, long n, int S, int nd1, int dim, const T* diff, const T* vel,
                      const T* src, const T* ddx, int td, T* u, T* res, hipStream_t s, char* err, size_t errlen) {
  if (n <= 0) return 0;
  const VnNet& net = w->net;
  // two ping-pong buffers of S x c x hmax and the S x c output vector
  const long per_row = 2l * S * w->hmax_all + S;
  const long c = chunk_rows<T>(n, per_row);
  if (int rc = ensure_ws(w, (size_t)per_row * c * sizeof(T), err, errlen)) return rc;
  T* buf0 = (T*)w->ws;
  T* buf1 = buf0 + (long)S * c * w->hmax_all;
  T* y = buf1 + (long)S * c * w->hmax_all;
  const T one = T(1), zero = T(0);
  const int HL = net.H[net.L];
  for (long r0 = 0; r0 < n; r0 += c) {
    const long cn = (n - r0 < c) ? n - r0 : c;
    T* cur = buf0;
    T* nxt = buf1;
    if (res) {
      hipLaunchKernelGGL(k_pack_res<T>, dim3(blocks(cn * net.d_in)), dim3(EB), 0, s, X + r0 * net.d_in, cn, net.d_in, nd1, S, cur);
      LHIP(hipGetLastError());
    }
    for (int l = 1; l <= net.L; ++l) {
      const T* in = (l == 1 && !res) ? X + r0 * net.d_in : cur;
      if (int rc = gemm_fwd<T>(w, (long)S * cn, net.H[l - 1], net.H[l], in, theta + net.woff[l], nxt, s, err, errlen)) return rc;
      hipLaunchKernelGGL(k_act_res<T>, dim3(blocks(cn * net.H[l])), dim3(EB), 0, s, nxt, theta + net.boff[l], cn, net.H[l],
                         res ? nd1 : 0, res ? dim : 0, net.actl[l]);
      LHIP(hipGetLastError());
      T* t = cur; cur = nxt; nxt = t;
    }
    if (!res) {
      hipLaunchKernelGGL(k_fill_from<T>, dim3(blocks(cn)), dim3(EB), 0, s, u + r0, cn, theta + net.boff[net.L + 1]);
      LHIP(hipGetLastError());
      if (int rc = gemv_rows<T>(w, cn, HL, cur, theta + net.woff[net.L + 1], one, u + r0, s, err, errlen)) return rc;
    } else {
      if (int rc = gemv_rows<T>(w, (long)S * cn, HL, cur, theta + net.woff[net.L + 1], zero, y, s, err, errlen)) return rc;
      hipLaunchKernelGGL(k_res_combine<T>, dim3(blocks(cn)), dim3(EB), 0, s, y, theta + net.boff[net.L + 1], diff + r0,
                         vel + r0 * dim, src ? src + r0 : nullptr, ddx ? ddx + r0 * dim : nullptr, td, cn, dim, nd1,
                         u ? u + r0 : nullptr, res + r0);
      LHIP(hipGetLastError());
    }
  }
  return 0;
}

}  // namespace

int vn_layered_forward_f64(VnLayered* w, const double* theta, const double* X, long n, double* u, hipStream_t s, char* err,
                           size_t errlen) {
  return pointwise_streams<double>(w, theta, X, n, 1, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, u, nullptr, s, err, errlen);
}

int vn_layered_residual_f32(VnLayered* w, const float* theta, const float* X, const float* diff, const float* vel,
                            const float* src, const float* ddx, int td, long n, float* u, float* res, hipStream_t s, char* err,
                            size_t errlen) {
  const int dim = w->net.dim, nd1 = dim + (td ? 1 : 0);
  return pointwise_streams<float>(w, theta, X, n, 1 + nd1 + dim, nd1, dim, diff, vel, src, ddx, td, u, res, s, err, errlen);
}

int vn_layered_residual_f64(VnLayered* w, const double* theta, const double* X, const double* diff, const double* vel,
                            const double* src, const double* ddx, int td, long n, double* u, double* res, hipStream_t s,
                            char* err, size_t errlen) {
  const int dim = w->net.dim, nd1 = dim + (td ? 1 : 0);
  return pointwise_streams<double>(w, theta, X, n, 1 + nd1 + dim, nd1, dim, diff, vel, src, ddx, td, u, res, s, err, errlen);
}
