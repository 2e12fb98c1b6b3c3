// De-duplicated weak-form assembly.  On the reference's uniform grids neighbouring hat functions
// share elements, so every quadrature point is evaluated 2^feDim times (once per test function
// whose support contains it; VarNet.py:576-588).  The network value and input gradient depend
// only on the point, so they are computed ONCE per unique point; this file holds the two small
// HBM-bound kernels that connect unique points and (test function, quadrature point) rows:
//   vn_dedup_seed_kernel   rows gather (u, grad u) of their point, form the weak-form integrand
//                          sum_d u_{x_d} gcoef_d - u dNt - s N (TFModel.py:653-657), R_k, lossVec,
//                          loss partials and the per-row seed 2 w2 detJ R_k w_p;
//   vn_dedup_gather_kernel each unique point sums the seeds of its rows, in CSR order (fixed ->
//                          bitwise reproducible), into d loss/d u and d loss/d u_{x_d}.
#include "vn_internal.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// One block = VN_DEDUP_TFB test functions (the loss-partial layout of the caller: one partial per block).  Their rows are
// walked in chunks of whole test functions, ONE ROW PER THREAD: consecutive threads read consecutive rows of uid / gcoef / source
// and write consecutive rows of srow (round 4: the first version gave every thread one test function, i.e. a stride of integ_num
// rows between neighbouring lanes -- every 4-byte access its own cache line).  R_k of a test function is summed in a fixed
// order (16-row groups, then the groups) whatever the chunking, so results are bitwise reproducible.
__global__ __launch_bounds__(256) void vn_dedup_seed_kernel(VnDedupArgs a) {
  __shared__ float red[4];
  __shared__ float sval[256];
  __shared__ float sgrp[16];
  __shared__ float sR[256];
  const int q = a.q, dim = a.dim, tid = threadIdx.x;
  const long k0 = (long)blockIdx.x * VN_DEDUP_TFB;
  const long k1 = (k0 + VN_DEDUP_TFB < a.n_k) ? k0 + VN_DEDUP_TFB : a.n_k;
  float lv = 0.f;
  // q <= 128 here: the formulation needs the 8-wave fused kernel, whose tile holds whole test functions (vn_set_dedup)
  const int tpc = 256 / q;                           // test functions per chunk
  const int tf = tid / q, p = tid - tf * q;
  const bool act = tf < tpc;
  for (long kc = k0; kc < k1; kc += tpc) {
    const long k = kc + tf;
    const bool live = act && k < k1;
    const long r = k * q + p;
    float t = 0.f;
    if (live) {
      const long j = a.uid[r];
      for (int d = 0; d < dim; ++d) t += a.ug[j * dim + d] * a.gcoef[r * dim + d];   // TFModel.py:653-654
      if (a.time_dependent) t -= a.uv[j] * a.fedNt[p];                              // :655
      if (a.source) t -= a.source[r] * a.feN[p];                                    // :657
      if (a.feW) t *= a.feW[p];                                                     // :660
    }
    sval[tid] = t;
    __syncthreads();
    // level 1: the first row of every 16-row group of a test function sums its group; level 2: row 0 sums the groups
    if ((q & 15) == 0 && live && (p & 15) == 0) {
      const int n = (q - p < 16) ? q - p : 16;
      float g = 0.f;
      for (int i = 0; i < n; ++i) g += sval[tid + i];
      sgrp[tid >> 4] = g;                            // tid = tf*q + p, q a multiple of 16: groups of different test functions
    }                                                // never share tid >> 4; any other q: level 2 re-reads sval (below)
    __syncthreads();
    if (live && p == 0) {
      float R = 0.f;
      if ((q & 15) == 0) {
        for (int i = 0; i < q / 16; ++i) R += sgrp[(tid >> 4) + i];
      } else {
        for (int i = 0; i < q; ++i) R += sval[tid + i];
      }
      const float dj = a.detJv ? a.detJv[k] : a.detJ;
      const float l = dj * R * R;
      lv += l;
      if (a.lossVec) a.lossVec[k] = l;
      sR[tf] = 2.f * a.w2 * dj * R;
    }
    __syncthreads();
    if (live && a.srow) {
      const float s0 = sR[tf];
      a.srow[r] = a.feW ? s0 * a.feW[p] : s0;
    }
    // (the next chunk's stores to sval / sgrp / sR come behind its own first barrier for sgrp and sR; sval is rewritten at
    // once, but every read of this chunk's sval happened before the barrier above)
  }
  const float s = block_sum256(lv, red);
  if (threadIdx.x == 0) {
    a.part[blockIdx.x * 3 + 0] = s;
    a.part[blockIdx.x * 3 + 1] = 0.f;
    a.part[blockIdx.x * 3 + 2] = 0.f;
  }
}

// Seed gather: d loss / d u and d loss / d u_{x_d} of a unique point = sum over its rows, in CSR order (fixed -> bitwise
// reproducible).  A block owns VN_GATHER_PB consecutive unique points, i.e. one contiguous range of CSR entries.  Phase 1 walks
// that range ONE ENTRY PER THREAD -- row index and CSR-ordered gcoef are contiguous streams, the row's seed is the one true gather
// (4 bytes out of a 25-MB array that lives in the last-level cache) -- and leaves the per-entry products in LDS; all loads of a
// thread are independent, none waits for a neighbour's.  Phase 2: every point adds up its own entries from LDS in order.
// (Round 5, first form: one thread per point walking its rows -- three dependent load latencies per four rows, 55 us; and before
// that a gather of gcoef by row, which fetched 2.6 x the bytes it used.)
constexpr int VN_GATHER_PB = 256;           // unique points per block
constexpr int VN_GATHER_CH = 2304;          // CSR entries per LDS chunk (256 points x 8 rows + slack: normally one chunk)
__global__ __launch_bounds__(256) void vn_dedup_gather_kernel(VnDedupArgs a) {
  __shared__ float sp[VN_GATHER_CH][4];     // (-dNt s, g0 s, g1 s, g2 s) per entry
  __shared__ int sptr[VN_GATHER_PB + 1];
  const int tid = threadIdx.x, dim = a.dim, q = a.q;
  const long j0 = (long)blockIdx.x * VN_GATHER_PB;
  const int nj = (int)((a.U - j0 < VN_GATHER_PB) ? a.U - j0 : VN_GATHER_PB);
  for (int i = tid; i <= nj; i += 256) sptr[i] = a.rowptr[j0 + i];
  __syncthreads();
  const int e0 = sptr[0], e1 = sptr[nj];
  const int s0 = (tid < nj) ? sptr[tid] : 0, s1 = (tid < nj) ? sptr[tid + 1] : 0;
  float su = 0.f, sg[3] = {0.f, 0.f, 0.f};
  for (int base = e0; base < e1; base += VN_GATHER_CH) {
    const int n = (e1 - base < VN_GATHER_CH) ? e1 - base : VN_GATHER_CH;
    for (int i = tid; i < n; i += 256) {
      const long e = (long)base + i;
      const long r = a.rowidx[e];
      const float s = a.srow[r];
      sp[i][0] = a.time_dependent ? -(a.fedNt[(int)(r % q)] * s) : 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) sp[i][1 + d] = (d < dim) ? a.gcoef_csr[e * dim + d] * s : 0.f;
    }
    __syncthreads();
    const int lo = (s0 > base) ? s0 : base, hi = (s1 < base + n) ? s1 : base + n;
    for (int e = lo; e < hi; ++e) {
      su += sp[e - base][0];
#pragma unroll
      for (int d = 0; d < 3; ++d) sg[d] += sp[e - base][1 + d];
    }
    __syncthreads();
  }
  if (tid < nj) {
    a.seed_u[j0 + tid] = su;
    for (int d = 0; d < dim; ++d) a.seed_g[(j0 + tid) * dim + d] = sg[d];
  }
}

// Registration-time check of a de-duplication map (vn_set_dedup): every later kernel indexes device memory with these
// arrays, so an inconsistent map must come back as an error code, not as a GPU fault.  err[0] = number of violations:
// uid[r] in [0, U); rowptr[0] = 0, non-decreasing, rowptr[U] = nT; rowidx[e] in [0, nT) and uid[rowidx[e]] = the point
// whose segment holds e.
__global__ __launch_bounds__(256) void vn_dedup_check_kernel(const int* uid, const int* rowptr, const int* rowidx, long nT, long U,
                                                             int* err) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  int bad = 0;
  if (i < nT) bad += (uid[i] < 0 || uid[i] >= U);
  if (i == 0) bad += (rowptr[0] != 0) + (rowptr[U] != nT);
  if (i < U) {
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    if (e0 > e1 || e0 < 0 || e1 > nT) {
      bad += 1;
    } else {
      for (int e = e0; e < e1; ++e) {
        const int r = rowidx[e];
        if (r < 0 || r >= nT) bad += 1;
        else if (uid[r] != i) bad += 1;
      }
    }
  }
  if (bad) atomicAdd(err, bad);
}

// gcoef_csr[e] = gcoef[rowidx[e]]: once per vn_set_dedup
__global__ __launch_bounds__(256) void vn_dedup_permute_kernel(const float* gcoef, const int* rowidx, float* out, long nT, int dim) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nT) return;
  const long r = rowidx[e];
  for (int d = 0; d < dim; ++d) out[e * dim + d] = gcoef[r * dim + d];
}

}  // namespace

hipError_t vn_dedup_seed_launch(const VnDedupArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL(vn_dedup_seed_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t vn_dedup_check_launch(const int* uid, const int* rowptr, const int* rowidx, long nT, long U, int* err_dev, hipStream_t s) {
  const long n = nT > U ? nT : U;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(vn_dedup_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, uid, rowptr, rowidx, nT, U, err_dev);
  return hipGetLastError();
}

hipError_t vn_dedup_permute_launch(const float* gcoef, const int* rowidx, float* gcoef_csr, long nT, int dim, hipStream_t s) {
  if (nT <= 0) return hipSuccess;
  hipLaunchKernelGGL(vn_dedup_permute_kernel, dim3((unsigned)((nT + 255) / 256)), dim3(256), 0, s, gcoef, rowidx, gcoef_csr, nT, dim);
  return hipGetLastError();
}

hipError_t vn_dedup_gather_launch(const VnDedupArgs& a, hipStream_t s) {
  const int grid = (int)((a.U + VN_GATHER_PB - 1) / VN_GATHER_PB);
  hipLaunchKernelGGL(vn_dedup_gather_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
