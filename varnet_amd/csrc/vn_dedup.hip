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

// One thread per unique point sums the seeds of its rows in CSR order (fixed -> bitwise reproducible).  The chain
// rowidx[e] -> srow[r], gcoef[r] is two dependent gathers per row; the loads of four rows are issued together before
// the sums (same order of additions as a one-row-at-a-time walk), so a thread has eight gathers in flight, not two.
__global__ __launch_bounds__(256) void vn_dedup_gather_kernel(VnDedupArgs a) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.U) return;
  const int dim = a.dim, q = a.q;
  float su = 0.f, sg[3] = {0.f, 0.f, 0.f};
  const int e1 = a.rowptr[j + 1];
  for (int e = a.rowptr[j]; e < e1; e += 4) {
    long r[4];
    float s[4], g[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (e + i < e1) ? a.rowidx[e + i] : -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] = (r[i] >= 0) ? a.srow[r[i]] : 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) g[i][d] = (r[i] >= 0 && d < dim) ? a.gcoef[r[i] * dim + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (r[i] < 0) continue;
      if (a.time_dependent) su -= a.fedNt[(int)(r[i] % q)] * s[i];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (d < dim) sg[d] += g[i][d] * s[i];
    }
  }
  a.seed_u[j] = su;
  for (int d = 0; d < dim; ++d) a.seed_g[j * dim + d] = sg[d];
}

}  // namespace

hipError_t vn_dedup_seed_launch(const VnDedupArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL(vn_dedup_seed_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t vn_dedup_gather_launch(const VnDedupArgs& a, hipStream_t s) {
  const int grid = (int)((a.U + 255) / 256);
  hipLaunchKernelGGL(vn_dedup_gather_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
