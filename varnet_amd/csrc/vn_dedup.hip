// De-duplicated weak-form assembly.  On the reference's uniform grids neighbouring hat functions
// share elements, so every quadrature point is evaluated 2^feDim times (once per test function
// whose support contains it; VarNet.py:576-588).  The network value and input gradient depend
// only on the point, so they are computed ONCE per unique point; this file holds the two small
// HBM-bound kernels that connect unique points and (test function, quadrature point) rows:
//   vn_dedup_seed_kernel   rows gather (u, grad u) of their point, form the weak-form integrand
//                          sum_d u_{x_d} gcoef_d - u dNt - s N (TFModel.py:653-657), R_k, lossVec,
//                          loss partials and the seed 2 w2 detJ R_k of every TEST FUNCTION (the seed of a row
//                          is that times the quadrature weight of its point: no per-row array is written);
//   vn_dedup_gather_kernel each unique point sums the seeds of its rows, in CSR order (fixed ->
//                          bitwise reproducible), into d loss/d u and d loss/d u_{x_d}.
// With constant coefficients gcoef = kappa dN/dx + v N repeats with period integ_num along the rows (the reference tiles it to
// nT rows, VarNet.py:837): vn_set_dedup detects that (bitwise) and both kernels then read the integ_num-entry table that the
// first test function's rows are, instead of 8 bytes per row each.
#include "vn_internal.h"
#include "vn_dedup.h"

typedef float f32x4d __attribute__((ext_vector_type(4)));

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
// (round 4: the first version gave every thread one test function, i.e. a stride of integ_num rows between neighbouring lanes --
// every 4-byte access its own cache line).  R_k of a test function is summed in a fixed
// order (16-row groups, then the groups) whatever the chunking, so results are bitwise reproducible.
__global__ __launch_bounds__(256) void vn_dedup_seed_kernel(VnDedupArgs a) {
  // The kernel is bound by dependent-load latency (row -> point index -> point data -> LDS -> barriers), not by bytes: NCH
  // chunks of 256 rows are in flight per thread -- all their index loads, then all their gathers, then ONE barrier sequence
  // for the NCH reductions.  Order of additions inside a test function unchanged (16-row groups, then the groups).
  constexpr int NCH = 4;
  __shared__ float red[4];
  __shared__ float sval[NCH][256];
  __shared__ float sgrp[NCH][16];
  const int q = a.q, dim = a.dim, tid = threadIdx.x;
  const long k0 = (long)blockIdx.x * VN_DEDUP_TFB;
  const long k1 = (k0 + VN_DEDUP_TFB < a.n_k) ? k0 + VN_DEDUP_TFB : a.n_k;
  float lv = 0.f;
  // q <= 256 here (vn_set_dedup): a chunk holds whole test functions
  const int tpc = 256 / q;                           // test functions per chunk
  const int tf = tid / q, p = tid - tf * q;
  const bool act = tf < tpc;
  const float dnt = a.time_dependent ? a.fedNt[p < q ? p : 0] : 0.f, fN = a.feN[p < q ? p : 0], fw = a.feW ? a.feW[p < q ? p : 0] : 1.f;
  for (long kc = k0; kc < k1; kc += (long)tpc * NCH) {
    long j[NCH];
    bool live[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const long k = kc + (long)c * tpc + tf;
      live[c] = act && k < k1;
      j[c] = live[c] ? a.uid[k * q + p] : 0;
    }
    float t[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      t[c] = 0.f;
      if (live[c]) {
        const long r = (kc + (long)c * tpc + tf) * q + p;
        const long gr = a.gper ? p : r;                                               // periodic gcoef: the table = the rows of test function 0
        const f32x4d pd = *reinterpret_cast<const f32x4d*>(a.upack + j[c] * 4);          // one 16-byte gather per row
        for (int d = 0; d < dim; ++d) t[c] += pd[1 + d] * a.gcoef[gr * dim + d];         // TFModel.py:653-654
        if (a.time_dependent) t[c] -= pd[0] * dnt;                                     // :655
        if (a.source) t[c] -= a.source[r] * fN;                                        // :657
        if (a.feW) t[c] *= fw;                                                         // :660
      }
      sval[c][tid] = t[c];
    }
    __syncthreads();
    // level 1: the first row of every 16-row group of a test function sums its group; level 2: row 0 sums the groups
    if ((q & 15) == 0 && (p & 15) == 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (!live[c]) continue;
        float g = 0.f;
        for (int i = 0; i < 16; ++i) g += sval[c][tid + i];
        sgrp[c][tid >> 4] = g;                       // tid = tf*q + p, q a multiple of 16: groups of different test functions
      }                                              // never share tid >> 4; any other q: level 2 re-reads sval (below)
    }
    __syncthreads();
    if (p == 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (!live[c]) continue;
        const long k = kc + (long)c * tpc + tf;
        float R = 0.f;
        if ((q & 15) == 0) {
          for (int i = 0; i < q / 16; ++i) R += sgrp[c][(tid >> 4) + i];
        } else {
          for (int i = 0; i < q; ++i) R += sval[c][tid + i];
        }
        const float dj = a.detJv ? a.detJv[k] : a.detJ;
        const float l = dj * R * R;
        lv += l;
        if (a.lossVec) a.lossVec[k] = l;
        if (a.stf) a.stf[k] = 2.f * a.w2 * dj * R;   // seed of the test function; a row's seed is this x its quadrature weight
      }
    }
    __syncthreads();
    // (the next iteration rewrites sval / sgrp behind this barrier: every read above has happened)
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
// that range ONE ENTRY PER THREAD: the row index is a contiguous stream; the row's seed is the seed of its TEST FUNCTION
// (row / integ_num: an n_k-entry array that stays in L2) times its quadrature weight; gcoef comes from the integ_num-entry table
// when it is periodic, else from the CSR-ordered copy (contiguous).  The per-entry products go to LDS; all loads of a thread
// are independent, none waits for a neighbour's.  Phase 2: every point adds up its own entries from LDS in order.
// (Round 5, earlier forms: one thread per point walking its rows -- three dependent load latencies per four rows; a gather of
// gcoef by row, which fetched 2.6 x the bytes it used; per-row seeds written by the seed kernel and gathered here: 8 bytes of
// HBM traffic per row for what n_k floats hold.)
constexpr int VN_GATHER_PB = 256;           // unique points per block
constexpr int VN_GATHER_CH = 2304;          // CSR entries per LDS chunk (256 points x 8 rows + slack: normally one chunk)
__global__ __launch_bounds__(256) void vn_dedup_gather_kernel(VnDedupArgs a) {
  __shared__ float sp[VN_GATHER_CH][4];     // (-dNt s, g0 s, g1 s, g2 s) per entry
  __shared__ int sptr[VN_GATHER_PB + 1];
  const int tid = threadIdx.x, dim = a.dim, q = a.q;
  const bool qpow2 = (q & (q - 1)) == 0;
  const int qshift = __ffs(q) - 1;
  const long j0 = (long)blockIdx.x * VN_GATHER_PB;
  const int nj = (int)((a.U - j0 < VN_GATHER_PB) ? a.U - j0 : VN_GATHER_PB);
  for (int i = tid; i <= nj; i += 256) sptr[i] = a.rowptr[j0 + i];
  __syncthreads();
  const int e0 = sptr[0], e1 = sptr[nj];
  const int s0 = (tid < nj) ? sptr[tid] : 0, s1 = (tid < nj) ? sptr[tid + 1] : 0;
  float su = 0.f, sg[3] = {0.f, 0.f, 0.f};
  for (int base = e0; base < e1; base += VN_GATHER_CH) {
    const int n = (e1 - base < VN_GATHER_CH) ? e1 - base : VN_GATHER_CH;
    // four entries per thread in flight: all row indices, then all dependent gathers, then the LDS stores
    for (int i0 = tid; i0 < n; i0 += 4 * 256) {
      int r[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = (i0 + 256 * c < n) ? a.rowidx[(long)base + i0 + 256 * c] : -1;
      float sv[4], g[4][3];
      int pp[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // row -> (test function, quadrature point): a shift when integ_num is a power of two (16, 64: two-point Gauss), else one
        // unsigned division (a signed one by a run-time divisor is ~30 vector instructions per entry)
        const unsigned ru = r[c] >= 0 ? (unsigned)r[c] : 0u;
        const int k = (int)(qpow2 ? ru >> qshift : ru / (unsigned)q);
        pp[c] = (int)(ru - (unsigned)k * (unsigned)q);
        sv[c] = r[c] >= 0 ? a.stf[k] : 0.f;
        const float* gp = a.gper ? a.gcoef + (long)pp[c] * dim : a.gcoef_csr + ((long)base + i0 + 256 * c) * dim;
#pragma unroll
        for (int d = 0; d < 3; ++d) g[c][d] = (r[c] >= 0 && d < dim) ? gp[d] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (r[c] < 0) continue;
        const int i = i0 + 256 * c;
        const float s = a.feW ? sv[c] * a.feW[pp[c]] : sv[c];
        sp[i][0] = a.time_dependent ? -(a.fedNt[pp[c]] * s) : 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) sp[i][1 + d] = g[c][d] * s;
      }
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

// *err += number of rows whose gcoef differs (bitwise) from the row of test function 0 at the same quadrature point
__global__ __launch_bounds__(256) void vn_dedup_periodic_kernel(const unsigned* gcoef, long nT, int q, int dim, int* err) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nT) return;
  const long p = r % q;
  int bad = 0;
  for (int d = 0; d < dim; ++d) bad += gcoef[r * dim + d] != gcoef[p * dim + d];
  if (bad) atomicAdd(err, 1);
}

// Registration-time check of a de-duplication map (vn_set_dedup): every later kernel indexes device memory with these
// arrays, so an inconsistent map must come back as an error code, not as a GPU fault.  err[0] = number of violations:
// uid[r] in [0, U); rowptr[0] = 0, non-decreasing, rowptr[U] = nT; rowidx[e] in [0, nT), uid[rowidx[e]] = the point
// whose segment holds e, and the rows of a point in strictly increasing order.  Together: the nT entries of rowidx are
// distinct (within a segment by the order, across segments by their uid) and in range, i.e. a permutation of the rows --
// a row listed twice (and another one never) would pass every range test and give a silently wrong gradient.
// The arrays' LENGTHS are the caller's contract (uid, rowidx: n_k * integ_num entries, rowptr: U + 1): the ABI carries
// pointers only, the Python binding asserts them.
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
      int prev = -1;
      for (int e = e0; e < e1; ++e) {
        const int r = rowidx[e];
        if (r < 0 || r >= nT) bad += 1;
        else if (uid[r] != i || r <= prev) bad += 1;
        prev = r;
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
  if (grid <= 0) return hipSuccess;              // no test functions: nothing to assemble (a grid of 0 is a launch error)
  hipLaunchKernelGGL(vn_dedup_seed_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t vn_dedup_periodic_launch(const float* gcoef, long nT, int q, int dim, int* err_dev, hipStream_t s) {
  if (nT <= 0) return hipSuccess;
  hipLaunchKernelGGL(vn_dedup_periodic_kernel, dim3((unsigned)((nT + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const unsigned*>(gcoef), nT, q, dim, err_dev);
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
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(vn_dedup_gather_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
