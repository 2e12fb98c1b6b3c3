// Host-side declarations of vn_dedup.hip (kept out of vn_internal.h, which every kernel's source hash covers).
#pragma once
#include "vn_internal.h"

// ---- de-duplicated weak-form assembly (vn_dedup.hip) -----------------------------------------
struct VnDedupArgs {
  const float* upack;                        // [U, 4]: (u, du/dx_0, du/dx_1, du/dx_2) at the unique points (vn_pgrad16's out_pack)
  const int* uid;                            // [nT] row -> unique point
  const int* rowptr; const int* rowidx;      // CSR unique point -> rows
  const float* gcoef; const float* source;   // [nT, dim], [nT] or nullptr
  const float* gcoef_csr;                    // [nT, dim]: gcoef[rowidx[e]] (what the gather kernel reads, contiguously)
  const float* feN; const float* fedNt; const float* feW;
  const float* detJv; float detJ;
  long n_k, U; int q, dim, time_dependent;
  float w2;
  int gper;                                  // gcoef repeats with period q along the rows: read rows [0, q) as the table
  float* stf;                                // [n_k] seed 2 w2 detJ R_k of every test function (nullptr: loss only)
  float* lossVec;                            // [n_k] or nullptr
  float* part;                               // [grid*3] block partials (var, 0, 0)
  float* seed_u; float* seed_g;              // [U], [U, dim] gathered seeds
};
constexpr int VN_DEDUP_TFB = 32;            // test functions per workgroup of the seed kernel = per loss partial (grid = ceil(n_k / 32))
hipError_t vn_dedup_seed_launch(const VnDedupArgs& a, int grid, hipStream_t s);
hipError_t vn_dedup_gather_launch(const VnDedupArgs& a, hipStream_t s);
// *err_dev += number of inconsistencies of the map (see vn_dedup_check_kernel); err_dev must hold 0 on entry
hipError_t vn_dedup_check_launch(const int* uid, const int* rowptr, const int* rowidx, long nT, long U, int* err_dev, hipStream_t s);
// *err_dev += number of rows whose gcoef differs bitwise from the row of test function 0 at the same quadrature point
hipError_t vn_dedup_periodic_launch(const float* gcoef, long nT, int q, int dim, int* err_dev, hipStream_t s);
hipError_t vn_dedup_permute_launch(const float* gcoef, const int* rowidx, float* gcoef_csr, long nT, int dim, hipStream_t s);

