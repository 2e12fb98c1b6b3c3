// Internal declarations shared by the kernel translation units and the C-ABI host layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/varnet_hip.h"

// Network description passed by value to every kernel.  Layer index l = 1..L are the hidden
// (sigmoid) layers, l = L+1 is the linear output layer; H[0] = d_in, H[L+1] = 1.  The arrays hold the ABI maximum;
// the kernels themselves cover VN_KMAX_LAYERS x VN_KMAX_WIDTH (vn_net_in_kernel_range), the rest is vn_layered.hip.
struct VnNet {
  int d_in, dim, L, P, hmax;
  int act;                       // VN_ACT_SIGMOID | VN_ACT_TANH, uniform over the hidden layers; VN_ACT_PER_LAYER: see actl
  int actl[VN_MAX_LAYERS + 2];   // activation of hidden layer l = 1..L (what vn_layered.hip reads)
  int H[VN_MAX_LAYERS + 2];
  int woff[VN_MAX_LAYERS + 2];   // offset of W_l (row-major [H[l-1], H[l]]) in the flat vector
  int boff[VN_MAX_LAYERS + 2];   // offset of b_l
};

// One contiguous set of rows (points) with optional tangent direction and backward seeds.
struct VnRows {
  const float* X;      // [n, d_in]
  const float* G;      // [n, dim] tangent direction (gcoef) or nullptr (zero tangent)
  const float* ubar;   // [n] d loss / d u      (backward only)
  const float* udbar;  // [n] d loss / d udot   (backward only; nullptr = 0)
  float* u;            // [n] out (forward only)
  float* ud;           // [n] out (forward only; may be nullptr)
  long n;
};

// Weak-form epilogue arguments (TFModel.py:643-668 restated on device).
struct VnSeedArgs {
  // interior
  const float* u; const float* ud;          // [nT]
  const float* source;                      // [nT] or nullptr
  const float* feN; const float* fedNt; const float* feW;   // [integ_num] tables (feW may be nullptr)
  const float* Nrow; const float* dNtrow;   // [nT] per-row overrides or nullptr
  const float* detJv; float detJ;           // per-test-function [n_k] or scalar
  long n_k; int integ_num; int time_dependent;
  float* ubar; float* udbar;                // [nT] out (may be nullptr: loss only)
  float* lossVec;                           // [n_k] out or nullptr
  // boundary / initial
  const float* ub; const float* label;      // [nB]
  long nB, bDof; float biDimVal;
  float* ubar_b;                            // [nB] out (may be nullptr)
  float w0, w1, w2;
  float* part;                              // [gridDim.x * 3] block partials (var, bc, ic)
};

inline bool vn_net_in_kernel_range(const VnNet& net) {
  return net.L <= VN_KMAX_LAYERS && net.hmax <= VN_KMAX_WIDTH && net.d_in <= VN_KMAX_DIN && net.act != VN_ACT_PER_LAYER;
}

// ---- generic (any width <= 64, any integ_num) kernels: vn_generic.hip -------------------
size_t vn_generic_fwd_lds_bytes(const VnNet& net);
size_t vn_generic_bwd_lds_bytes(const VnNet& net);
hipError_t vn_generic_forward(const VnNet& net, const float* theta, VnRows seg0, VnRows seg1,
                              int grid, hipStream_t s);
hipError_t vn_generic_backward(const VnNet& net, const float* theta, VnRows seg0, VnRows seg1,
                               float* partial, int grid, hipStream_t s);
hipError_t vn_seed_launch(const VnSeedArgs& a, int grid, hipStream_t s);
// Optional optimizer step fused into the reduction (one launch less per training step when no collective
// sits between gradient and update): kind -1 = none, VN_OPT_ADAM (lr = lr_t), VN_OPT_RMSPROP.
struct VnOptArgs {
  int kind = -1;
  float* theta = nullptr; float* m = nullptr; float* v = nullptr;
  float lr = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f;
  float* loss_acc = nullptr;          // += loss (device scalar) or nullptr
};
// grad[p] = sum_g partial[g*P+p]; tail[0..3] = loss, BC, IC, var from the seed partials.
hipError_t vn_reduce_launch(const float* partial, int nparts, int P, const float* losspart,
                            int nlossparts, long bDof, long nB, float w0, float w1, float w2,
                            float* gradbuf, hipStream_t s, VnOptArgs opt = VnOptArgs());
hipError_t vn_adam_launch(float* theta, float* m, float* v, const float* grad, int P, float lr_t,
                          float b1, float b2, float eps, float* loss_acc, hipStream_t s);
hipError_t vn_rmsprop_launch(float* theta, float* mom, float* ms, const float* grad, int P, float lr, float rho,
                              float momentum, float eps, float* loss_acc, hipStream_t s);

// ---- fused forward+epilogue+backward kernel (vn_fused.hip) --------------------------------
struct VnFusedArgs {
  VnNet net;
  const float* theta;
  const float* X; const float* G; const float* src;   // interior rows
  long nT, n_k; int integ_num;
  const float* feN; const float* fedNt; const float* feW;
  const float* Nrow; const float* dNtrow;              // per-row overrides or nullptr
  const float* detJv; float detJ; int time_dependent;
  float* lossVec;
  const float* Xb; const float* label; long nB, bDof; float biDimVal;   // BC/IC rows
  float w0, w1, w2;
  float* partial;    // [grid, P] per-workgroup gradient partials
  float* losspart;   // [grid, 3] per-workgroup (var, bc, ic) partial sums
  unsigned long long* stamps;   // diagnostic builds (-DVN_STAMPS) only: 8 phase cycle sums, else nullptr
  // de-duplicated formulation (8-wave kernel only; rows X = unique quadrature points):
  int mode;                     // 0 fused step | 1 forward only | 2 reverse pass with external seeds
  int dir;                      // tangent direction e_dir (modes 1, 2)
  int ostride;                  // element stride of out_ud / seed_ud
  float* out_u; float* out_ud;  // mode 1 outputs
  const float* seed_u; const float* seed_ud;   // mode 2 seeds (seed_u nullptr = 0; seed_ud nullptr = 1: the direction G carries them)
};
bool vn_fused_supported(const VnNet& net, int integ_num);
size_t vn_fused_lds_bytes(const VnNet& net);
hipError_t vn_fused_launch(const VnFusedArgs& a, int grid, hipStream_t s);
// 8-wave / 16x16x4 geometry of the same kernel (vn_fused16.hip)
bool vn_fused16_supported(const VnNet& net, int integ_num);
bool vn_fused16_net_supported(const VnNet& net);   // network instantiated (any integ_num: two-pass route)
size_t vn_fused16_lds_bytes(const VnNet& net);
int vn_fused16_ks(const VnNet& net);               // k-steps per hidden layer of the instantiation that serves `net` (0: none)
hipError_t vn_fused16_launch(const VnFusedArgs& a, int grid, hipStream_t s);

// ---- measurement aid (vn_calib.hip): sustained fp32 MFMA rate and fp32 vector issue cost of this GPU; out[5], see there
hipError_t vn_calibrate(int ncu, hipStream_t s, double out[5]);

// (value + input gradient at points: vn_pgrad16.h; de-duplicated weak-form assembly: vn_dedup.h; strong residual on the matrix
// pipe: vn_taylor16.h -- kept out of this header,
// which every kernel's source hash covers)

// ---- simple per-point evaluation kernels (float / double): vn_pointwise.hip --------------
hipError_t vn_pointwise_forward_f32(const VnNet& net, const float* theta, const float* X, long n,
                                    float* u, hipStream_t s);
hipError_t vn_pointwise_forward_f64(const VnNet& net, const double* theta, const double* X,
                                    long n, double* u, hipStream_t s);
hipError_t vn_pointwise_residual_f32(const VnNet& net, const float* theta, const float* X,
                                     const float* diff, const float* vel, const float* src,
                                     const float* ddx, int time_dependent, long n, float* u,
                                     float* res, hipStream_t s);
hipError_t vn_pointwise_residual_f64(const VnNet& net, const double* theta, const double* X,
                                     const double* diff, const double* vel, const double* src,
                                     const double* ddx, int time_dependent, long n, double* u,
                                     double* res, hipStream_t s);

// ---- layer-by-layer route for networks outside the kernels' range: vn_layered.hip ------------------
// Activations of a chunk of rows in HBM, one GEMM per layer over the stacked (value, tangent [, derivative]) streams
// (the MFMA products of vn_gemm.hip), elementwise kernels in between.  All calls enqueue on `s`; errors come back as a message.
struct VnLayered;
int vn_layered_create(VnLayered** out, const VnNet& net, char* err, size_t errlen);
void vn_layered_destroy(VnLayered* w);
// u (and ud along G, if both given) at seg.n rows
// keep_slot >= 0: keep the activations of these rows for the vn_layered_backward call with the same slot (if they fit)
int vn_layered_forward(VnLayered* w, const float* theta, const VnRows& seg, hipStream_t s, char* err, size_t errlen,
                       int keep_slot = -1);
// grad[0..P) += d loss / d theta from the rows of seg with seeds seg.ubar / seg.udbar (forward recomputed per chunk
// unless the slot holds the activations of exactly these rows)
int vn_layered_backward(VnLayered* w, const float* theta, const VnRows& seg, float* grad, hipStream_t s, char* err,
                        size_t errlen, int keep_slot = -1);
int vn_layered_forward_f64(VnLayered* w, const double* theta, const double* X, long n, double* u, hipStream_t s,
                           char* err, size_t errlen);
int vn_layered_residual_f32(VnLayered* w, const float* theta, const float* X, const float* diff, const float* vel,
                            const float* src, const float* ddx, int td, long n, float* u, float* res, hipStream_t s,
                            char* err, size_t errlen);
int vn_layered_residual_f64(VnLayered* w, const double* theta, const double* X, const double* diff, const double* vel,
                            const double* src, const double* ddx, int td, long n, double* u, double* res, hipStream_t s,
                            char* err, size_t errlen);

// ---- tile kernels of the layer-by-layer route for hidden widths <= 256: vn_wide.hip -------------------------------
// A workgroup carries 32 points through all layers with the activations in LDS; the forward stores (a, ad) of every layer
// in HBM for the reverse kernel.  vn_layered.hip hands qualifying networks over (VN_LAYERED_NOWIDE=1 keeps them on the GEMMs).
struct VnWide;
bool vn_wide_supported(const VnNet& net);
int vn_wide_create(VnWide** out, const VnNet& net, char* err, size_t errlen);
void vn_wide_destroy(VnWide* w);
// u (and ud) at seg.n rows; keep_slot >= 0 and may_keep: store the activations for vn_wide_backward (if they fit)
int vn_wide_forward(VnWide* w, const float* theta, const VnRows& seg, int keep_slot, bool may_keep, hipStream_t s, char* err,
                    size_t errlen);
bool vn_wide_has_kept(const VnWide* w, int slot, const VnRows& seg);
// grad[0..P) += d loss / d theta from the rows of seg (needs the activations stored by the forward of the same rows)
int vn_wide_backward(VnWide* w, const float* theta, const VnRows& seg, float* grad, int keep_slot, hipStream_t s, char* err,
                     size_t errlen);

// ---- fp32 GEMMs of the layer-by-layer route (widths beyond the tile kernels): vn_gemm.hip ---------------------------
// Row-major operands; return value = hipError_t of the launch (0 = ok).
int vn_gemm_nn(const float* A, const float* W, float* C, long M, int N, int K, hipStream_t s);          // C[M,N] = A[M,K] W[K,N]
// stacked forward with the layer epilogue: C[S][c][N] = (act(A0 W + b) | act'(.) (A1 W)), A = [S][c][K]
int vn_gemm_fwd(const float* A, const float* W, const float* bias, float* C, long c, int S, int N, int K, int act, hipStream_t s);
int vn_transpose(const float* W, float* Wt, int K, int N, hipStream_t s);                               // Wt[N,K] = W[K,N]^T
// parts[g][K1,N] = A_g^T Z_g over the rows g*rows .. min(M,(g+1)*rows) of A[M,K1], Z[M,N]; ceil(M/rows) groups
long vn_gemm_tn_rows(long M, int K1, int N, int ncu);       // rows per group that fill the chip evenly
int vn_gemm_tn_parts(const float* A, const float* Z, float* parts, long M, int K1, int N, long rows, hipStream_t s);
int vn_rowdot(const float* A, const float* w, float* y, long M, int H, float beta, hipStream_t s);      // y = beta y + A w
// fp64 forward product / output layer of the fp64 entry points (v_mfma_f64_16x16x4_f64)
int vn_dgemm_nn(const double* A, const double* W, double* C, long M, int N, int K, hipStream_t s);
int vn_drowdot(const double* A, const double* w, double* y, long M, int H, double beta, hipStream_t s);
