// C-ABI host layer of libvarnet_hip.so (see include/varnet_hip.h for the contract and the
// reference call sites each entry point replaces).
#include "vn_internal.h"
#include "vn_dedup.h"
#include "vn_pgrad16.h"
#include "vn_taylor16.h"
#include "vn_split16.h"

hipError_t vn_calibrate_f64(int ncu, hipStream_t s, double ghz, double out[3]);      // vn_calib.hip (fp64 MFMA loop)

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen'ed by vn_comm_*, never linked

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return fail(VN_EHIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

// (a failed call leaves earlier launches of the same step in flight: drain them before the caller unwinds)
#define LAYCHK(call)                                                          \
  do {                                                                        \
    char lerr_[384] = "";                                                      \
    (void)hipGetLastError();                                                  \
    if (call) {                                                               \
      (void)hipStreamSynchronize(h->stream);                                  \
      return fail(VN_EHIP, "layer-by-layer route: %s", lerr_);                \
    }                                                                         \
  } while (0)

struct Batch {
  const float* Input = nullptr;
  const float* gcoef = nullptr;
  const float* source = nullptr;
  const float* detJv = nullptr;
  const float* Nrow = nullptr;
  const float* dNtrow = nullptr;
  long n_k = 0;
  double detJ = 0.0;
  bool set = false;
  // optional per-batch copy of the BC/IC rows (the reference's shuffle permutes them per feed: vn_set_batch_bic)
  const float* biInput = nullptr;
  const float* biLabel = nullptr;
  // de-duplicated formulation (vn_set_dedup)
  const float* Xu = nullptr;
  const int* uid = nullptr;
  const int* rowptr = nullptr;
  const int* rowidx = nullptr;
  long U = 0;
  float* gcsr = nullptr;      // owned: gcoef in CSR order, [n_k*integ_num, dim] (static per batch; built by vn_set_dedup)
  long gcsr_cap = 0;
  bool gper = false;          // gcoef repeats with period integ_num along the rows (constant coefficients): no CSR copy needed
};

constexpr int PROF_CAP = 4096;

#ifdef VN_WITH_FUSED32
constexpr bool kWithFused32 = true;
#else
constexpr bool kWithFused32 = false;
#endif
#ifdef VN_XCHECK_F32_POINT
constexpr bool kWithF32Point = true;     // cross-check build: vn_pgrad16 / vn_taylor16 also instantiated for the nets vn_split16 serves
#else
constexpr bool kWithF32Point = false;
#endif

}  // namespace

#ifndef VN_WITH_FUSED32
// Product build: the 4-wave geometry (vn_fused.hip) is not linked -- it serves no automatic route (every network it takes is
// one of the 8-wave kernel's).  It lives in the tests' cross-check library (make xcheck: -DVN_WITH_FUSED32 + vn_fused.o).
bool vn_fused_supported(const VnNet&, int) { return false; }
hipError_t vn_fused_launch(const VnFusedArgs&, int, hipStream_t) { return hipErrorInvalidValue; }
#endif


struct vn_engine {
  vn_config cfg{};
  VnNet net{};
  hipStream_t stream = nullptr;
  int ncu = 256;

  float *theta = nullptr, *m = nullptr, *v = nullptr;
  double* theta64 = nullptr;
  float* gradbuf_int = nullptr;
  float* gradbuf = nullptr;
  float* lossbuf = nullptr;       // [4] for vn_eval_loss
  float* partial = nullptr;       // [bwd_grid, P]
  int bwd_grid = 0, fwd_grid = 0;

  float *feN = nullptr, *fedNt = nullptr, *feW = nullptr;
  bool has_fe = false, has_feW = false;

  std::vector<Batch> batches;
  const float *biInput = nullptr, *biLabel = nullptr;
  long nB = 0, bDof = 0;
  double biDimVal = 1.0;
  double w[3] = {1.0, 1.0, 1.0};

  float *u = nullptr, *ud = nullptr, *ubar = nullptr, *udbar = nullptr;
  long work_rows = 0;
  float *ub = nullptr, *ubar_b = nullptr;
  long work_b = 0;
  float* losspart = nullptr;
  long losspart_cap = 0;

  int64_t step = 0;
  bool use_fused = false;
  bool use_fused16 = false;
  bool full_grid = false;            // VN_FULL_GRID=1 (diagnostic): #CU workgroups whatever the tile count (fixed-cost measurements)
  VnOptArgs fuse;                    // optimizer step to fold into the next gradient reduction (kind -1: none)
  bool two_pass = false;             // fused kernel twice around the row-wise seed kernel (integ_num > 128)
  bool fused_only = false;           // 7-8 hidden layers: no generic kernels for this net
  VnLayered* layered = nullptr;      // layer-by-layer route (networks outside the kernels' range, or forced)
  float* tp_losspart = nullptr; long tp_losspart_cap = 0;
  float* fused_losspart = nullptr;   // [ncu*3]
  unsigned long long* stamps = nullptr;   // 8 counters, diagnostic builds
  // de-duplicated formulation work buffers
  float *dd_uv = nullptr, *dd_ug = nullptr, *dd_su = nullptr, *dd_sg = nullptr, *dd_partial = nullptr,
        *dd_losspart = nullptr;
  long dd_capU = 0, dd_cap_lp = 0;
  float* snap = nullptr;       // vn_state_snapshot: device copy of (theta | m | v), 3 P floats
  int64_t snap_step = -1;      // step counter at the snapshot (-1: none)
  bool point_kernels = false;  // vn_debug_point_route(1): vn_residual / vn_*_f64 on the per-thread kernels (the tests' cross-check)
  bool eval_rowwise = false;   // vn_debug_point_route(route | 8): vn_eval_loss on the row-wise forward although the batch carries a de-duplication map
  bool no_gtable = false;      // vn_debug_point_route(route | 4): vn_set_dedup keeps the CSR-ordered copy of gcoef although it is periodic
  bool no_split = false;       // vn_debug_point_route(2): the f32-MFMA point kernels where the bf16-piece kernels (vn_split16.hip) would run
  int pgrad_wgs = 0;           // workgroups per CU of vn_pgrad16: 0 = what fits, at most 2 (diagnostic override: $VN_PGRAD_WGS = 1..4)

  // tower gradient SUM over RCCL (vn_comm_init); nullptr = single process or host-side collective
  ncclComm_t comm = nullptr;
  int comm_world = 1, comm_rank = 0;
  // ncclCommInitRank has no timeout: a caller may run vn_comm_init on a helper thread and give up on it (vn_comm_abandon).
  // `comm` is committed / withdrawn under this mutex only, so the thread that trains never sees a communicator appear late.
  std::mutex comm_mu;
  bool comm_abandoned = false;
  ncclComm_t comm_orphan = nullptr;     // a communicator that came up after (or was up at) abandonment: never used, never destroyed

  // profiling of the dominant kernel
  bool prof_on = false;
  int prof_n = 0;
  std::vector<hipEvent_t> ev0, ev1;
  std::vector<hipEvent_t> cev0, cev1;   // around the all-reduce
  int cprof_n = 0;
  std::string prof_name = "vn_generic_bwd_kernel";
};

namespace {

int build_net(const vn_config& c, VnNet& net) {
  if (c.n_layers < 1 || c.n_layers > VN_MAX_LAYERS)
    return fail(VN_EINVAL, "n_layers=%d outside [1,%d]", c.n_layers, VN_MAX_LAYERS);
  if (c.d_in < 1 || c.d_in > VN_MAX_DIN) return fail(VN_EINVAL, "d_in=%d outside [1,%d]", c.d_in, VN_MAX_DIN);
  if (c.dim < 1 || c.dim > c.d_in) return fail(VN_EINVAL, "dim=%d must be in [1,d_in]", c.dim);
  if (c.integ_num < 1) return fail(VN_EINVAL, "integ_num must be positive");
  if (c.activation != VN_ACT_SIGMOID && c.activation != VN_ACT_TANH && c.activation != VN_ACT_PER_LAYER)
    return fail(VN_EUNSUPPORTED, "activation must be sigmoid or tanh (VarNet.py:97)");
  if (c.optimizer != VN_OPT_ADAM && c.optimizer != VN_OPT_RMSPROP) return fail(VN_EINVAL, "unknown optimizer requested!");
  if (c.lr < 0.0) return fail(VN_EINVAL, "learning rate must be positive!");  // TFModel.py:130
  if (c.optimizer == VN_OPT_ADAM) {
    // taken literally, never defaulted: a zero-initialised config (eps = 0: 0/0 in the update of a zero-gradient
    // parameter) is an error, not a NaN three steps later
    if (!(c.beta1 >= 0.0 && c.beta1 < 1.0) || !(c.beta2 >= 0.0 && c.beta2 < 1.0))
      return fail(VN_EINVAL, "Adam beta1 = %g, beta2 = %g must lie in [0, 1) (TF-1 defaults 0.9, 0.999; the struct is not defaulted)",
                  c.beta1, c.beta2);
    if (!(c.eps > 0.0))
      return fail(VN_EINVAL, "Adam epsilon = %g must be positive (TF-1 default 1e-8; the struct is not defaulted)", c.eps);
  }
  memset(&net, 0, sizeof net);
  net.d_in = c.d_in;
  net.dim = c.dim;
  net.L = c.n_layers;
  net.H[0] = c.d_in;
  int off = 0, hmax = 0;
  for (int l = 1; l <= net.L + 1; ++l) {
    const int h = (l <= net.L) ? c.widths[l - 1] : 1;
    if (h < 1 || h > VN_MAX_WIDTH) return fail(VN_EINVAL, "layer width %d outside [1,%d]", h, VN_MAX_WIDTH);
    net.H[l] = h;
    net.woff[l] = off;
    off += net.H[l - 1] * h;
    net.boff[l] = off;
    off += h;
    if (l <= net.L && h > hmax) hmax = h;
  }
  net.P = off;
  net.hmax = hmax;
  // per-layer list (TFModel.py:113-119): a list whose entries agree is the uniform case
  bool mixed = false;
  for (int l = 1; l <= net.L; ++l) {
    const int a = (c.activation == VN_ACT_PER_LAYER) ? c.layer_act[l - 1] : c.activation;
    if (a != VN_ACT_SIGMOID && a != VN_ACT_TANH) return fail(VN_EUNSUPPORTED, "activation must be sigmoid or tanh (VarNet.py:97)");
    net.actl[l] = a;
    if (a != net.actl[1]) mixed = true;
  }
  net.act = mixed ? VN_ACT_PER_LAYER : net.actl[1];
  return VN_OK;
}

int ensure(float** p, long* cap, long need) {
  if (need <= *cap) return VN_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  HIPCHK(hipMalloc((void**)p, (size_t)need * sizeof(float)));
  *cap = need;
  return VN_OK;
}

uint64_t splitmix64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int refresh_theta64(vn_engine* h) {
  // fp32 parameters widened on the host side of the stream: small, off the hot path
  std::vector<float> t(h->net.P);
  HIPCHK(hipMemcpyAsync(t.data(), h->theta, t.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<double> d(t.begin(), t.end());
  if (!h->theta64) HIPCHK(hipMalloc((void**)&h->theta64, d.size() * sizeof(double)));
  HIPCHK(hipMemcpyAsync(h->theta64, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VN_OK;
}

int check_batch(vn_engine* h, int32_t batch) {
  if (batch < 0 || batch >= (int)h->batches.size() || !h->batches[batch].set)
    return fail(VN_ESTATE, "batch %d has no interior data (call vn_set_interior first)", batch);
  if (!h->has_fe && !(h->batches[batch].Nrow && h->batches[batch].dNtrow))
    return fail(VN_ESTATE, "FE tables missing (call vn_set_fe_table first)");
  return VN_OK;
}

inline const float* bi_x(const vn_engine* h, const Batch& b) { return b.biInput ? b.biInput : h->biInput; }
inline const float* bi_y(const vn_engine* h, const Batch& b) { return b.biLabel ? b.biLabel : h->biLabel; }

// Model value (and directional derivative along G, if given) at n rows with the 8-wave fused kernel in its
// forward-only mode: 2 F_pt per row at the fused kernel's efficiency instead of the generic forward kernel.
int fused_forward(vn_engine* h, const float* X, const float* G, long n, float* out_u, float* out_ud) {
  if (n <= 0) return VN_OK;
  VnFusedArgs f{};
  f.net = h->net; f.theta = h->theta; f.X = X; f.G = G; f.src = nullptr;
  f.nT = n; f.n_k = 0; f.integ_num = h->cfg.integ_num;
  f.feN = h->feN; f.fedNt = h->fedNt; f.feW = nullptr; f.detJv = nullptr; f.detJ = 0.f;
  f.time_dependent = h->cfg.time_dependent; f.lossVec = nullptr;
  f.Xb = nullptr; f.label = nullptr; f.nB = 0; f.bDof = 0; f.biDimVal = 0.f;
  f.w0 = f.w1 = f.w2 = 0.f;
  f.partial = h->partial; f.losspart = h->fused_losspart ? h->fused_losspart : h->tp_losspart; f.stamps = nullptr;
  f.mode = 1; f.dir = G ? -1 : 0; f.ostride = 1; f.out_u = out_u; f.out_ud = G ? out_ud : nullptr;
  if (!f.losspart || !f.partial) return fail(VN_ESTATE, "fused forward without its work buffers");   // the kernel stores to both
  const long tiles = (n + 127) / 128;
  const int grid = (int)(tiles < h->ncu ? tiles : h->ncu);
  HIPCHK(vn_fused16_launch(f, grid, h->stream));
  return VN_OK;
}

// Loss components and loss field of a batch that carries a de-duplication map (vn_set_dedup), without the row-wise forward: (u, grad u)
// once per unique point (2 F_pt per POINT where the forward-only mode of the fused kernel costs 2 F_pt per ROW), the assembly kernel of
// the training step in its loss-only form (R_k, lossVec, variational partials), the BC/IC rows through the forward-only mode and the
// row-wise seed kernel with an empty interior set.  What every monitor of a run on the de-duplicated formulation calls (splitLoss,
// VarNet.py:1365): 2.9 -> 0.45 ms on BASELINE config 3.
int eval_dedup(vn_engine* h, const Batch& b, float* lossVec, float* lossdst) {
  const int q = h->cfg.integ_num, dim = h->cfg.dim;
  const int sblk = (int)((b.n_k + VN_DEDUP_TFB - 1) / VN_DEDUP_TFB);
  const int bgrid = (int)(((h->nB > 0 ? h->nB : 1) + 255) / 256);
  if (int rc = ensure(&h->losspart, &h->losspart_cap, (long)(sblk + bgrid) * 3)) return rc;
  if (!h->no_split && vn_split16_supported(h->net)) HIPCHK(vn_split16_pgrad(h->net, h->theta, b.Xu, b.U, nullptr, nullptr, h->dd_uv, h->ncu, h->stream));
  else HIPCHK(vn_pgrad16_launch(h->net, h->theta, b.Xu, b.U, nullptr, nullptr, h->dd_uv, h->ncu, h->pgrad_wgs, h->stream));
  VnDedupArgs a{};
  a.upack = h->dd_uv; a.uid = b.uid; a.rowptr = b.rowptr; a.rowidx = b.rowidx;
  a.gcoef = b.gcoef; a.gcoef_csr = b.gcsr; a.source = h->cfg.has_source ? b.source : nullptr;
  a.feN = h->feN; a.fedNt = h->fedNt; a.feW = (h->cfg.has_integw && h->has_feW) ? h->feW : nullptr;
  a.detJv = b.detJv; a.detJ = (float)b.detJ; a.n_k = b.n_k; a.U = b.U; a.q = q; a.dim = dim;
  a.time_dependent = h->cfg.time_dependent; a.w2 = (float)h->w[2]; a.gper = b.gper ? 1 : 0;
  a.stf = nullptr; a.lossVec = lossVec; a.part = h->losspart;          // loss only: no seeds
  a.seed_u = nullptr; a.seed_g = nullptr;
  HIPCHK(vn_dedup_seed_launch(a, sblk, h->stream));
  if (int rc = fused_forward(h, bi_x(h, b), nullptr, h->nB, h->ub, nullptr)) return rc;
  VnSeedArgs s{};
  s.u = h->u; s.ud = h->ud; s.source = nullptr; s.feN = h->feN; s.fedNt = h->fedNt; s.feW = nullptr;
  s.Nrow = nullptr; s.dNtrow = nullptr; s.detJv = nullptr; s.detJ = (float)b.detJ;
  s.n_k = 0; s.integ_num = q; s.time_dependent = h->cfg.time_dependent;      // interior set empty: the BC/IC terms alone
  s.ubar = nullptr; s.udbar = nullptr; s.lossVec = nullptr;
  s.ub = h->ub; s.label = bi_y(h, b); s.nB = h->nB; s.bDof = h->bDof; s.biDimVal = (float)h->biDimVal; s.ubar_b = nullptr;
  s.w0 = (float)h->w[0]; s.w1 = (float)h->w[1]; s.w2 = (float)h->w[2];
  s.part = h->losspart + (long)sblk * 3;
  HIPCHK(vn_seed_launch(s, bgrid, h->stream));
  if (lossdst)
    HIPCHK(vn_reduce_launch(nullptr, 0, 0, h->losspart, sblk + bgrid, h->bDof, h->nB, s.w0, s.w1, s.w2, lossdst, h->stream));
  return VN_OK;
}

// forward + weak-form epilogue; with_seeds = also produce backward seeds.
int run_forward_and_seed(vn_engine* h, const Batch& b, bool with_seeds, float* lossVec, float* lossdst) {
  const long nT = b.n_k * h->cfg.integ_num;
  if (!with_seeds && b.Xu && !h->layered && (h->use_fused16 || h->two_pass) && h->has_fe && !h->eval_rowwise)
    return eval_dedup(h, b, lossVec, lossdst);
  if (h->layered) {
    VnRows s0{}, s1{};
    s0.X = b.Input; s0.G = b.gcoef; s0.u = h->u; s0.ud = h->ud; s0.n = nT;
    s1.X = bi_x(h, b); s1.G = nullptr; s1.u = h->ub; s1.ud = nullptr; s1.n = h->nB;
    LAYCHK(vn_layered_forward(h->layered, h->theta, s0, h->stream, lerr_, sizeof lerr_, with_seeds ? 0 : -1));
    LAYCHK(vn_layered_forward(h->layered, h->theta, s1, h->stream, lerr_, sizeof lerr_, with_seeds ? 1 : -1));
  } else if (((h->use_fused16 || h->two_pass) && h->has_fe && !with_seeds) || h->fused_only) {
    // splitLoss / trainWeight / the monitors: the fused kernel's forward-only mode for both row sets
    if (int rc = fused_forward(h, b.Input, b.gcoef, nT, h->u, h->ud)) return rc;
    if (int rc = fused_forward(h, bi_x(h, b), nullptr, h->nB, h->ub, nullptr)) return rc;
  } else {
    VnRows s0{}, s1{};
    s0.X = b.Input; s0.G = b.gcoef; s0.u = h->u; s0.ud = h->ud; s0.n = nT;
    s1.X = bi_x(h, b); s1.G = nullptr; s1.u = h->ub; s1.ud = nullptr; s1.n = h->nB;
    HIPCHK(vn_generic_forward(h->net, h->theta, s0, s1, h->fwd_grid, h->stream));
  }

  const long nthreads = b.n_k > h->nB ? b.n_k : h->nB;
  const int grid = (int)(((nthreads > 0 ? nthreads : 1) + 255) / 256);      // an empty set still zeroes its partials
  if (int rc = ensure(&h->losspart, &h->losspart_cap, (long)grid * 3)) return rc;
  VnSeedArgs a{};
  a.u = h->u; a.ud = h->ud; a.source = h->cfg.has_source ? b.source : nullptr;
  a.feN = h->feN; a.fedNt = h->fedNt; a.feW = (h->cfg.has_integw && h->has_feW) ? h->feW : nullptr;
  a.Nrow = b.Nrow; a.dNtrow = b.dNtrow;
  a.detJv = b.detJv; a.detJ = (float)b.detJ;
  a.n_k = b.n_k; a.integ_num = h->cfg.integ_num; a.time_dependent = h->cfg.time_dependent;
  a.ubar = with_seeds ? h->ubar : nullptr; a.udbar = with_seeds ? h->udbar : nullptr;
  a.lossVec = lossVec;
  a.ub = h->ub; a.label = bi_y(h, b); a.nB = h->nB; a.bDof = h->bDof; a.biDimVal = (float)h->biDimVal;
  a.ubar_b = with_seeds ? h->ubar_b : nullptr;
  a.w0 = (float)h->w[0]; a.w1 = (float)h->w[1]; a.w2 = (float)h->w[2];
  a.part = h->losspart;
  HIPCHK(vn_seed_launch(a, grid, h->stream));
  if (lossdst) {
    HIPCHK(vn_reduce_launch(nullptr, 0, 0, h->losspart, grid, h->bDof, h->nB, a.w0, a.w1, a.w2, lossdst, h->stream));
  }
  return VN_OK;
}

// Test functions that do not fit one 128-point tile (integNum 216: 3-point Gauss in 2D+t) cannot have their
// R_k formed inside a tile.  Two launches of the 8-wave fused kernel around the row-wise seed kernel:
//   1. forward only  -> u, directional derivative per row          (2 F_pt)
//   2. vn_seed_kernel -> R_k, lossVec, variational loss partials, per-row seeds
//   3. reverse pass with those seeds (recomputes the forward); BC/IC tiles ride along   (6 F_pt)
// 8 F_pt per point instead of 6, against 8 F_pt at 0.07 of peak on the generic kernels.
int run_twopass(vn_engine* h, const Batch& b, float* gradbuf) {
  const int grid = h->ncu, P = h->net.P, q = h->cfg.integ_num;
  const long nT = b.n_k * q;
  const int sgrid = (int)((b.n_k + 255) / 256);
  if (int rc = ensure(&h->tp_losspart, &h->tp_losspart_cap, (long)(grid + sgrid) * 3)) return rc;
  float* lp = h->tp_losspart;
  VnFusedArgs f{};
  f.net = h->net; f.theta = h->theta; f.X = b.Input; f.G = b.gcoef; f.src = nullptr;
  f.nT = nT; f.n_k = 0; f.integ_num = q;
  f.feN = h->feN; f.fedNt = h->fedNt; f.feW = nullptr; f.detJv = nullptr; f.detJ = 0.f;
  f.time_dependent = h->cfg.time_dependent; f.lossVec = nullptr;
  f.Xb = bi_x(h, b); f.label = bi_y(h, b); f.nB = 0; f.bDof = h->bDof; f.biDimVal = (float)h->biDimVal;
  f.w0 = (float)h->w[0]; f.w1 = (float)h->w[1]; f.w2 = (float)h->w[2];
  f.partial = h->partial; f.losspart = lp; f.stamps = nullptr;
  f.dir = -1; f.ostride = 1;
  f.mode = 1; f.out_u = h->u; f.out_ud = h->ud;
  HIPCHK(vn_fused16_launch(f, grid, h->stream));

  VnSeedArgs a{};
  a.u = h->u; a.ud = h->ud; a.source = h->cfg.has_source ? b.source : nullptr;
  a.feN = h->feN; a.fedNt = h->fedNt; a.feW = (h->cfg.has_integw && h->has_feW) ? h->feW : nullptr;
  a.Nrow = b.Nrow; a.dNtrow = b.dNtrow;
  a.detJv = b.detJv; a.detJ = (float)b.detJ;
  a.n_k = b.n_k; a.integ_num = q; a.time_dependent = h->cfg.time_dependent;
  a.ubar = h->ubar; a.udbar = h->udbar; a.lossVec = nullptr;
  a.ub = nullptr; a.label = nullptr; a.nB = 0; a.bDof = 0; a.biDimVal = 0.f; a.ubar_b = nullptr;   // BC/IC: step 3
  a.w0 = f.w0; a.w1 = f.w1; a.w2 = f.w2;
  a.part = lp + (long)grid * 3;
  HIPCHK(vn_seed_launch(a, sgrid, h->stream));

  f.mode = 2; f.out_u = nullptr; f.out_ud = nullptr; f.seed_u = h->ubar; f.seed_ud = h->udbar; f.nB = h->nB;
  const bool rec = h->prof_on && h->prof_n < PROF_CAP;
  if (rec) {
    if (!h->ev0[h->prof_n]) { HIPCHK(hipEventCreate(&h->ev0[h->prof_n])); HIPCHK(hipEventCreate(&h->ev1[h->prof_n])); }
    HIPCHK(hipEventRecord(h->ev0[h->prof_n], h->stream));
  }
  HIPCHK(vn_fused16_launch(f, grid, h->stream));
  if (rec) { HIPCHK(hipEventRecord(h->ev1[h->prof_n], h->stream)); h->prof_n++; }
  HIPCHK(vn_reduce_launch(h->partial, grid, P, lp, grid + sgrid, h->bDof, h->nB, f.w0, f.w1, f.w2, gradbuf, h->stream, h->fuse));
  return VN_OK;
}

// One gradient evaluation in the de-duplicated formulation (vn_dedup.hip header), 8 F_pt per unique point:
//   1. (u, du/dx_d) at the unique points: value forward + value-adjoint sweep to the inputs     (2 F_pt, vn_pgrad16.hip)
//   2. weak-form assembly over (test function, quadrature point) rows -> R_k, loss, per-row seeds
//   3. seed gather per unique point: su = d loss / d u, sg[d] = d loss / d u_{x_d}
//   4. ONE reverse launch of the fused kernel (recomputes the forward): the directional derivative is linear in its
//      direction, sum_d sg_d * d(u_{x_d})/d theta = d(sg . grad u)/d theta with sg held fixed, so the per-point direction
//      G = sg with tangent seed 1 and value seed su gives the whole gradient; BC/IC tiles ride along       (6 F_pt)
int run_dedup(vn_engine* h, const Batch& b, float* gradbuf) {
  const int dim = h->cfg.dim, q = h->cfg.integ_num, grid = h->ncu, P = h->net.P;
  const int sblk = (int)((b.n_k + VN_DEDUP_TFB - 1) / VN_DEDUP_TFB);
  float* lp = h->dd_losspart;                       // [grid + sblk][3]
  // vn_profile_*: HIP events around the formulation's whole kernel sequence (steps 1-4; the reduction stays outside as
  // in the row-wise step)
  const bool rec = h->prof_on && h->prof_n < PROF_CAP;
  if (rec) {
    if (!h->ev0[h->prof_n]) { HIPCHK(hipEventCreate(&h->ev0[h->prof_n])); HIPCHK(hipEventCreate(&h->ev1[h->prof_n])); }
    HIPCHK(hipEventRecord(h->ev0[h->prof_n], h->stream));
  }
  if (!h->no_split && vn_split16_supported(h->net)) HIPCHK(vn_split16_pgrad(h->net, h->theta, b.Xu, b.U, nullptr, nullptr, h->dd_uv, grid, h->stream));
  else HIPCHK(vn_pgrad16_launch(h->net, h->theta, b.Xu, b.U, nullptr, nullptr, h->dd_uv, grid, h->pgrad_wgs, h->stream));
  VnDedupArgs a{};
  a.upack = h->dd_uv; a.uid = b.uid; a.rowptr = b.rowptr; a.rowidx = b.rowidx;
  a.gcoef = b.gcoef; a.gcoef_csr = b.gcsr; a.source = h->cfg.has_source ? b.source : nullptr;
  a.feN = h->feN; a.fedNt = h->fedNt; a.feW = (h->cfg.has_integw && h->has_feW) ? h->feW : nullptr;
  a.detJv = b.detJv; a.detJ = (float)b.detJ; a.n_k = b.n_k; a.U = b.U; a.q = q; a.dim = dim;
  a.time_dependent = h->cfg.time_dependent; a.w2 = (float)h->w[2]; a.gper = b.gper ? 1 : 0;
  a.stf = h->u; a.lossVec = nullptr; a.part = lp + (long)grid * 3;
  a.seed_u = h->dd_su; a.seed_g = h->dd_sg;
  HIPCHK(vn_dedup_seed_launch(a, sblk, h->stream));
  HIPCHK(vn_dedup_gather_launch(a, h->stream));
  VnFusedArgs f{};
  f.net = h->net; f.theta = h->theta; f.X = b.Xu; f.G = h->dd_sg; f.src = nullptr;
  f.nT = b.U; f.n_k = 0; f.integ_num = q;
  f.feN = h->feN; f.fedNt = h->fedNt; f.feW = nullptr; f.detJv = nullptr; f.detJ = 0.f;
  f.time_dependent = h->cfg.time_dependent; f.lossVec = nullptr;
  f.Xb = bi_x(h, b); f.label = bi_y(h, b); f.nB = h->nB; f.bDof = h->bDof; f.biDimVal = (float)h->biDimVal;
  f.w0 = (float)h->w[0]; f.w1 = (float)h->w[1]; f.w2 = (float)h->w[2];
  f.partial = h->dd_partial; f.losspart = lp; f.stamps = nullptr;
  f.mode = 2; f.dir = -1; f.ostride = 1; f.out_u = nullptr; f.out_ud = nullptr;
  f.seed_u = h->dd_su; f.seed_ud = nullptr;          // tangent seed 1
  HIPCHK(vn_fused16_launch(f, grid, h->stream));
  if (rec) { HIPCHK(hipEventRecord(h->ev1[h->prof_n], h->stream)); h->prof_n++; }
  HIPCHK(vn_reduce_launch(h->dd_partial, grid, P, lp, grid + sblk, h->bDof, h->nB, f.w0, f.w1, f.w2, gradbuf, h->stream, h->fuse));
  return VN_OK;
}

// ---- RCCL, loaded at run time -------------------------------------------------------------
// librccl.so.1 is resolved by SONAME, so a process that already carries RCCL (PyTorch-ROCm does) shares that
// copy; VN_RCCL_LIB names another file.  Nothing here is touched unless vn_comm_* is called, so the library
// loads (and every other entry point works) on a machine without RCCL.
struct Rccl {
  void* dl = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.dl) return VN_OK;
  // $VN_RCCL_LIB names THE library to use (no fall-through to another copy: a host that points at a specific build
  // must not silently get a different one); otherwise the SONAME a PyTorch-ROCm process already carries, then /opt/rocm
  const char* user = getenv("VN_RCCL_LIB");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* dl = nullptr;
  if (user && *user) {
    dl = dlopen(user, RTLD_NOW | RTLD_LOCAL);
    if (!dl) return fail(VN_EUNSUPPORTED, "RCCL: VN_RCCL_LIB=%s cannot be loaded (%s)", user, dlerror());
  } else {
    for (const char* n : names) {
      dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (dl) break;
    }
  }
  if (!dl) return fail(VN_EUNSUPPORTED, "RCCL not found (%s): set VN_RCCL_LIB", dlerror());
  Rccl r;
  r.dl = dl;
#define VN_SYM(field, name)                                                          \
  r.field = (decltype(r.field))dlsym(dl, name);                                      \
  if (!r.field) { dlclose(dl); return fail(VN_EUNSUPPORTED, "RCCL symbol %s missing", name); }
  VN_SYM(GetUniqueId, "ncclGetUniqueId")
  VN_SYM(CommInitRank, "ncclCommInitRank")
  VN_SYM(CommDestroy, "ncclCommDestroy")
  VN_SYM(CommCount, "ncclCommCount")
  VN_SYM(AllReduce, "ncclAllReduce")
  VN_SYM(GetErrorString, "ncclGetErrorString")
  VN_SYM(GetVersion, "ncclGetVersion")
#undef VN_SYM
  g_rccl = r;
  return VN_OK;
}

// (the collective library probes peers / IPC handles with HIP calls that may fail benignly; their stale last-error
// must not be reported by the launch check of the next kernel of this library)
#define RCCLCHK(expr)                                                                        \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    (void)hipGetLastError();                                                                 \
    if (r_ != ncclSuccess) return fail(VN_ECOMM, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)

}  // namespace

extern "C" {

const char* vn_last_error(void) { return g_err.c_str(); }
int vn_abi_version(void) { return VN_ABI_VERSION; }   // 7: vn_comm_abandon; 6: vn_forward_grad; 5: vn_comm_version; 4: vn_comm_available, validated Adam hyper-parameters (3: vn_config.widths[16], VN_KERNEL_LAYERED)

int vn_create(const vn_config* cfg, vn_engine** out) {
  if (!cfg || !out) return fail(VN_EINVAL, "null argument");
  *out = nullptr;
  VnNet net;
  if (int rc = build_net(*cfg, net)) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(VN_EHIP, "no HIP device available (%s): the VarNet engine has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(VN_EINVAL, "requested processor %d is unavailable!", cfg->device);
  HIPCHK(hipSetDevice(cfg->device));
  // Route.  Networks outside the kernels' range (VN_KMAX_*), and nets whose generic-kernel tile does not fit LDS while
  // no fused instantiation exists, go layer by layer (vn_layered.hip); VN_KERNEL_LAYERED forces that route.  One
  // extension of the range: 7 and 8 hidden layers up to 50 wide are instantiated in the 8-wave fused kernel (deep,
  // narrow nets); the generic kernels do not cover them, so every path of such an engine runs on the fused kernel.
  const bool generic_range = vn_net_in_kernel_range(net);
  const bool deep_fused = !generic_range && net.L <= 8 && net.hmax <= VN_KMAX_WIDTH && net.d_in <= VN_KMAX_DIN &&
                          net.act != VN_ACT_PER_LAYER && vn_fused16_net_supported(net) &&
                          (cfg->kernel == VN_KERNEL_AUTO || cfg->kernel == VN_KERNEL_FUSED16);
  const bool in_range = generic_range || deep_fused;
  if (!in_range && cfg->kernel != VN_KERNEL_AUTO && cfg->kernel != VN_KERNEL_LAYERED)
    return fail(VN_EUNSUPPORTED, "network (%d layers, widest %d, %d inputs%s) is outside the range of the requested kernel family "
                "(<= %d layers, width <= %d, <= %d inputs, one activation): use VN_KERNEL_AUTO or VN_KERNEL_LAYERED",
                net.L, net.hmax, net.d_in, net.act == VN_ACT_PER_LAYER ? ", mixed activations" : "", VN_KMAX_LAYERS,
                VN_KMAX_WIDTH, VN_KMAX_DIN);
  const bool fused_ok = in_range && cfg->kernel != VN_KERNEL_GENERIC && cfg->kernel != VN_KERNEL_FUSED &&
                        cfg->kernel != VN_KERNEL_LAYERED && vn_fused16_net_supported(net);
  bool use_layered = cfg->kernel == VN_KERNEL_LAYERED || !in_range;
  if (!use_layered && !fused_ok && vn_generic_bwd_lds_bytes(net) > 160 * 1024) {
    if (cfg->kernel == VN_KERNEL_AUTO) use_layered = true;
    else
      return fail(VN_EUNSUPPORTED, "network needs %zu B of LDS per tile on the generic kernels (> 160 KiB): reduce depth/width",
                  vn_generic_bwd_lds_bytes(net));
  }
  vn_engine* h = new vn_engine();
  h->cfg = *cfg;              // taken literally (lr = 0 is a legal, if useless, TF learning rate: TFModel.py:130)
  h->net = net;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) h->ncu = prop.multiProcessorCount;
  const size_t P = net.P;
  if (use_layered) {
    h->fwd_grid = h->bwd_grid = 1;            // one gradient vector, no per-workgroup partials
  } else {
    const size_t fwd_lds = vn_generic_fwd_lds_bytes(net), bwd_lds = vn_generic_bwd_lds_bytes(net);
    int fpc = (int)((160 * 1024) / fwd_lds); if (fpc > 4) fpc = 4; if (fpc < 1) fpc = 1;
    int bpc = (int)((160 * 1024) / bwd_lds); if (bpc > 2) bpc = 2; if (bpc < 1) bpc = 1;
    h->fwd_grid = h->ncu * fpc;
    h->bwd_grid = h->ncu * bpc;
  }
  hipError_t a = hipSuccess;
  if (a == hipSuccess) a = hipMalloc((void**)&h->theta, P * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->m, P * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->v, P * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->gradbuf_int, (P + 4) * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->lossbuf, 4 * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->partial, (size_t)h->bwd_grid * P * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->feN, cfg->integ_num * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->fedNt, cfg->integ_num * sizeof(float));
  if (a == hipSuccess) a = hipMalloc((void**)&h->feW, cfg->integ_num * sizeof(float));
  if (a == hipSuccess) a = hipMemset(h->theta, 0, P * sizeof(float));
  if (a == hipSuccess) a = hipMemset(h->m, 0, P * sizeof(float));
  if (a == hipSuccess) a = hipMemset(h->v, 0, P * sizeof(float));
  if (a == hipSuccess) a = hipMemset(h->gradbuf_int, 0, (P + 4) * sizeof(float));
  if (a != hipSuccess) {
    vn_destroy(h);
    return fail(VN_ENOMEM, "device allocation failed: %s", hipGetErrorString(a));
  }
  h->gradbuf = h->gradbuf_int;
  if (use_layered) {
    char lerr[384] = "";
    if (vn_layered_create(&h->layered, net, lerr, sizeof lerr)) {
      vn_destroy(h);
      return fail(VN_EUNSUPPORTED, "layer-by-layer route unavailable: %s", lerr);
    }
    h->prof_name = "vn_layered_backward";
    h->ev0.resize(PROF_CAP, nullptr);
    h->ev1.resize(PROF_CAP, nullptr);
    h->cev0.resize(PROF_CAP, nullptr);
    h->cev1.resize(PROF_CAP, nullptr);
    *out = h;
    return VN_OK;
  }
  if (cfg->kernel == VN_KERNEL_FUSED && !vn_fused_supported(net, cfg->integ_num)) {
    vn_destroy(h);
    return fail(VN_EUNSUPPORTED, kWithFused32 ? "fused kernel unsupported for this network / integ_num"
                                              : "VN_KERNEL_FUSED (the 4-wave geometry) is not part of the product library: it lives in the "
                                                "tests' cross-check build, libvarnet_hip_xcheck.so (make -C varnet_amd/csrc xcheck)");
  }
  const bool tp_ok = cfg->integ_num > 128 && vn_fused16_net_supported(net);
  if (cfg->kernel == VN_KERNEL_FUSED16 && !vn_fused16_supported(net, cfg->integ_num) && !tp_ok) {
    vn_destroy(h);
    return fail(VN_EUNSUPPORTED, "fused16 kernel unsupported for this network / integ_num");
  }
  // AUTO: the 8-wave geometry where instantiated (faster: two waves per SIMD overlap VALU/LDS work
  // with MFMA), else the 4-wave geometry, else the generic kernels
  h->use_fused16 = (cfg->kernel == VN_KERNEL_FUSED16 || cfg->kernel == VN_KERNEL_AUTO) &&
                   vn_fused16_supported(net, cfg->integ_num);
  h->use_fused = h->use_fused16 ||
                 (cfg->kernel != VN_KERNEL_GENERIC && vn_fused_supported(net, cfg->integ_num));
  h->two_pass = !h->use_fused && tp_ok && (cfg->kernel == VN_KERNEL_FUSED16 || cfg->kernel == VN_KERNEL_AUTO);
  h->fused_only = deep_fused;
  { const char* fg = getenv("VN_FULL_GRID"); h->full_grid = fg && *fg && *fg != '0'; }
  { const char* pw = getenv("VN_PGRAD_WGS"); if (pw && *pw >= '1' && *pw <= '4') h->pgrad_wgs = *pw - '0'; }
  if (h->use_fused || h->two_pass) {
    // (the forward-only mode of the 8-wave kernel writes its per-workgroup loss partials here too: vn_forward and
    // vn_eval_loss of a two-pass engine must not find it NULL)
    if (hipMalloc((void**)&h->fused_losspart, (size_t)h->ncu * 3 * sizeof(float)) != hipSuccess) {
      vn_destroy(h);
      return fail(VN_ENOMEM, "device allocation failed");
    }
    h->prof_name = (h->use_fused16 || h->two_pass) ? "vn_fused16_kernel" : "vn_fused_kernel";
    if (h->use_fused16 || h->two_pass) {       // the instantiation that runs, as rocprofv3 prints it (template arguments)
      char nm[96];
      snprintf(nm, sizeof nm, "vn_fused16_kernel<%d, %d, %s>", net.L, vn_fused16_ks(net), net.act == VN_ACT_TANH ? "true" : "false");
      h->prof_name = nm;
    }
    if (hipMalloc((void**)&h->stamps, 8 * sizeof(unsigned long long)) == hipSuccess)
      (void)hipMemset(h->stamps, 0, 8 * sizeof(unsigned long long));
  }
  h->ev0.resize(PROF_CAP, nullptr);
  h->ev1.resize(PROF_CAP, nullptr);
  h->cev0.resize(PROF_CAP, nullptr);
  h->cev1.resize(PROF_CAP, nullptr);
  *out = h;
  return VN_OK;
}

int vn_destroy(vn_engine* h) {
  if (!h) return VN_OK;
  (void)hipSetDevice(h->cfg.device);
  if (h->layered) { (void)hipStreamSynchronize(h->stream); vn_layered_destroy(h->layered); h->layered = nullptr; }
  if (h->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(h->stream); (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  void* ptrs[] = {h->theta, h->m, h->v, h->snap, h->theta64, h->gradbuf_int, h->lossbuf, h->partial, h->feN, h->fedNt,
                  h->feW, h->u, h->ud, h->ubar, h->udbar, h->ub, h->ubar_b, h->losspart, h->fused_losspart, h->stamps, h->dd_uv, h->dd_ug, h->dd_su, h->dd_sg, h->dd_partial,
                  h->dd_losspart, h->tp_losspart};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (Batch& b : h->batches)
    if (b.gcsr) (void)hipFree(b.gcsr);
  for (auto e : h->ev0) if (e) (void)hipEventDestroy(e);
  for (auto e : h->ev1) if (e) (void)hipEventDestroy(e);
  for (auto e : h->cev0) if (e) (void)hipEventDestroy(e);
  for (auto e : h->cev1) if (e) (void)hipEventDestroy(e);
  delete h;
  return VN_OK;
}

int vn_set_stream(vn_engine* h, void* s) {
  if (!h) return fail(VN_EINVAL, "null handle");
  h->stream = (hipStream_t)s;
  return VN_OK;
}

int vn_param_count(const vn_engine* h, int64_t* n) {
  if (!h || !n) return fail(VN_EINVAL, "null argument");
  *n = h->net.P;
  return VN_OK;
}

int vn_params_init(vn_engine* h, uint64_t seed) {
  if (!h) return fail(VN_EINVAL, "null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  const VnNet& net = h->net;
  std::vector<float> t(net.P, 0.f);
  uint64_t s = seed;
  for (int l = 1; l <= net.L + 1; ++l) {
    const int fi = net.H[l - 1], fo = net.H[l];
    const double lim = std::sqrt(6.0 / (double)(fi + fo));          // keras glorot_uniform
    for (int i = 0; i < fi * fo; ++i) {
      const double u01 = (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
      t[net.woff[l] + i] = (float)((2.0 * u01 - 1.0) * lim);
    }
  }
  HIPCHK(hipMemcpyAsync(h->theta, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->m, 0, t.size() * sizeof(float), h->stream));
  if (h->cfg.optimizer == VN_OPT_RMSPROP) {          // TF-1 initialises the mean-square slot to ones
    std::vector<float> ones(net.P, 1.f);
    HIPCHK(hipMemcpyAsync(h->v, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  } else {
    HIPCHK(hipMemsetAsync(h->v, 0, t.size() * sizeof(float), h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->step = 0;
  return VN_OK;
}

int vn_params_get(vn_engine* h, float* host, int64_t n) {
  if (!h || !host) return fail(VN_EINVAL, "null argument");
  if (n != h->net.P) return fail(VN_EINVAL, "expected %d parameters, got %lld", h->net.P, (long long)n);
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipMemcpyAsync(host, h->theta, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VN_OK;
}

int vn_params_set(vn_engine* h, const float* host, int64_t n) {
  if (!h || !host) return fail(VN_EINVAL, "null argument");
  if (n != h->net.P) return fail(VN_EINVAL, "expected %d parameters, got %lld", h->net.P, (long long)n);
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipMemcpyAsync(h->theta, host, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VN_OK;
}

int vn_state_size(const vn_engine* h, int64_t* bytes) {
  if (!h || !bytes) return fail(VN_EINVAL, "null argument");
  *bytes = (int64_t)sizeof(int64_t) + 3ll * h->net.P * (int64_t)sizeof(float);
  return VN_OK;
}

int vn_state_export(vn_engine* h, void* host, int64_t bytes) {
  int64_t need = 0;
  if (!h || !host) return fail(VN_EINVAL, "null argument");
  vn_state_size(h, &need);
  if (bytes != need) return fail(VN_EINVAL, "state buffer must be %lld bytes", (long long)need);
  HIPCHK(hipSetDevice(h->cfg.device));
  char* p = (char*)host;
  memcpy(p, &h->step, sizeof(int64_t));
  p += sizeof(int64_t);
  const size_t nb = (size_t)h->net.P * sizeof(float);
  HIPCHK(hipMemcpyAsync(p, h->theta, nb, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(p + nb, h->m, nb, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(p + 2 * nb, h->v, nb, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VN_OK;
}

// Device-side snapshot of the optimizer state (parameters, both slots, step counter) and the way back to it: no host copy, no
// synchronisation -- everything is ordered on the engine stream.  What train()'s lossLag blocks stand on: a block of k epochs is
// enqueued before ONE read-back of its k losses; when the stopping test fires inside the block, the state is rolled back to the
// block's start and the epochs up to the one that met the tolerance are replayed (the steps are bitwise reproducible), so the
// run ends in exactly the state the reference's one-read-back-per-epoch loop ends in (VarNet.py:1346-1383).
int vn_state_snapshot(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t nb = (size_t)h->net.P * sizeof(float);
  if (!h->snap) HIPCHK(hipMalloc((void**)&h->snap, 3 * nb));
  HIPCHK(hipMemcpyAsync(h->snap, h->theta, nb, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->snap + h->net.P, h->m, nb, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->snap + 2 * (size_t)h->net.P, h->v, nb, hipMemcpyDeviceToDevice, h->stream));
  h->snap_step = h->step;
  return VN_OK;
}

int vn_state_rollback(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (!h->snap || h->snap_step < 0) return fail(VN_ESTATE, "no snapshot to roll back to (call vn_state_snapshot first)");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t nb = (size_t)h->net.P * sizeof(float);
  HIPCHK(hipMemcpyAsync(h->theta, h->snap, nb, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->m, h->snap + h->net.P, nb, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->v, h->snap + 2 * (size_t)h->net.P, nb, hipMemcpyDeviceToDevice, h->stream));
  h->step = h->snap_step;
  return VN_OK;
}

int vn_state_import(vn_engine* h, const void* host, int64_t bytes) {
  int64_t need = 0;
  if (!h || !host) return fail(VN_EINVAL, "null argument");
  vn_state_size(h, &need);
  if (bytes != need) return fail(VN_EINVAL, "state buffer must be %lld bytes", (long long)need);
  HIPCHK(hipSetDevice(h->cfg.device));
  const char* p = (const char*)host;
  memcpy(&h->step, p, sizeof(int64_t));
  p += sizeof(int64_t);
  const size_t nb = (size_t)h->net.P * sizeof(float);
  HIPCHK(hipMemcpyAsync(h->theta, p, nb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->m, p + nb, nb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->v, p + 2 * nb, nb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return VN_OK;
}

int vn_set_fe_table(vn_engine* h, const float* N, const float* dNt, const float* integW) {
  if (!h || !N || !dNt) return fail(VN_EINVAL, "null argument");
  if (h->cfg.has_integw && !integW) return fail(VN_EINVAL, "config has integW but none was given");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t nb = (size_t)h->cfg.integ_num * sizeof(float);
  HIPCHK(hipMemcpyAsync(h->feN, N, nb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->fedNt, dNt, nb, hipMemcpyHostToDevice, h->stream));
  if (integW) HIPCHK(hipMemcpyAsync(h->feW, integW, nb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));   // host buffers may be released on return
  h->has_fe = true;
  h->has_feW = integW != nullptr;
  return VN_OK;
}

int vn_set_interior(vn_engine* h, int32_t batch, const float* Input, const float* gcoef, const float* source,
                    int64_t n_k, const float* detJ_dev, double detJ, const float* N_rows, const float* dNt_rows) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (batch < 0 || batch > 65535) return fail(VN_EINVAL, "batch index %d out of range", batch);
  if (n_k < 0) return fail(VN_EINVAL, "negative number of test functions");
  // n_k == 0 is a legal, empty tower feed (VarNetUtility.py:830-838 slices past the end): only the BC/IC rows
  // contribute, and the rank still takes part in the gradient SUM.
  if (n_k > 0 && (!Input || !gcoef)) return fail(VN_EINVAL, "null argument");
  if (n_k > 0 && h->cfg.has_source && !source) return fail(VN_EINVAL, "config has a source term but source is NULL");
  if ((N_rows == nullptr) != (dNt_rows == nullptr)) return fail(VN_EINVAL, "N_rows and dNt_rows must be given together");
  HIPCHK(hipSetDevice(h->cfg.device));
  if ((int)h->batches.size() <= batch) h->batches.resize(batch + 1);
  Batch& b = h->batches[batch];
  b.Input = Input; b.gcoef = gcoef; b.source = source; b.detJv = detJ_dev; b.detJ = detJ;
  b.Nrow = N_rows; b.dNtrow = dNt_rows; b.n_k = n_k; b.set = true;
  b.Xu = nullptr; b.uid = nullptr; b.rowptr = nullptr; b.rowidx = nullptr; b.U = 0;   // re-register with vn_set_dedup
  b.biInput = nullptr; b.biLabel = nullptr;                                            // ... and vn_set_batch_bic
  const long nT = n_k * h->cfg.integ_num;
  if (nT > h->work_rows) {
    long c0 = h->work_rows, c1 = h->work_rows, c2 = h->work_rows, c3 = h->work_rows;
    if (int rc = ensure(&h->u, &c0, nT)) return rc;
    if (int rc = ensure(&h->ud, &c1, nT)) return rc;
    if (int rc = ensure(&h->ubar, &c2, nT)) return rc;
    if (int rc = ensure(&h->udbar, &c3, nT)) return rc;
    h->work_rows = nT;
  }
  return VN_OK;
}

int vn_set_dedup(vn_engine* h, int32_t batch, const float* Xu, int64_t U, const int32_t* uid,
                 const int32_t* rowptr, const int32_t* rowidx) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (batch < 0 || batch >= (int)h->batches.size() || !h->batches[batch].set)
    return fail(VN_ESTATE, "batch %d has no interior data (call vn_set_interior first)", batch);
  Batch& b = h->batches[batch];
  if (!Xu) {                                   // switch the formulation off for this batch
    b.Xu = nullptr; b.uid = nullptr; b.rowptr = nullptr; b.rowidx = nullptr; b.U = 0;
    return VN_OK;
  }
  // A registration replaces the previous one: from here on the batch is row-wise until THIS map has been accepted, so a
  // rejected call never leaves the engine pointing at the previous call's arrays (which the caller may release on error).
  b.Xu = nullptr; b.uid = nullptr; b.rowptr = nullptr; b.rowidx = nullptr; b.U = 0;
  if (b.n_k <= 0) return fail(VN_EINVAL, "batch %d has no interior rows: nothing to de-duplicate", batch);
  if (!uid || !rowptr || !rowidx || U <= 0) return fail(VN_EINVAL, "null argument");
  // the formulation has no tiles of whole test functions (its rows are unique points), so it also serves integ_num beyond one
  // 128-point tile -- the networks of the two-pass route (216: three-point Gauss in 2D+t) -- up to the seed kernel's 256-row chunk
  if (!h->use_fused16 && !h->two_pass) return fail(VN_EUNSUPPORTED, "de-duplication needs the 8-wave fused kernel for this network");
  if (h->cfg.integ_num > 256) return fail(VN_EUNSUPPORTED, "de-duplication supports integ_num <= 256");
  if (b.Nrow || b.detJv) return fail(VN_EUNSUPPORTED, "de-duplication needs uniform supports (no per-row tables)");
  if (h->cfg.dim > 3) return fail(VN_EUNSUPPORTED, "de-duplication supports dim <= 3");
  HIPCHK(hipSetDevice(h->cfg.device));
  const int dim = h->cfg.dim;
  if (U > h->dd_capU) {
    float** bufs[] = {&h->dd_uv, &h->dd_su, &h->dd_sg};          // dd_uv: [U, 4] packed (u, grad u) records
    const long sizes[] = {4 * U, U, U * dim};
    for (int i = 0; i < 3; ++i) {
      if (*bufs[i]) (void)hipFree(*bufs[i]);
      *bufs[i] = nullptr;
      HIPCHK(hipMalloc((void**)bufs[i], (size_t)sizes[i] * sizeof(float)));
    }
    h->dd_capU = U;
  }
  if (!h->dd_partial) HIPCHK(hipMalloc((void**)&h->dd_partial, (size_t)h->ncu * h->net.P * sizeof(float)));
  const long need_lp = ((long)h->ncu + (b.n_k + VN_DEDUP_TFB - 1) / VN_DEDUP_TFB) * 3;
  if (need_lp > h->dd_cap_lp) {
    if (h->dd_losspart) (void)hipFree(h->dd_losspart);
    h->dd_losspart = nullptr;
    HIPCHK(hipMalloc((void**)&h->dd_losspart, (size_t)need_lp * sizeof(float)));
    h->dd_cap_lp = need_lp;
  }
  // The map indexes device memory in every later kernel: validate it once, here (a registration call may synchronise), so
  // that an inconsistent map is an error code and never a GPU fault.  The rowptr reads assume U + 1 entries, rowidx / uid nT.
  {
    const long nTc = b.n_k * h->cfg.integ_num;
    int* err_dev = nullptr;
    HIPCHK(hipMalloc((void**)&err_dev, sizeof(int)));
    hipError_t e = hipMemsetAsync(err_dev, 0, sizeof(int), h->stream);
    if (e == hipSuccess) e = vn_dedup_check_launch(uid, rowptr, rowidx, nTc, U, err_dev, h->stream);
    int bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, err_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(err_dev);
    if (e != hipSuccess) return fail(VN_EHIP, "vn_set_dedup: %s", hipGetErrorString(e));
    if (bad) return fail(VN_EINVAL, "inconsistent de-duplication map: %d violation(s) (need 0 <= uid < U, rowptr[0] = 0 <= ... <= rowptr[U] = n_k*integ_num, "
                                    "0 <= rowidx < n_k*integ_num, uid[rowidx[e]] = the point whose segment holds e, rows of a point in increasing order)", bad);
  }
  // With constant coefficients gcoef = kappa dN/dx + v N repeats with period integ_num along the rows (the reference tiles the
  // tables to nT rows, VarNet.py:837): detected here, bitwise, and both assembly kernels then read the rows of test function 0
  // as an integ_num-entry table instead of 8 bytes per row each.  Otherwise the gather kernel gets gcoef in CSR order (it then
  // reads it, like rowidx, as one contiguous stream: a per-row gather of 8-byte entries fetched 2.6 x the bytes it used,
  // profiles/r5_pmc_traffic_dedup.json).  gcoef is static per batch: examined / permuted once, here.
  const long nT = b.n_k * h->cfg.integ_num;
  {
    int* err_dev = nullptr;
    HIPCHK(hipMalloc((void**)&err_dev, sizeof(int)));
    hipError_t e = hipMemsetAsync(err_dev, 0, sizeof(int), h->stream);
    if (e == hipSuccess) e = vn_dedup_periodic_launch(b.gcoef, nT, h->cfg.integ_num, dim, err_dev, h->stream);
    int bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, err_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(err_dev);
    if (e != hipSuccess) return fail(VN_EHIP, "vn_set_dedup: %s", hipGetErrorString(e));
    b.gper = bad == 0 && !h->no_gtable;
  }
  if (!b.gper) {
    if (nT * dim > b.gcsr_cap) {
      if (b.gcsr) (void)hipFree(b.gcsr);
      b.gcsr = nullptr; b.gcsr_cap = 0;
      HIPCHK(hipMalloc((void**)&b.gcsr, (size_t)nT * dim * sizeof(float)));
      b.gcsr_cap = nT * dim;
    }
    HIPCHK(vn_dedup_permute_launch(b.gcoef, rowidx, b.gcsr, nT, dim, h->stream));
  }
  b.Xu = Xu; b.U = U; b.uid = uid; b.rowptr = rowptr; b.rowidx = rowidx;
  return VN_OK;
}

int vn_set_bic(vn_engine* h, const float* biInput, const float* biLabel, int64_t nB, int64_t bDof, double biDimVal) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (nB < 0 || bDof < 0 || bDof > nB) return fail(VN_EINVAL, "need 0 <= bDof <= nB");
  if (nB > 0 && (!biInput || !biLabel)) return fail(VN_EINVAL, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  h->biInput = biInput; h->biLabel = biLabel; h->nB = nB; h->bDof = bDof; h->biDimVal = biDimVal;
  if (nB > h->work_b) {
    long c0 = h->work_b, c1 = h->work_b;
    if (int rc = ensure(&h->ub, &c0, nB)) return rc;
    if (int rc = ensure(&h->ubar_b, &c1, nB)) return rc;
    h->work_b = nB;
  }
  return VN_OK;
}

int vn_set_batch_bic(vn_engine* h, int32_t batch, const float* biInput, const float* biLabel) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (batch < 0 || batch >= (int)h->batches.size() || !h->batches[batch].set)
    return fail(VN_ESTATE, "batch %d has no interior data (call vn_set_interior first)", batch);
  if ((biInput == nullptr) != (biLabel == nullptr)) return fail(VN_EINVAL, "biInput and biLabel must be given together");
  h->batches[batch].biInput = biInput;
  h->batches[batch].biLabel = biLabel;
  return VN_OK;
}

int vn_set_weights(vn_engine* h, const double w[3]) {
  if (!h || !w) return fail(VN_EINVAL, "null argument");
  h->w[0] = w[0]; h->w[1] = w[1]; h->w[2] = w[2];
  return VN_OK;
}

int vn_bind_grad_buffer(vn_engine* h, float* dev) {
  if (!h) return fail(VN_EINVAL, "null handle");
  h->gradbuf = dev ? dev : h->gradbuf_int;
  return VN_OK;
}

int vn_grad(vn_engine* h, int32_t batch) {
  if (!h) return fail(VN_EINVAL, "null handle");
  (void)hipGetLastError();   // a stale last-error of another library on this thread is not ours to report
  if (int rc = check_batch(h, batch)) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  const Batch& b = h->batches[batch];
  if (b.Xu && (h->use_fused16 || h->two_pass)) return run_dedup(h, b, h->gradbuf);
  if (h->two_pass) return run_twopass(h, b, h->gradbuf);
  if (h->use_fused) {
    VnFusedArgs a{};
    a.net = h->net; a.theta = h->theta;
    a.X = b.Input; a.G = b.gcoef; a.src = h->cfg.has_source ? b.source : nullptr;
    a.nT = b.n_k * h->cfg.integ_num; a.n_k = b.n_k; a.integ_num = h->cfg.integ_num;
    a.feN = h->feN; a.fedNt = h->fedNt; a.feW = (h->cfg.has_integw && h->has_feW) ? h->feW : nullptr;
    a.Nrow = b.Nrow; a.dNtrow = b.dNtrow;
    a.detJv = b.detJv; a.detJ = (float)b.detJ; a.time_dependent = h->cfg.time_dependent;
    a.lossVec = nullptr;
    a.Xb = bi_x(h, b); a.label = bi_y(h, b); a.nB = h->nB; a.bDof = h->bDof; a.biDimVal = (float)h->biDimVal;
    a.w0 = (float)h->w[0]; a.w1 = (float)h->w[1]; a.w2 = (float)h->w[2];
    a.partial = h->partial; a.losspart = h->fused_losspart; a.stamps = h->stamps;
    // one persistent workgroup per CU, but never more workgroups than tiles (small mini-batches: idle workgroups would still
    // image the weights, flush and store an all-zero partial that the reduction then has to read)
    const long tt = 128 / a.integ_num > 0 ? 128 / a.integ_num : 1;
    const long tiles = (a.n_k + tt - 1) / tt + (a.nB + 127) / 128;
    const int grid = h->full_grid ? h->ncu : (int)(tiles < 1 ? 1 : tiles < h->ncu ? tiles : h->ncu);
    const bool rec = h->prof_on && h->prof_n < PROF_CAP;
    if (rec) {
      if (!h->ev0[h->prof_n]) { HIPCHK(hipEventCreate(&h->ev0[h->prof_n])); HIPCHK(hipEventCreate(&h->ev1[h->prof_n])); }
      HIPCHK(hipEventRecord(h->ev0[h->prof_n], h->stream));
    }
    if (h->use_fused16) HIPCHK(vn_fused16_launch(a, grid, h->stream));
    else HIPCHK(vn_fused_launch(a, grid, h->stream));
    if (rec) { HIPCHK(hipEventRecord(h->ev1[h->prof_n], h->stream)); h->prof_n++; }
    HIPCHK(vn_reduce_launch(h->partial, grid, h->net.P, h->fused_losspart, grid, h->bDof, h->nB, a.w0, a.w1, a.w2,
                            h->gradbuf, h->stream, h->fuse));
    return VN_OK;
  }
  if (int rc = run_forward_and_seed(h, b, true, nullptr, nullptr)) return rc;
  const long nT = b.n_k * h->cfg.integ_num;
  VnRows s0{}, s1{};
  s0.X = b.Input; s0.G = b.gcoef; s0.ubar = h->ubar; s0.udbar = h->udbar; s0.n = nT;
  s1.X = bi_x(h, b); s1.G = nullptr; s1.ubar = h->ubar_b; s1.udbar = nullptr; s1.n = h->nB;
  if (h->layered) {
    // one gradient vector (no per-workgroup partials): the GEMMs accumulate into it chunk by chunk
    HIPCHK(hipMemsetAsync(h->partial, 0, (size_t)h->net.P * sizeof(float), h->stream));
    const bool lrec = h->prof_on && h->prof_n < PROF_CAP;
    if (lrec) {
      if (!h->ev0[h->prof_n]) { HIPCHK(hipEventCreate(&h->ev0[h->prof_n])); HIPCHK(hipEventCreate(&h->ev1[h->prof_n])); }
      HIPCHK(hipEventRecord(h->ev0[h->prof_n], h->stream));
    }
    LAYCHK(vn_layered_backward(h->layered, h->theta, s0, h->partial, h->stream, lerr_, sizeof lerr_, 0));
    LAYCHK(vn_layered_backward(h->layered, h->theta, s1, h->partial, h->stream, lerr_, sizeof lerr_, 1));
    if (lrec) { HIPCHK(hipEventRecord(h->ev1[h->prof_n], h->stream)); h->prof_n++; }
    const long nth = b.n_k > h->nB ? b.n_k : h->nB;
    const int lg = (int)(((nth > 0 ? nth : 1) + 255) / 256);
    HIPCHK(vn_reduce_launch(h->partial, 1, h->net.P, h->losspart, lg, h->bDof, h->nB, (float)h->w[0], (float)h->w[1],
                            (float)h->w[2], h->gradbuf, h->stream, h->fuse));
    return VN_OK;
  }
  const bool rec = h->prof_on && h->prof_n < PROF_CAP;
  if (rec) {
    if (!h->ev0[h->prof_n]) { HIPCHK(hipEventCreate(&h->ev0[h->prof_n])); HIPCHK(hipEventCreate(&h->ev1[h->prof_n])); }
    HIPCHK(hipEventRecord(h->ev0[h->prof_n], h->stream));
  }
  HIPCHK(vn_generic_backward(h->net, h->theta, s0, s1, h->partial, h->bwd_grid, h->stream));
  if (rec) { HIPCHK(hipEventRecord(h->ev1[h->prof_n], h->stream)); h->prof_n++; }
  const long nthreads = b.n_k > h->nB ? b.n_k : h->nB;
  const int lgrid = (int)((nthreads + 255) / 256);
  HIPCHK(vn_reduce_launch(h->partial, h->bwd_grid, h->net.P, h->losspart, lgrid, h->bDof, h->nB, (float)h->w[0],
                          (float)h->w[1], (float)h->w[2], h->gradbuf, h->stream, h->fuse));
  return VN_OK;
}

static int apply_impl(vn_engine* h, float* loss_acc) {
  (void)hipGetLastError();   // a stale last-error of another library on this thread is not ours to report
  HIPCHK(hipSetDevice(h->cfg.device));
  // the step counter moves only when the update was launched (a failed call leaves the optimizer state as it found it)
  if (h->cfg.optimizer == VN_OPT_RMSPROP) {
    HIPCHK(vn_rmsprop_launch(h->theta, h->m, h->v, h->gradbuf, h->net.P, (float)h->cfg.lr, 0.9f, 0.0f, 1e-10f, loss_acc,
                             h->stream));
    h->step += 1;
    return VN_OK;
  }
  const double t = (double)(h->step + 1);
  const double lr_t = h->cfg.lr * std::sqrt(1.0 - std::pow(h->cfg.beta2, t)) / (1.0 - std::pow(h->cfg.beta1, t));
  HIPCHK(vn_adam_launch(h->theta, h->m, h->v, h->gradbuf, h->net.P, (float)lr_t, (float)h->cfg.beta1,
                        (float)h->cfg.beta2, (float)h->cfg.eps, loss_acc, h->stream));
  h->step += 1;
  return VN_OK;
}

int vn_apply(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  return apply_impl(h, nullptr);
}

// gradient + optimizer step with the update folded into the gradient reduction (no collective in between)
static int step_fused(vn_engine* h, int32_t batch, float* loss_acc) {
  if (h->comm) {
    // towers: gradient -> SUM over ranks -> update, all on the engine stream, no host round trip.
    // One-rank failure (include/varnet_hip.h, "Failure under a communicator"): this rank returns its error WITHOUT entering
    // the collective and with its optimizer state untouched; its peers' ncclAllReduce for this step never completes, so the
    // caller must end the job -- varnet_amd/launch.py ends the peers of a rank that exits non-zero by PID, within its poll
    // interval, and reports the failing rank's last stage (VNEngine._ck leaves `engine_error: ...` there).
    if (int rc = vn_grad(h, batch)) return rc;
    if (int rc = vn_allreduce_grad(h)) return rc;
    return apply_impl(h, loss_acc);
  }
  h->step += 1;
  VnOptArgs o;
  o.kind = h->cfg.optimizer; o.theta = h->theta; o.m = h->m; o.v = h->v; o.loss_acc = loss_acc;
  if (h->cfg.optimizer == VN_OPT_RMSPROP) {
    o.lr = (float)h->cfg.lr; o.b1 = 0.9f; o.b2 = 0.0f; o.eps = 1e-10f;          // decay, momentum, epsilon
  } else {
    const double t = (double)h->step;
    o.lr = (float)(h->cfg.lr * std::sqrt(1.0 - std::pow(h->cfg.beta2, t)) / (1.0 - std::pow(h->cfg.beta1, t)));
    o.b1 = (float)h->cfg.beta1; o.b2 = (float)h->cfg.beta2; o.eps = (float)h->cfg.eps;
  }
  h->fuse = o;
  const int rc = vn_grad(h, batch);
  h->fuse = VnOptArgs();
  if (rc) h->step -= 1;
  return rc;
}

int vn_train_epoch(vn_engine* h, const int32_t* batches, int32_t n, float* loss_acc_dev) {
  if (!h || (n > 0 && !batches)) return fail(VN_EINVAL, "null argument");
  for (int32_t i = 0; i < n; ++i)
    if (int rc = step_fused(h, batches[i], loss_acc_dev)) return rc;
  return VN_OK;
}

int vn_train_step(vn_engine* h, int32_t batch, float* loss_out_dev) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (int rc = step_fused(h, batch, nullptr)) return rc;
  if (loss_out_dev)
    HIPCHK(hipMemcpyAsync(loss_out_dev, h->gradbuf + h->net.P, sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  return VN_OK;
}

int vn_eval_loss(vn_engine* h, int32_t batch, double out[4], float* lossVec_dev) {
  if (!h || !out) return fail(VN_EINVAL, "null argument");
  (void)hipGetLastError();   // a stale last-error of another library on this thread is not ours to report
  if (int rc = check_batch(h, batch)) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc = run_forward_and_seed(h, h->batches[batch], false, lossVec_dev, h->lossbuf)) return rc;
  float t[4];
  HIPCHK(hipMemcpyAsync(t, h->lossbuf, sizeof t, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 4; ++i) out[i] = (double)t[i];
  return VN_OK;
}

int vn_forward(vn_engine* h, const float* X, int64_t n, float* u) {
  if (!h || (n > 0 && (!X || !u))) return fail(VN_EINVAL, "null argument");
  (void)hipGetLastError();   // a stale last-error of another library on this thread is not ours to report
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->layered) {
    VnRows sl{};
    sl.X = X; sl.G = nullptr; sl.u = u; sl.ud = nullptr; sl.n = n;
    LAYCHK(vn_layered_forward(h->layered, h->theta, sl, h->stream, lerr_, sizeof lerr_));
    return VN_OK;
  }
  // networks of the 8-wave family: the value-only sweep of vn_pgrad16 (F_pt per point; the fused kernel's forward-only mode
  // would carry a tangent stream of zeros through every layer)
  if (h->use_fused16 || h->two_pass) {
    // hidden widths 33..64: the products as six bf16-piece MFMAs, fp32-class (vn_split16.hip)
    if (!h->no_split && vn_split16_supported(h->net)) HIPCHK(vn_split16_forward(h->net, h->theta, X, n, u, h->ncu, h->stream));
    else HIPCHK(vn_pgrad16_launch(h->net, h->theta, X, n, u, nullptr, nullptr, h->ncu, h->pgrad_wgs, h->stream));
    return VN_OK;
  }
  if (h->fused_only) return fused_forward(h, X, nullptr, n, u, nullptr);
  VnRows s0{}, s1{};
  s0.X = X; s0.G = nullptr; s0.u = u; s0.ud = nullptr; s0.n = n;
  HIPCHK(vn_generic_forward(h->net, h->theta, s0, s1, h->fwd_grid, h->stream));
  return VN_OK;
}

int vn_forward_grad(vn_engine* h, const float* X, int64_t n, float* u, float* g) {
  if (!h || (n > 0 && (!X || !u || !g))) return fail(VN_EINVAL, "null argument");
  (void)hipGetLastError();   // a stale last-error of another library on this thread is not ours to report
  if (!h->use_fused16 && !h->two_pass) return fail(VN_EUNSUPPORTED, "vn_forward_grad needs a network of the 8-wave fused kernel");
  if (h->cfg.dim > 3) return fail(VN_EUNSUPPORTED, "vn_forward_grad supports dim <= 3");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->no_split && vn_split16_supported(h->net)) HIPCHK(vn_split16_pgrad(h->net, h->theta, X, n, u, g, nullptr, h->ncu, h->stream));
  else HIPCHK(vn_pgrad16_launch(h->net, h->theta, X, n, u, g, nullptr, h->ncu, h->pgrad_wgs, h->stream));
  return VN_OK;
}

int vn_forward_f64(vn_engine* h, const double* X, int64_t n, double* u) {
  if (!h || (n > 0 && (!X || !u))) return fail(VN_EINVAL, "null argument");
  (void)hipGetLastError();   // a stale last-error of another library on this thread is not ours to report
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc = refresh_theta64(h)) return rc;
  if (h->layered) {
    LAYCHK(vn_layered_forward_f64(h->layered, h->theta64, X, n, u, h->stream, lerr_, sizeof lerr_));
    return VN_OK;
  }
  // networks of the 8-wave family whose fp64 images fit the LDS: the fp64 matrix pipe (vn_taylor16d.hip); else per thread
  if ((h->use_fused16 || h->two_pass) && vn_taylor16d_supported(h->net) && !h->point_kernels) {
    HIPCHK(vn_taylor16d_launch(h->net, h->theta64, X, nullptr, nullptr, nullptr, nullptr, h->cfg.time_dependent, n, u, nullptr, h->ncu, h->stream));
    return VN_OK;
  }
  HIPCHK(vn_pointwise_forward_f64(h->net, h->theta64, X, n, u, h->stream));
  return VN_OK;
}

int vn_residual(vn_engine* h, const float* X, const float* diff, const float* vel, const float* src,
                const float* ddx, int64_t n, float* u, float* res) {
  if (!h || (n > 0 && (!X || !diff || !vel || !res))) return fail(VN_EINVAL, "null argument");
  if (h->cfg.dim > 3) return fail(VN_EUNSUPPORTED, "residual supports dim <= 3");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->layered) {
    LAYCHK(vn_layered_residual_f32(h->layered, h->theta, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res, h->stream,
                                   lerr_, sizeof lerr_));
    return VN_OK;
  }
  // networks of the 8-wave family: second-order forward mode on the matrix pipe (vn_taylor16.hip); the per-point kernel keeps
  // the generic / 4-wave requests (and is what the new kernel is cross-checked against)
  if ((h->use_fused16 || h->two_pass) && vn_taylor16_supported(h->net, h->cfg.time_dependent) && !h->point_kernels) {
    if (!h->no_split && vn_split16_supported(h->net))
      HIPCHK(vn_split16_residual(h->net, h->theta, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res, h->ncu, h->stream));
    else
      HIPCHK(vn_taylor16_residual(h->net, h->theta, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res, h->ncu, h->stream));
    return VN_OK;
  }
  HIPCHK(vn_pointwise_residual_f32(h->net, h->theta, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res,
                                   h->stream));
  return VN_OK;
}

int vn_residual_f64(vn_engine* h, const double* X, const double* diff, const double* vel, const double* src,
                    const double* ddx, int64_t n, double* u, double* res) {
  if (!h || (n > 0 && (!X || !diff || !vel || !res))) return fail(VN_EINVAL, "null argument");
  if (h->cfg.dim > 3) return fail(VN_EUNSUPPORTED, "residual supports dim <= 3");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (int rc = refresh_theta64(h)) return rc;
  if (h->layered) {
    LAYCHK(vn_layered_residual_f64(h->layered, h->theta64, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res, h->stream,
                                   lerr_, sizeof lerr_));
    return VN_OK;
  }
  if ((h->use_fused16 || h->two_pass) && vn_taylor16d_supported(h->net) && !h->point_kernels) {
    HIPCHK(vn_taylor16d_launch(h->net, h->theta64, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res, h->ncu, h->stream));
    return VN_OK;
  }
  HIPCHK(vn_pointwise_residual_f64(h->net, h->theta64, X, diff, vel, src, ddx, h->cfg.time_dependent, n, u, res,
                                   h->stream));
  return VN_OK;
}


// ---- tower gradient SUM over RCCL (TFModel.py:342-377) ------------------------------------
int vn_comm_available(void) { return load_rccl(); }

int vn_comm_version(int32_t* version_out) {
  if (!version_out) return fail(VN_EINVAL, "null argument");
  if (int rc = load_rccl()) return rc;
  int v = 0;
  RCCLCHK(g_rccl.GetVersion(&v));
  *version_out = v;
  return VN_OK;
}

int vn_comm_unique_id(void* id_out) {
  if (!id_out) return fail(VN_EINVAL, "null argument");
  if (int rc = load_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == VN_COMM_ID_BYTES, "RCCL unique id size");
  ncclUniqueId id;
  RCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return VN_OK;
}

int vn_comm_init(vn_engine* h, int32_t rank, int32_t world, const void* unique_id) {
  if (!h || !unique_id) return fail(VN_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(VN_EINVAL, "need 0 <= rank < world (got %d of %d)", rank, world);
  {
    std::lock_guard<std::mutex> lk(h->comm_mu);
    if (h->comm_abandoned) return fail(VN_ESTATE, "this engine's communicator was abandoned (vn_comm_abandon): it takes no other");
    if (h->comm) return fail(VN_ESTATE, "communicator already initialised (call vn_comm_destroy first)");
  }
  if (int rc = load_rccl()) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  ncclComm_t c = nullptr;
  RCCLCHK(g_rccl.CommInitRank(&c, world, id, rank));
  int cnt = 0;
  RCCLCHK(g_rccl.CommCount(c, &cnt));
  if (cnt != world) { (void)g_rccl.CommDestroy(c); return fail(VN_ECOMM, "RCCL reports %d ranks, expected %d", cnt, world); }
  std::lock_guard<std::mutex> lk(h->comm_mu);
  if (h->comm_abandoned) {       // the caller gave up on this call while it sat in ncclCommInitRank: the engine must not start using it
    h->comm_orphan = c;
    return fail(VN_ESTATE, "ncclCommInitRank returned after the communicator was abandoned (vn_comm_abandon)");
  }
  h->comm = c; h->comm_world = world; h->comm_rank = rank;
  return VN_OK;
}

int vn_comm_abandon(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  std::lock_guard<std::mutex> lk(h->comm_mu);
  h->comm_abandoned = true;
  if (h->comm) { h->comm_orphan = h->comm; h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0; }
  return VN_OK;
}

int vn_comm_size(const vn_engine* h, int32_t* world, int32_t* rank) {
  if (!h || !world) return fail(VN_EINVAL, "null argument");
  int cnt = 1;
  if (h->comm) RCCLCHK(g_rccl.CommCount(h->comm, &cnt));
  *world = cnt;
  if (rank) *rank = h->comm ? h->comm_rank : 0;
  return VN_OK;
}

int vn_comm_destroy(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (!h->comm) return VN_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  ncclComm_t c = h->comm;
  h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0;
  RCCLCHK(g_rccl.CommDestroy(c));
  return VN_OK;
}

int vn_allreduce_grad(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  if (!h->comm) return fail(VN_ESTATE, "no communicator (call vn_comm_init first)");
  HIPCHK(hipSetDevice(h->cfg.device));
  // one collective per step: P gradient floats + (loss, BC, IC, var), in place, on the engine stream
  const bool rec = h->prof_on && h->cprof_n < PROF_CAP;
  if (rec) {
    if (!h->cev0[h->cprof_n]) { HIPCHK(hipEventCreate(&h->cev0[h->cprof_n])); HIPCHK(hipEventCreate(&h->cev1[h->cprof_n])); }
    HIPCHK(hipEventRecord(h->cev0[h->cprof_n], h->stream));
  }
  RCCLCHK(g_rccl.AllReduce(h->gradbuf, h->gradbuf, (size_t)h->net.P + 4, ncclFloat32, ncclSum, h->comm, h->stream));
  if (rec) { HIPCHK(hipEventRecord(h->cev1[h->cprof_n], h->stream)); h->cprof_n++; }
  return VN_OK;
}

int vn_get_step(const vn_engine* h, int64_t* step) {
  if (!h || !step) return fail(VN_EINVAL, "null argument");
  *step = h->step;
  return VN_OK;
}

int vn_profile_comm(vn_engine* h, double* mean_ms, int64_t* calls) {
  if (!h) return fail(VN_EINVAL, "null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (int i = 0; i < h->cprof_n; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->cev0[i], h->cev1[i]));
    tot += ms;
  }
  if (mean_ms) *mean_ms = h->cprof_n ? tot / h->cprof_n : 0.0;
  if (calls) *calls = h->cprof_n;
  return VN_OK;
}

int vn_kernel_path(const vn_engine* h, int32_t* kernel, int32_t* two_pass) {
  if (!h || !kernel) return fail(VN_EINVAL, "null argument");
  *kernel = h->layered ? VN_KERNEL_LAYERED : h->use_fused16 ? VN_KERNEL_FUSED16 : h->use_fused ? VN_KERNEL_FUSED : h->two_pass ? VN_KERNEL_FUSED16 : VN_KERNEL_GENERIC;
  if (two_pass) *two_pass = h->two_pass ? 1 : 0;
  return VN_OK;
}

int vn_debug_calibrate(vn_engine* h, double out[5]) {
  if (!h || !out) return fail(VN_EINVAL, "null argument");
  (void)hipGetLastError();
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(vn_calibrate(h->ncu, h->stream, out));
  return VN_OK;
}

int vn_debug_calibrate_f64(vn_engine* h, double ghz, double out[3]) {
  if (!h || !out) return fail(VN_EINVAL, "null argument");
  (void)hipGetLastError();
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(vn_calibrate_f64(h->ncu, h->stream, ghz, out));
  return VN_OK;
}

int vn_debug_point_route(vn_engine* h, int32_t per_thread) {
  if (!h) return fail(VN_EINVAL, "null handle");
  h->no_gtable = (per_thread & 4) != 0;
  h->eval_rowwise = (per_thread & 8) != 0;
  per_thread &= 3;
  if (per_thread == 2 && !kWithF32Point)
    return fail(VN_EUNSUPPORTED, "route 2 (f32-MFMA point kernels for the networks the bf16-piece kernels serve) exists in the tests' "
                                 "cross-check build only: libvarnet_hip_xcheck.so (make -C varnet_amd/csrc xcheck)");
  h->point_kernels = per_thread == 1;
  h->no_split = per_thread == 2;
  return VN_OK;
}

int vn_debug_stamps(vn_engine* h, unsigned long long out[8]) {
  if (!h || !out) return fail(VN_EINVAL, "null argument");
  if (!h->stamps) { memset(out, 0, 8 * sizeof(unsigned long long)); return VN_OK; }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(out, h->stamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return VN_OK;
}

int vn_profile_begin(vn_engine* h) {
  if (!h) return fail(VN_EINVAL, "null handle");
  h->prof_on = true;
  h->prof_n = 0;
  h->cprof_n = 0;
  return VN_OK;
}

int vn_profile_end(vn_engine* h, double* mean_ms, int64_t* launches, char* name, int32_t name_len) {
  if (!h) return fail(VN_EINVAL, "null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (int i = 0; i < h->prof_n; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]));
    tot += ms;
  }
  if (mean_ms) *mean_ms = h->prof_n ? tot / h->prof_n : 0.0;
  if (launches) *launches = h->prof_n;
  if (name && name_len > 0) {
    strncpy(name, h->prof_name.c_str(), name_len - 1);
    name[name_len - 1] = 0;
  }
  h->prof_on = false;
  return VN_OK;
}

}  // extern "C"
