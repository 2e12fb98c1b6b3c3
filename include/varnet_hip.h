/*
 * varnet_hip.h -- C ABI of libvarnet_hip.so, the MI355X (gfx950) engine behind the VarNet
 * variational-loss training loop.
 *
 * The reference has no FFI layer: its device boundary is the `TFNN` object that
 * `VarNet` / `ManageTrainData` drive through `sess.run(feed_dict)`.  Every entry point below
 * replaces one of those call sites (cited as /root/reference/<file>:<line>).  All functions
 * are `extern "C"`, take plain pointers and sizes, return 0 on success and a non-zero
 * VN_E* code on failure (message via vn_last_error()).
 *
 * Conventions
 *   - "dev" pointers are device (HBM) addresses owned by the caller; they must stay valid
 *     while registered.  "host" pointers are ordinary host memory.
 *   - Rows of the interior arrays are grouped by test function: row r = k*integ_num + p
 *     (test function k, quadrature point p), exactly the reference's `Input` layout
 *     (VarNet.py:576-588, VarNetUtility.py:820).
 *   - Flat parameter order: W_1[d_in,H_1] row-major, b_1[H_1], ..., w_o[H_L,1], b_o[1]
 *     (Keras kernel is [in,out]; TFModel.py:208-242).
 *   - All work is enqueued on the stream given to vn_set_stream (default: the null stream);
 *     no entry point synchronises with the host unless its comment says so.
 *   - A handle is not thread-safe; use one handle per GPU / per process.
 */
#ifndef VARNET_HIP_H
#define VARNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* What a vn_config may describe (the reference takes any layerWidth list, TFModel.py:208-221).  The hand-written
 * kernels cover the VN_K* range; anything beyond it runs on the layer-by-layer route (VN_KERNEL_LAYERED). */
#define VN_MAX_LAYERS 16    /* hidden layers                                   */
#define VN_MAX_WIDTH  2048  /* hidden width                                    */
#define VN_MAX_DIN    32    /* network inputs: dim + time + MOR parameters     */
#define VN_KMAX_LAYERS 6    /* ... covered by the fused / generic kernels      */
#define VN_KMAX_WIDTH  64
#define VN_KMAX_DIN    8

enum {
  VN_OK = 0,
  VN_EINVAL = 1,   /* bad argument                                             */
  VN_EHIP = 2,     /* a HIP runtime call failed                                */
  VN_ESTATE = 3,   /* call order violated (e.g. step before data registered)   */
  VN_ENOMEM = 4,
  VN_EUNSUPPORTED = 5,
  VN_ECOMM = 6     /* an RCCL call failed                                      */
};

/* activationFun options of the constructor (VarNet.py:97).  VN_ACT_PER_LAYER: vn_config.layer_act holds one of the two
 * per hidden layer (the reference accepts a list, TFModel.py:113-119); a list with different entries runs on the
 * layer-by-layer route, the kernels take one activation for all hidden layers. */
enum { VN_ACT_SIGMOID = 0, VN_ACT_TANH = 1, VN_ACT_PER_LAYER = 2 };
enum { VN_OPT_ADAM = 0, VN_OPT_RMSPROP = 1 };   /* tf.train.AdamOptimizer / RMSPropOptimizer (TFModel.py:183-186) */
/* Kernel families.  AUTO picks the 8-wave fused kernel where it is instantiated: uniform or ragged hidden widths
 * <= 50 with 1..8 layers, <= 64 with 1..6 layers, d_in <= 8, sigmoid or tanh; integ_num <= 128 in one launch, larger
 * through the two-pass route.  The generic kernels serve VN_KERNEL_GENERIC requests (the independent cross-check of the
 * tests); networks beyond VN_KMAX_* run on the layer-by-layer route. */
enum { VN_KERNEL_AUTO = 0, VN_KERNEL_GENERIC = 1, VN_KERNEL_FUSED = 2 /* 4 waves, 32x32x2 */,
       VN_KERNEL_FUSED16 = 3 /* 8 waves, 16x16x4 */,
       /* Layer-by-layer route for networks outside the kernels' range (more than 6 hidden layers, widths above 64,
        * more than 8 inputs).  Hidden widths <= 256: tile
        * kernels that carry 32 points through all layers with the activations in LDS and keep (a, ad) of every layer
        * in HBM for the reverse kernel (6 F_pt per point).  Otherwise: activations of a chunk of rows live in HBM,
        * every layer is one GEMM over the stacked (value, tangent) rows -- the hand-written MFMA products of vn_gemm.hip --
        * plus hand-written elementwise kernels (6 F_pt when a step's activations fit in HBM, else 8).  Same results
        * contract as the other routes. */
       VN_KERNEL_LAYERED = 4 };

typedef struct vn_engine vn_engine;

/* Replaces the arguments of TFNN.__init__ (TFModel.py:85-191) + lossOpt (VarNet.py:182-185). */
typedef struct vn_config {
  int32_t dim;                      /* spatial dimension                                   */
  int32_t d_in;                     /* network inputs (VarNet.py:174-180)                  */
  int32_t n_layers;                 /* hidden layers L                                     */
  int32_t widths[VN_MAX_LAYERS];    /* layerWidth                                          */
  int32_t activation;               /* VN_ACT_SIGMOID | VN_ACT_TANH for all hidden layers, or VN_ACT_PER_LAYER */
  int32_t integ_num;                /* quadrature points per test function (FiniteElement.py:416) */
  int32_t time_dependent;           /* TFModel.py:537,646,655                              */
  int32_t has_source;               /* lossOpt['isSource']  (TFModel.py:656)               */
  int32_t has_integw;               /* lossOpt['integWflag'] (TFModel.py:660)              */
  int32_t device;                   /* HIP device ordinal                                  */
  int32_t optimizer;                /* VN_OPT_ADAM | VN_OPT_RMSPROP (TFModel.py:183-186)   */
  int32_t kernel;                   /* VN_KERNEL_*                                         */
  double  lr, beta1, beta2, eps;    /* taken literally, NO defaulting (TF-1's own defaults are 1e-3, .9, .999, 1e-8).  lr >= 0
                                     * (lr = 0 is legal TF: TFModel.py:130).  With VN_OPT_ADAM: 0 <= beta1, beta2 < 1 and
                                     * eps > 0, else vn_create returns VN_EINVAL -- a zero-initialised struct (eps = 0 turns a
                                     * zero gradient into 0/0 = NaN in the update) is rejected, not trained.  RMSProp ignores
                                     * beta1, beta2, eps (TF-1 constants: decay 0.9, momentum 0, epsilon 1e-10).           */
  int32_t layer_act[VN_MAX_LAYERS]; /* with VN_ACT_PER_LAYER: VN_ACT_SIGMOID | VN_ACT_TANH of hidden layer i */
} vn_config;

const char* vn_last_error(void);
int  vn_abi_version(void);   /* 2: towers (vn_comm_*), tanh, empty feeds, vn_kernel_path, vn_profile_comm;
                                * 3: vn_config.widths holds VN_MAX_LAYERS = 16 entries, layer_act, VN_KERNEL_LAYERED;
                                * 4: vn_comm_available, Adam hyper-parameters validated (no silent NaN from a zeroed config);
                                * 5: vn_comm_version;
                                * 6: vn_forward_grad, vn_debug_calibrate
                                * 7: vn_comm_abandon, vn_debug_point_route, vn_debug_calibrate_f64, vn_state_snapshot / vn_state_rollback */
#define VN_ABI_VERSION 7     /* what this header describes: a binding must refuse a library that reports another number */

/* TFNN.__init__ / graph + session construction (TFModel.py:85-191, 293-338). */
int vn_create(const vn_config* cfg, vn_engine** out);
int vn_destroy(vn_engine* h);
/* Stream all later work is enqueued on (pass torch.cuda.current_stream().cuda_stream). */
int vn_set_stream(vn_engine* h, void* hip_stream);

/* Number of trainable scalars P. */
int vn_param_count(const vn_engine* h, int64_t* n);
/* sess.run(global_variables_initializer()) (TFModel.py:326, VarNet.py:1412): glorot-uniform
 * kernels, zero biases, zero Adam slots, step 0.  Deterministic in `seed`. */
int vn_params_init(vn_engine* h, uint64_t seed);
/* saver.save / restore of the trainable variables, saveNNparam (VarNet.py:1362,1498,2231).
 * Host buffers of P floats.  These two synchronise the stream. */
int vn_params_get(vn_engine* h, float* host, int64_t n);
int vn_params_set(vn_engine* h, const float* host, int64_t n);
/* Full optimiser state {theta, m, v, step}: what tf.train.Saver writes (TFModel.py:307).
 * Layout: int64 step, then 3*P floats.  Synchronises. */
int vn_state_size(const vn_engine* h, int64_t* bytes);
int vn_state_export(vn_engine* h, void* host, int64_t bytes);
int vn_state_import(vn_engine* h, const void* host, int64_t bytes);
/* Device-side snapshot of the same state (parameters, both optimizer slots, step counter) in the engine's one snapshot slot, and
 * the way back to it; on the engine stream, no host copy, no synchronisation (ABI 7).  VarNet.train stands its `lossLag` blocks on
 * it: k epochs are enqueued before ONE read-back of their losses (the reference reads one loss per epoch, VarNetUtility.py:1044);
 * when the stopping test `loss < tol` (VarNet.py:1378) fires inside a block, the state is rolled back to the block's start and the
 * epochs up to the one that met the tolerance are replayed -- the steps are bitwise reproducible -- so the run ends in exactly the
 * state the one-read-back-per-epoch loop ends in.  vn_state_rollback without a snapshot is VN_ESTATE. */
int vn_state_snapshot(vn_engine* h);
int vn_state_rollback(vn_engine* h);

/* Feed of tower.N / tower.dNt / tower.integW (VarNetUtility.py:845-852) for the uniform case,
 * where those nT-row arrays are period-integ_num tables (FiniteElement.py:426-432).
 * Host pointers, integ_num floats each; integW may be NULL (all ones). */
int vn_set_fe_table(vn_engine* h, const float* N, const float* dNt, const float* integW);

/* Feed of tower.Input / gcoef / source / intShape / detJ for one (mini-|MOR-)batch
 * (VarNetUtility.py:840-854).  Device pointers: Input [n_k*integ_num, d_in], gcoef
 * [n_k*integ_num, dim], source [n_k*integ_num] or NULL.  detJ_dev: per-test-function
 * determinants [n_k] (the reference's detJvec=True case) or NULL to use the scalar `detJ`.
 * N_rows/dNt_rows: per-row basis arrays [n_k*integ_num] (non-uniform supports) or NULL to use
 * the table of vn_set_fe_table.  n_k == 0 registers an empty tower feed (the reference slices past the end
 * of the set when batchLen*towers > nt, VarNetUtility.py:830-838): the pointers may then be NULL, only the
 * BC/IC rows contribute and the rank still joins the gradient SUM. */
int vn_set_interior(vn_engine* h, int32_t batch, const float* Input_dev, const float* gcoef_dev,
                    const float* source_dev, int64_t n_k, const float* detJ_dev, double detJ,
                    const float* N_rows_dev, const float* dNt_rows_dev);
/* OPTIONAL, no reference counterpart: de-duplicated formulation for `batch`.  On uniform grids every
 * quadrature point is shared by the 2^feDim hat functions around it, so the reference evaluates the
 * network 2^feDim times per point (VarNet.py:576-588).  Given the unique points Xu [U, d_in], the map
 * uid [n_k*integ_num] row -> unique point and its CSR inverse (rowptr [U+1], rowidx [n_k*integ_num]),
 * vn_grad evaluates value and input gradient once per unique point and assembles the same loss and
 * gradient (same math, different rounding).  All device pointers.  Xu == NULL switches it off.
 * The map is validated on the device at this call (which therefore synchronises): an inconsistent one returns VN_EINVAL
 * (ranges, rowptr a partition of [0, n_k*integ_num), uid[rowidx[e]] = the point whose segment holds e, the rows of a point in
 * increasing order -- hence rowidx a permutation).  The array LENGTHS are the caller's contract (the ABI carries pointers only).
 * A call replaces the batch's previous registration even when it fails: after an error the batch is row-wise.
 * A batch without interior rows (n_k == 0) is VN_EINVAL.
 * Requires a network of the 8-wave fused kernel (integ_num <= 256: the two-pass route's 216 included) and uniform supports.
 * The batch's gcoef is READ at this call (the engine keeps a copy
 * in CSR order for its seed gather): register again after changing gcoef in place; vn_set_interior clears the registration. */
int vn_set_dedup(vn_engine* h, int32_t batch, const float* Xu_dev, int64_t U, const int32_t* uid_dev,
                 const int32_t* rowptr_dev, const int32_t* rowidx_dev);

/* Feed of tower.biInput / biLabel / bDof / biDimVal (VarNetUtility.py:841-849).
 * biInput [nB, d_in], biLabel [nB]; rows [0,bDof) are boundary, [bDof,nB) initial condition. */
int vn_set_bic(vn_engine* h, const float* biInput_dev, const float* biLabel_dev, int64_t nB,
               int64_t bDof, double biDimVal);
/* Optional per-batch copy of the BC/IC rows: the reference's shuffleTrainData feeds every (mini-batch, tower) its own
 * permutation of biInput / biLabel (VarNetUtility.py:988-996).  Same nB, bDof, biDimVal as vn_set_bic; NULL, NULL (or a
 * new vn_set_interior for the batch) returns to the shared set. */
int vn_set_batch_bic(vn_engine* h, int32_t batch, const float* biInput_dev, const float* biLabel_dev);
/* updateDictFields('trainW') (VarNetUtility.py:921-922); the caller applies the
 * w[0:2] /= batchNum*puNum rule (VarNetUtility.py:900-901). */
int vn_set_weights(vn_engine* h, const double w[3]);

/* Optional externally owned gradient buffer of P+4 floats (gradient | loss, BC, IC, var) so
 * that the host can all-reduce it (tower gradient SUM, TFModel.py:342-377) between vn_grad
 * and vn_apply.  NULL restores the internal buffer. */
int vn_bind_grad_buffer(vn_engine* h, float* dev);

/* compute_gradients(loss) (TFModel.py:709): forward, weak-form loss, backward for `batch`;
 * leaves d loss/d theta and the 4 loss scalars in the gradient buffer. */
int vn_grad(vn_engine* h, int32_t batch);
/* optimizer.apply_gradients (TFModel.py:313): TF-1 Adam (or RMSProp: decay 0.9, momentum 0, eps 1e-10,
 * mean-square slot initialised to ones) step from the gradient buffer. */
int vn_apply(vn_engine* h);
/* sess.run([optMinimize, loss]) (VarNetUtility.py:1044) = vn_grad + vn_apply.  If
 * loss_out_dev != NULL the pre-update loss is copied there (device scalar, async). */
int vn_train_step(vn_engine* h, int32_t batch, float* loss_out_dev);
/* ManageTrainData.optimIter (VarNetUtility.py:1021-1047) for one process: `n` consecutive steps
 * (vn_grad + vn_apply) over batches[0..n), the pre-update loss of each step ADDED to the device scalar
 * *loss_acc_dev (may be NULL).  One host call per epoch instead of four per mini-batch; no sync. */
int vn_train_epoch(vn_engine* h, const int32_t* batches, int32_t n, float* loss_acc_dev);

/* ManageTrainData.splitLoss (VarNetUtility.py:1080-1088): out = {loss, BCloss, ICloss,
 * varLoss} (host doubles), lossVec_dev [n_k] or NULL.  Synchronises. */
int vn_eval_loss(vn_engine* h, int32_t batch, double out[4], float* lossVec_dev);

/* runSession(['model']) (VarNetUtility.py:1123-1128,1142; VarNet.py:1930): u = model(X). */
int vn_forward(vn_engine* h, const float* X_dev, int64_t n, float* u_dev);
int vn_forward_f64(vn_engine* h, const double* X_dev, int64_t n, double* u_dev);
/* NNModel.modelGrad's tf.gradients(model(Input), Input) (TFModel.py:536-541): u = model(X) [n] and
 * g = d u / d X[:, :dim] [n, dim] in one pass (value forward + value-adjoint sweep to the inputs, 2 F_pt per point).
 * Networks the 8-wave fused kernel serves (vn_kernel_path == VN_KERNEL_FUSED16), dim <= 3; VN_EUNSUPPORTED otherwise. */
int vn_forward_grad(vn_engine* h, const float* X_dev, int64_t n, float* u_dev, float* g_dev);
/* runSession(['model','residual']) (VarNetUtility.py:1130-1142; TFModel.py:743-754):
 * res = -u_t + diff*Lap(u) - (vel - diff_dx).grad(u) + source.  diff [n], vel [n,dim],
 * source [n] or NULL, diff_dx [n,dim] or NULL.  fp32 and fp64 forms. */
int vn_residual(vn_engine* h, const float* X_dev, const float* diff_dev, const float* vel_dev,
                const float* source_dev, const float* diff_dx_dev, int64_t n, float* u_dev,
                float* res_dev);
int vn_residual_f64(vn_engine* h, const double* X_dev, const double* diff_dev,
                    const double* vel_dev, const double* source_dev, const double* diff_dx_dev,
                    int64_t n, double* u_dev, double* res_dev);

/* ---- towers: one process per GPU, gradient SUM over RCCL --------------------------------------------
 * Replaces TFNN.towerSetup / sum_grads (TFModel.py:253-289, 342-377): the reference builds one NNModel per
 * device in ONE process and reduces the per-tower gradients with tf.reduce_sum on a controller device.
 * Here every GPU has its own process and handle; the handles are joined into one RCCL communicator and the
 * only collective per step is a SUM all-reduce of the P+4 floats of the gradient buffer
 * (gradient | loss, BC, IC, var -- the tower loss sums of TFModel.py:315-319 ride along).
 *   rank 0:    vn_comm_unique_id(id)            -> 128 opaque bytes (ncclUniqueId), host memory
 *   host:      ship `id` to every rank (any side channel: MPI, a TCP store, torch.distributed ...)
 *   all ranks: vn_comm_init(h, rank, world, id) -> collective; returns when every rank has joined
 * With a communicator attached, vn_train_step / vn_train_epoch run gradient -> all-reduce -> optimizer on
 * the engine stream without a host round trip; vn_grad + vn_allreduce_grad + vn_apply is the same step in
 * three calls.  RCCL is loaded at run time (librccl.so.1 by SONAME, or $VN_RCCL_LIB); without it the vn_comm_* 
 * entry points return VN_EUNSUPPORTED and everything else works. */
#define VN_COMM_ID_BYTES 128
/* VN_OK if RCCL can be loaded in this process (no collective, no GPU work): lets every rank probe locally BEFORE any
 * rank enters the collective bootstrap, so that all ranks take the same route. */
int vn_comm_available(void);
/* Version of the RCCL library this process loaded, as ncclGetVersion reports it (e.g. 22205 = 2.22.5); VN_EUNSUPPORTED
 * when it cannot be loaded.  No collective, no GPU work: a first multi-GPU record can say which library summed the gradient. */
int vn_comm_version(int32_t* version_out);
int vn_comm_unique_id(void* id_out_host);
int vn_comm_init(vn_engine* h, int32_t rank, int32_t world, const void* unique_id_host);
/* Ranks RCCL reports for the attached communicator (1 if none); rank_out may be NULL. */
int vn_comm_size(const vn_engine* h, int32_t* world_out, int32_t* rank_out);
int vn_comm_destroy(vn_engine* h);
/* ncclCommInitRank has no timeout.  A caller that runs vn_comm_init on a helper thread and stops waiting for it (a peer is
 * wedged) calls vn_comm_abandon from the thread that drives the engine: from then on the engine has no communicator and
 * takes none -- a vn_comm_init that is still running leaves its communicator aside when it returns (VN_ESTATE), one that had
 * come up is withdrawn WITHOUT ncclCommDestroy (which could block on the wedged peer) -- so vn_train_step / vn_train_epoch
 * never enqueue a collective on it.  Idempotent.  After abandonment do not call vn_destroy while a vn_comm_init may still
 * be running (it writes into the handle): leave the handle to the process exit. */
int vn_comm_abandon(vn_engine* h);
/* Failure under a communicator.  vn_train_step / vn_train_epoch on a handle with a communicator are gradient -> all-reduce
 * -> update.  When the gradient fails on ONE rank (bad batch index, HIP error) that rank returns its error without entering
 * the collective and with parameters, optimizer slots and step counter untouched; the all-reduce its peers enqueued for
 * that step never completes.  A failed step under a communicator therefore ends the JOB: the failing process must exit
 * non-zero and its launcher must end the peers (varnet_amd/launch.py and torch.distributed.run both do; the former also
 * reports the failing rank's last stage).  Nothing in the library waits for a dead peer on the host: the peers' host
 * threads block at their next synchronisation, not inside vn_train_*.
 * In-place SUM all-reduce of the gradient buffer over the communicator, on the engine stream. */
int vn_allreduce_grad(vn_engine* h);

/* Adam step counter (global_step, TFModel.py:312). */
int vn_get_step(const vn_engine* h, int64_t* step);

/* Mean duration (ms, HIP events on the engine stream) of the vn_allreduce_grad calls recorded since
 * vn_profile_begin (call before vn_profile_end; synchronises). */
int vn_profile_comm(vn_engine* h, double* mean_ms, int64_t* calls);

/* Which kernel family VN_KERNEL_AUTO resolved to for this network (VN_KERNEL_GENERIC / _FUSED / _FUSED16);
 * *two_pass = 1 when integ_num > 128 runs the fused kernel twice around the row-wise epilogue. */
int vn_kernel_path(const vn_engine* h, int32_t* kernel_out, int32_t* two_pass_out);

/* Name / mean duration (ms, HIP events on the engine's stream) of the dominant kernel of the
 * last `vn_profile_begin` .. `vn_profile_end` window: used by bench.py for the roofline
 * object.  vn_profile_end synchronises. */
int vn_profile_begin(vn_engine* h);
int vn_profile_end(vn_engine* h, double* mean_ms, int64_t* launches, char* name, int32_t name_len);

/* Measurement aid (bench.py `roofline.peak_measured`, `issue_model`), ~30 ms, synchronises: what this GPU sustains on
 * the two instruction streams the fused kernels are priced against, two waves per SIMD on every SIMD.
 * out[0] fp32 MFMA TFLOP/s (v_mfma_f32_16x16x4_f32 loop; datasheet 157.3), out[1] ms of that launch,
 * out[2] cycles per independent v_fma_f32 per SIMD (at the clock out[3]), out[3] GHz implied by 32 cycles per MFMA,
 * out[4] ms of the vector launch. */
int vn_debug_calibrate(vn_engine* h, double out[5]);

/* The same for the fp64 matrix pipe (vn_forward_f64 / vn_residual_f64 run on v_mfma_f64_16x16x4_f64): out[0] fp64 MFMA
 * TFLOP/s of a loop of independent MFMAs, out[1] ms of that launch, out[2] cycles one MFMA occupies a SIMD at the clock
 * `ghz` (vn_debug_calibrate's out[3]; 0 = not asked). */
int vn_debug_calibrate_f64(vn_engine* h, double ghz, double out[3]);

/* Test aid: route = 1 sends vn_residual, vn_residual_f64 and vn_forward_f64 of THIS engine to the per-thread kernels of
 * vn_pointwise.hip (the independent implementation the matrix-pipe kernels are checked against); route = 2 keeps vn_forward and
 * vn_residual on the f32-MFMA kernels (vn_pgrad16 / vn_taylor16) where the bf16-piece kernels of vn_split16.hip would run
 * (A/B and cross-check of the two matrix-pipe forms; exists in the tests' cross-check library only); 0 restores the automatic choice;
 * | 4: vn_set_dedup keeps the CSR-ordered copy of gcoef although it found it periodic in integ_num (the general path of the two
 * assembly kernels, cross-checked bit for bit against the table path); | 8: vn_eval_loss of a batch that carries a de-duplication map
 * runs the row-wise forward anyway (cross-check of the loss-only form of the de-duplicated assembly).  (Round 5: an environment variable
 * read on every call.) */
int vn_debug_point_route(vn_engine* h, int32_t route);

/* Diagnostic builds (-DVN_STAMPS) only: per-phase s_memtime cycle sums of workgroup 0 of the last
 * fused launch (zeros otherwise).  Never part of a timed or shipped build. */
int vn_debug_stamps(vn_engine* h, unsigned long long out[8]);

#ifdef __cplusplus
}
#endif
#endif /* VARNET_HIP_H */
