/*
 * A host WITHOUT Python or PyTorch driving the VarNet engine through the C ABI alone
 * (include/varnet_hip.h): plain C, device memory from the HIP runtime API, nothing else.
 * It plays the role of the reference's `sess.run([optMinimize, loss], feed_dict)` loop
 * (/root/reference/VarNetUtility.py:1021-1047) on a small synthetic 2D+t batch.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host_step.c \
 *       -L/opt/rocm/lib -lamdhip64 -ldl -lm -o c_host_step
 *   ./c_host_step varnet_amd/libvarnet_hip.so
 */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "varnet_hip.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); \
  if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, p_vn_last_error()); return 3; } } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 4; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 65535.0f; }

int main(int argc, char** argv) {
  void* lib = dlopen(argc > 1 ? argv[1] : "varnet_amd/libvarnet_hip.so", RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  SYM(vn_last_error) SYM(vn_create) SYM(vn_destroy) SYM(vn_params_init) SYM(vn_param_count) SYM(vn_set_fe_table)
  SYM(vn_set_interior) SYM(vn_set_bic) SYM(vn_set_weights) SYM(vn_train_epoch) SYM(vn_eval_loss) SYM(vn_params_get)
  SYM(vn_get_step) SYM(vn_kernel_path) SYM(vn_comm_size)

  enum { Q = 64, NK = 512, NB = 300, BDOF = 180, DIN = 3, DIM = 2 };
  const long n = (long)NK * Q;
  vn_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.dim = DIM; cfg.d_in = DIN; cfg.n_layers = 3; cfg.widths[0] = cfg.widths[1] = cfg.widths[2] = 20;
  cfg.activation = VN_ACT_SIGMOID; cfg.integ_num = Q; cfg.time_dependent = 1; cfg.device = 0;
  cfg.optimizer = VN_OPT_ADAM; cfg.kernel = VN_KERNEL_AUTO; cfg.lr = 1e-3; cfg.beta1 = 0.9; cfg.beta2 = 0.999; cfg.eps = 1e-8;
  vn_engine* h = NULL;
  CK(p_vn_create(&cfg, &h));
  CK(p_vn_params_init(h, 7));
  int64_t P = 0;
  CK(p_vn_param_count(h, &P));
  int32_t kern = -1, two = -1;
  CK(p_vn_kernel_path(h, &kern, &two));

  unsigned seed = 12345u;
  float *X = malloc(n * DIN * sizeof(float)), *G = malloc(n * DIM * sizeof(float));
  float *Xb = malloc(NB * DIN * sizeof(float)), *Lb = malloc(NB * sizeof(float));
  float N1[Q], dNt[Q];
  for (long i = 0; i < n * DIN; ++i) X[i] = 2.f * frand(&seed) - 1.f;
  for (long i = 0; i < n * DIM; ++i) G[i] = 0.2f * (frand(&seed) - 0.5f);
  for (int i = 0; i < NB * DIN; ++i) Xb[i] = 2.f * frand(&seed) - 1.f;
  for (int i = 0; i < NB; ++i) Lb[i] = sinf(3.f * Xb[i * DIN]);
  for (int p = 0; p < Q; ++p) { N1[p] = 0.25f + 0.5f * frand(&seed); dNt[p] = 0.1f * (frand(&seed) - 0.5f); }
  CK(p_vn_set_fe_table(h, N1, dNt, NULL));

  float *dX, *dG, *dXb, *dLb, *dacc;
  HK(hipMalloc((void**)&dX, n * DIN * sizeof(float)));   HK(hipMemcpy(dX, X, n * DIN * sizeof(float), hipMemcpyHostToDevice));
  HK(hipMalloc((void**)&dG, n * DIM * sizeof(float)));   HK(hipMemcpy(dG, G, n * DIM * sizeof(float), hipMemcpyHostToDevice));
  HK(hipMalloc((void**)&dXb, NB * DIN * sizeof(float))); HK(hipMemcpy(dXb, Xb, NB * DIN * sizeof(float), hipMemcpyHostToDevice));
  HK(hipMalloc((void**)&dLb, NB * sizeof(float)));       HK(hipMemcpy(dLb, Lb, NB * sizeof(float), hipMemcpyHostToDevice));
  HK(hipMalloc((void**)&dacc, sizeof(float)));
  CK(p_vn_set_interior(h, 0, dX, dG, NULL, NK, NULL, 1e-3, NULL, NULL));
  CK(p_vn_set_bic(h, dXb, dLb, NB, BDOF, 2.0));
  const double w[3] = {10.0, 10.0, 1.0};
  CK(p_vn_set_weights(h, w));

  double l0[4], l1[4];
  CK(p_vn_eval_loss(h, 0, l0, NULL));
  const int32_t batches[1] = {0};
  float first = 0.f, last = 0.f;
  for (int epoch = 0; epoch < 200; ++epoch) {
    HK(hipMemset(dacc, 0, sizeof(float)));
    CK(p_vn_train_epoch(h, batches, 1, dacc));            /* gradient + TF-1 Adam, asynchronous */
    if (epoch == 0 || epoch == 199) {
      float v;
      HK(hipMemcpy(&v, dacc, sizeof v, hipMemcpyDeviceToHost));   /* the only host sync */
      if (epoch == 0) first = v; else last = v;
    }
  }
  CK(p_vn_eval_loss(h, 0, l1, NULL));
  int64_t step = 0;
  CK(p_vn_get_step(h, &step));
  int32_t world = 0, rank = -1;
  CK(p_vn_comm_size(h, &world, &rank));
  float* theta = malloc(P * sizeof(float));
  CK(p_vn_params_get(h, theta, P));
  int finite = 1;
  for (int64_t i = 0; i < P; ++i) finite &= isfinite(theta[i]) ? 1 : 0;
  printf("params %lld kernel %d steps %lld world %d | loss %.6f -> %.6f (eval: %.6f -> %.6f; BC %.4f IC %.4f var %.4f)\n",
         (long long)P, kern, (long long)step, world, first, last, l0[0], l1[0], l1[1], l1[2], l1[3]);
  const int ok = finite && step == 200 && world == 1 && fabs(first - l0[0]) <= 1e-4 * fabs(l0[0]) && l1[0] < 0.8 * l0[0];
  CK(p_vn_destroy(h));
  (void)hipFree(dX); (void)hipFree(dG); (void)hipFree(dXb); (void)hipFree(dLb); (void)hipFree(dacc);
  puts(ok ? "C_HOST_OK" : "C_HOST_FAILED");
  return ok ? 0 : 5;
}
