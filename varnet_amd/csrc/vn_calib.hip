// Measurement aid, not on the training path: what this GPU sustains on the two instruction streams the fused kernels
// are priced against (bench.py `roofline.peak_measured`, `issue_model`):
//   * fp32 MFMA: a loop of independent v_mfma_f32_16x16x4_f32 chains, two waves per SIMD on every SIMD of the device
//     -> TFLOP/s next to the datasheet's 157.3 (MI355X_MICROARCH.md);
//   * fp32 vector issue: the same geometry on independent v_fma_f32 chains -> ns, and cycles at the clock the MFMA loop
//     implies (one 16x16x4 MFMA = 32 cycles of a SIMD), per vector instruction per SIMD.  The guide's 4 cycles are the issue
//     cost of ONE wave; the fused kernels run two waves per SIMD (tools/micro/overlap2.hip measured ~2.7 in round 1).
#include "vn_internal.h"

namespace {

typedef float f32x4c __attribute__((ext_vector_type(4)));
constexpr int CAL_THREADS = 512;          // 8 waves = 2 per SIMD
constexpr int CAL_MFMA = 64;              // MFMAs per wave and iteration
constexpr int CAL_VALU = 128;             // v_fma_f32 per wave and iteration

typedef double f64x4c __attribute__((ext_vector_type(4)));

template <int KIND>                       // 0: fp32 MFMA loop, 1: VALU loop, 2: fp64 MFMA loop
__global__ __launch_bounds__(CAL_THREADS) void vn_calib_kernel(float* out, int iters) {
  if (KIND == 2) {
    // independent v_mfma_f64_16x16x4_f64 chains (the instruction of vn_taylor16d.hip / vn_dgemm_nn), two waves per SIMD
    f64x4c dacc[4];
    for (int i = 0; i < 4; ++i) dacc[i] = f64x4c{0., 0., 0., 0.};
    const double da = threadIdx.x * 1e-3, db = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < CAL_MFMA; ++i) dacc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, dacc[i & 3], 0, 0, 0);
    }
    double ds = 0.;
    for (int i = 0; i < 4; ++i) ds += dacc[i][0] + dacc[i][1] + dacc[i][2] + dacc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (float)ds;
    return;
  }
  f32x4c acc[4];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4c{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < CAL_MFMA; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < CAL_VALU; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
hipError_t time_one(float* out, int grid, int iters, hipStream_t s, double* ms_best) {
  hipEvent_t e0, e1;
  hipError_t e = hipEventCreate(&e0);
  if (e != hipSuccess) return e;
  e = hipEventCreate(&e1);
  if (e != hipSuccess) { (void)hipEventDestroy(e0); return e; }
  double best = 1e30;
  for (int rep = 0; rep < 4 && e == hipSuccess; ++rep) {          // first repetition = warm-up (clocks, code object)
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(vn_calib_kernel<KIND>, dim3(grid), dim3(CAL_THREADS), 0, s, out, iters);
    e = hipGetLastError();
    (void)hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_best = best;
  return e;
}

}  // namespace

// out[0] fp32 MFMA TFLOP/s, out[1] ms of the best MFMA launch, out[2] cycles per v_fma_f32 per SIMD at two waves per SIMD
// (at the clock of out[3]), out[3] clock in GHz the MFMA loop implies (32 cycles per 16x16x4 MFMA), out[4] ms of the best VALU launch
hipError_t vn_calibrate(int ncu, hipStream_t s, double out[5]) {
  float* buf = nullptr;
  hipError_t e = hipMalloc((void**)&buf, (size_t)ncu * CAL_THREADS * sizeof(float));
  if (e != hipSuccess) return e;
  const int it_m = 3000, it_v = 6000;                // ~5 ms and ~1.7 ms per launch
  double ms_m = 0.0, ms_v = 0.0;
  e = time_one<0>(buf, ncu, it_m, s, &ms_m);
  if (e == hipSuccess) e = time_one<1>(buf, ncu, it_v, s, &ms_v);
  (void)hipFree(buf);
  if (e != hipSuccess) return e;
  const double waves = (double)ncu * (CAL_THREADS / 64);
  const double flop = waves * it_m * CAL_MFMA * 2.0 * 16 * 16 * 4;
  const double mfma_per_simd = 2.0 * it_m * CAL_MFMA;            // two waves per SIMD
  const double ghz = mfma_per_simd * 32.0 / (ms_m * 1e-3) / 1e9;
  const double valu_per_simd = 2.0 * it_v * CAL_VALU;
  out[0] = flop / (ms_m * 1e-3) / 1e12;
  out[1] = ms_m;
  out[2] = (ms_v * 1e-3) * ghz * 1e9 / valu_per_simd;
  out[3] = ghz;
  out[4] = ms_v;
  return hipSuccess;
}

// out[0] fp64 MFMA TFLOP/s sustained by a loop of independent v_mfma_f64_16x16x4_f64 (2 * 16 * 16 * 4 FLOP each), two waves
// per SIMD on every SIMD; out[1] ms of the best launch; out[2] cycles one such MFMA occupies a SIMD at `ghz` (the clock the
// fp32 loop of vn_calibrate implied in the same process; pass 0 to skip)
hipError_t vn_calibrate_f64(int ncu, hipStream_t s, double ghz, double out[3]) {
  float* buf = nullptr;
  hipError_t e = hipMalloc((void**)&buf, (size_t)ncu * CAL_THREADS * sizeof(float));
  if (e != hipSuccess) return e;
  const int it_d = 1500;                             // ~5 ms per launch
  double ms_d = 0.0;
  e = time_one<2>(buf, ncu, it_d, s, &ms_d);
  (void)hipFree(buf);
  if (e != hipSuccess) return e;
  const double waves = (double)ncu * (CAL_THREADS / 64);
  out[0] = waves * it_d * CAL_MFMA * 2.0 * 16 * 16 * 4 / (ms_d * 1e-3) / 1e12;
  out[1] = ms_d;
  out[2] = ghz > 0 ? (ms_d * 1e-3) * ghz * 1e9 / (2.0 * it_d * CAL_MFMA) : 0.0;
  return hipSuccess;
}
