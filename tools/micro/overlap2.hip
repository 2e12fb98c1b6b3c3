// Microbenchmark: two waves per SIMD, NM f32 MFMAs (16x16x4) + NV independent v_fma_f32 per iteration,
// whole-kernel time (hipEvents).  Patterns: B = blocks (all MFMAs, then all VALU), both waves alike;
// C = complementary (waves 4-7 run VALU first); F = fine interleave (1 MFMA : NV/NM VALU).
// Optional workgroup barrier every iteration (lockstep) to mimic the fused kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/overlap2.hip -o tools/micro/overlap2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int PAT, int BAR>
__global__ void __launch_bounds__(512) kern(float* out, int iters) {
  f32x4 acc[4];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const bool second = (threadIdx.x >> 6) >= 4;
  auto mf = [&](int n) {
#pragma unroll
    for (int i = 0; i < n; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
  };
  auto va = [&](int n) {
#pragma unroll
    for (int j = 0; j < n; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
  };
  for (int it = 0; it < iters; ++it) {
    if (PAT == 0) { mf(NM); va(NV); }
    if (PAT == 1) { if (second) va(NV); mf(NM); if (!second) va(NV); }
    if (PAT == 2) {
#pragma unroll
      for (int i = 0; i < NM; ++i) { mf(1); va(NV / NM); }
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int PAT, int BAR>
double run(int threads) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 4000;
  kern<NM, NV, PAT, BAR><<<256, threads>>>(out, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  kern<NM, NV, PAT, BAR><<<256, threads>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms * 1e6 / iters;
}

template <int NM, int NV>
void row() {
  printf("NM %3d NV %3d | 256thr blocks %7.1f | 512thr: mfma-only %7.1f valu-only %7.1f | blocks %7.1f compl %7.1f fine %7.1f | +barrier: blocks %7.1f compl %7.1f fine %7.1f  (ns/iter)\n",
         NM, NV, run<NM, NV, 0, 0>(256), run<NM, 0, 0, 0>(512), run<0, NV, 0, 0>(512), run<NM, NV, 0, 0>(512), run<NM, NV, 1, 0>(512),
         run<NM, NV, 2, 0>(512), run<NM, NV, 0, 1>(512), run<NM, NV, 1, 1>(512), run<NM, NV, 2, 1>(512));
}

int main() {
  row<8, 16>();
  row<8, 32>();
  row<8, 64>();
  row<32, 64>();
  row<32, 128>();
  row<64, 128>();
  row<64, 256>();
  return 0;
}
