// Microbenchmark: do f32 MFMAs (v_mfma_f32_16x16x4_f32) and f32 VALU work overlap on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
// Prints s_memtime cycles per loop iteration for NM MFMAs + NV v_fma_f32 per iteration at one and
// two waves per SIMD.  Overlap => t(M+V) ~ max(t(M), t(V)); none => ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int KIND>
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[8];
  float v[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = threadIdx.x * 0.001f + i; }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (short)(threadIdx.x + i); hb[i] = (short)(threadIdx.x * 3 + i); }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < NM) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float x = v[(i + j) & 7];
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
        v[(i + j) & 7] = x;
      }
    }
  }
  unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int KIND>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 20000;
  kern<NM, NV, KIND><<<256, threads>>>(out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<NM, NV, KIND><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s threads %3d  NM %d NV/MFMA-slot %d : %8.1f memtime-ticks/iter  %8.3f ns/iter\n", name, threads, NM, NV,
         (double)c / iters, ms * 1e6 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<8, 0, 0>("f32 16x16x4 only", threads);
    run<0, 2, 0>("valu only (16/iter)", threads);
    run<0, 4, 0>("valu only (32/iter)", threads);
    run<0, 6, 0>("valu only (48/iter)", threads);
    run<8, 2, 0>("f32 mfma + 2 valu/slot", threads);
    run<8, 4, 0>("f32 mfma + 4 valu/slot", threads);
    run<8, 6, 0>("f32 mfma + 6 valu/slot", threads);
    run<8, 0, 1>("bf16 16x16x32 only", threads);
    run<8, 2, 1>("bf16 mfma + 2 valu/slot", threads);
    run<8, 4, 1>("bf16 mfma + 4 valu/slot", threads);
  }
  return 0;
}
