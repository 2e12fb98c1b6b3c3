#include <hip/hip_runtime.h>
__device__ __forceinline__ float rowsum4(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  float s = a + b;
  float c = s, d = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  return c + d;
}
__global__ void k(float* p) { p[threadIdx.x] = rowsum4(p[threadIdx.x]); }
int main() {
  float h[64], *d; for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 16)) * 1.0f + 0.001f * (i % 16);
  hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice); k<<<1, 64>>>(d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 5) printf("%d:%g ", i, h[i]); printf("\n"); return 0;
}
