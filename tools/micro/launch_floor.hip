// Floor of a dependent kernel boundary on one stream: N back-to-back launches of a kernel that does (almost) nothing,
// for several grid / block / dynamic-LDS / bytes-written-per-workgroup combinations.   hipcc --offload-arch=gfx950 -O3
//   ./launch_floor            -> us per launch (HIP events around 400 launches)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float lds[];
__global__ void k(float* out, int wbytes, const float* in, int rbytes) {
  float acc = 0.f;
  for (int i = threadIdx.x * 4; i < rbytes; i += blockDim.x * 4) acc += in[i / 4];
  if (acc == 123.456f) lds[threadIdx.x] = acc;
  for (int i = threadIdx.x * 4; i < wbytes; i += blockDim.x * 4) out[(size_t)blockIdx.x * (wbytes / 4) + i / 4] = acc + (float)i;
}
int main() {
  float *out, *in;
  hipMalloc(&out, 256u << 20); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { int grid, block, lds, wb, rb; } cfg[] = {
      {1, 64, 0, 0, 0},        {256, 64, 0, 0, 0},       {256, 512, 0, 0, 0},       {256, 512, 64 << 10, 0, 0},
      {256, 512, 160 << 10, 0, 0}, {256, 512, 160 << 10, 4096, 0}, {256, 512, 160 << 10, 43008, 0}, {256, 512, 160 << 10, 4096, 43008},
      {256, 512, 160 << 10, 43008, 43008}, {64, 512, 160 << 10, 4096, 4096}, {1024, 256, 0, 0, 0}};
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
  for (auto c : cfg) {
    for (int i = 0; i < 20; ++i) k<<<c.grid, c.block, c.lds, s>>>(out, c.wb, in, c.rb);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 400; ++i) k<<<c.grid, c.block, c.lds, s>>>(out, c.wb, in, c.rb);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid %4d block %3d lds %6d B  writes %5d B/wg  reads %5d B/wg : %.2f us per launch\n", c.grid, c.block, c.lds, c.wb, c.rb, ms / 400 * 1e3);
  }
  return 0;
}
