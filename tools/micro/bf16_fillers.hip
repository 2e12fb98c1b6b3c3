// Which vector instructions hide between v_mfma_f32_16x16x32_bf16?  8 MFMAs per iteration (distinct accumulators, two
// waves per SIMD) + N filler instructions of one kind after each MFMA; whole-kernel time.  An MFMA occupies the matrix
// pipe for 16 cycles and the issue port for 8: fillers that fit in the other 8 should be free.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/bf16_fillers.hip -o tools/micro/bf16_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int N, int MF>
__global__ void __launch_bounds__(512) kern(float* out, int iters) {
  f32x4 acc[8];
  float v[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 77u + i; }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (short)(0x3f80 + (threadIdx.x & 7)); hb[i] = (short)(0x3f00 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int r = (i + 3 * j) & 7;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(b), "v"(a));
        if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[r]));
        if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "s"(0x07060302));
        if (KIND == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[r]) : "v"(a));
        if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
        if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
        if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[r & 6])) : "v"(*reinterpret_cast<double*>(&v[(r + 2) & 6])));
        if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
        if (KIND == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(v[r]), "v"(v[(r + 1) & 7]));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int N, int MF>
double run() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    kern<KIND, N, MF><<<256, 512>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipFree(out);
  return ms * 1e-3 / iters * 2.4e9 / 16.0;          // cycles per MFMA slot of the SIMD (two waves x 8 MFMAs)
}

template <int KIND>
void row(const char* name) {
  printf("%-18s alone (2/slot) %5.1f | beside MFMAs: 1/slot %5.1f  2/slot %5.1f  3/slot %5.1f  4/slot %5.1f   (cycles per MFMA slot; MFMA alone %.1f)\n",
         name, run<KIND, 2, 0>(), run<KIND, 1, 1>(), run<KIND, 2, 1>(), run<KIND, 3, 1>(), run<KIND, 4, 1>(), run<0, 0, 1>());
}

int main() {
  row<0>("v_fma_f32");
  row<1>("v_and_b32");
  row<2>("v_perm_b32");
  row<3>("v_sub_f32");
  row<4>("v_exp_f32");
  row<5>("v_rcp_f32");
  row<6>("v_pk_add_f32");
  row<7>("v_mov_b32");
  row<8>("v_cvt_pk_bf16_f32");
  return 0;
}
