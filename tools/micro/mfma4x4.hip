// Layout and timing probe for v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4, k = 1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  float a = 100.f + l, b = 1000.f + 10.f * l;          // encode lane in the operand
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = d[i];
}
template <int DEP>
__global__ void timing(float* out, unsigned long long* cyc, int iters) {
  float a = threadIdx.x, b = 1.f;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[DEP ? 0 : (i & 3)] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[DEP ? 0 : (i & 3)], 0, 0, 0);
  }
  unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 4 * sizeof(float)); hipMalloc(&cyc, 8);
  probe<<<1, 64>>>(out);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  // d = a(lane x) * b(lane y): solve x, y from d = (100+x)(1000+10y)
  for (int l = 0; l < 64; l += 1) {
    if (l < 8 || l % 16 == 0) {
      printf("lane %2d:", l);
      for (int i = 0; i < 4; ++i) {
        int fx = -1, fy = -1;
        for (int x = 0; x < 64; ++x) for (int y = 0; y < 64; ++y) if ((100.f + x) * (1000.f + 10.f * y) == h[l * 4 + i]) { fx = x; fy = y; }
        printf("  reg%d = A[lane %2d] * B[lane %2d]", i, fx, fy);
      }
      printf("\n");
    }
  }
  for (int dep = 0; dep < 2; ++dep) {
    if (dep) timing<1><<<1, 64>>>(out, cyc, 1000); else timing<0><<<1, 64>>>(out, cyc, 1000);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s chain: %.2f cycles per 4x4x1 MFMA\n", dep ? "dependent" : "4 accumulators", (double)c / 16000.0);
  }
  return 0;
}
