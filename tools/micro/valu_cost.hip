// Microbenchmark: issue cost (cycles per wave-instruction, one wave per SIMD) of the vector
// instructions the fused kernel uses, alone and beside v_mfma_f32_16x16x4_f32.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_cost.hip -o tools/micro/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { FMA, PKFMA, PKMUL, PKADD, EXP, RCP, MOV, CNDMASK, DSREAD, DSREAD128, DSWRITE, BPERM, MUL, PERMSWAP };

template <int OP, int NOPS, int NM>
__global__ void __launch_bounds__(512) kern(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x4 acc[4];
  f32x2 p[8];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{v[i], v[i] + 1.f}; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  f32x2 pb = f32x2{b, b}, pa = f32x2{a, a};
  int addr = (threadIdx.x & 63) * 4;
  int addr16 = (threadIdx.x & 63) * 16;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NOPS; ++j) {
      const int r = j & 7;
      if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(b), "v"(a));
      if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[r]) : "v"(b));
      if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(pb), "v"(pa));
      if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(pb));
      if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(pb));
      if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
      if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
      if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(b));
      if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(b));
      if (OP == DSREAD) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(addr));
      if (OP == DSREAD128) asm volatile("ds_read_b128 %0, %1" : "=v"(acc[r & 3]) : "v"(addr16));
      if (OP == DSWRITE) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v[r]));
      if (OP == BPERM) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[r]) : "v"(addr), "v"(b));
      if (OP == PERMSWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 1) & 7]));
    }
    if (OP == DSREAD || OP == DSREAD128 || OP == DSWRITE || OP == BPERM) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP, int NOPS, int NM>
double run(int threads) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float)); (void)hipMalloc(&cyc, 8);
  const int iters = 4000;
  kern<OP, NOPS, NM><<<256, threads>>>(out, cyc, iters);
  kern<OP, NOPS, NM><<<256, threads>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  (void)hipFree(out); (void)hipFree(cyc);
  return (double)c / iters;
}

template <int OP>
void row(const char* name) {
  for (int threads : {256, 512}) {
    const double base = run<OP, 0, 8>(threads);         // 8 MFMAs alone
    const double alone = run<OP, 32, 0>(threads);       // 32 ops alone
    const double both = run<OP, 32, 8>(threads);        // 8 MFMAs then 32 ops
    printf("%-22s thr %3d: alone %6.2f cyc/op   8 mfma %6.1f   8 mfma + 32 ops %6.1f  => marginal %6.2f cyc/op\n", name, threads,
           alone / 32, base, both, (both - base) / 32);
  }
}

int main() {
  row<FMA>("v_fma_f32");
  row<MUL>("v_mul_f32");
  row<PKFMA>("v_pk_fma_f32");
  row<PKMUL>("v_pk_mul_f32");
  row<PKADD>("v_pk_add_f32");
  row<EXP>("v_exp_f32");
  row<RCP>("v_rcp_f32");
  row<MOV>("v_mov_b32");
  row<CNDMASK>("v_cndmask_b32");
  row<DSREAD>("ds_read_b32");
  row<DSREAD128>("ds_read_b128");
  row<DSWRITE>("ds_write_b32");
  row<BPERM>("ds_bpermute_b32");
  row<PERMSWAP>("v_permlane32_swap");
  return 0;
}
