// Floor of a PERSISTENT multi-step training launch (round 4, review item 3): everything a step of vn_train_epoch does besides
// its tile loop, once as ONE launch that runs S steps with two grid barriers per step, once as the two launches per step the
// engine uses today.  Same residency as the fused kernel (256 workgroups x 512 threads x 160 KB of LDS), same traffic:
//   per step and workgroup:  read theta (P floats) and scatter it into LDS weight images (57 KB zero-filled first),
//                            write a [P] gradient partial,
//   then, all workgroups:    sum the 256 partials of a slice of P in the fixed order of vn_reduce_kernel, TF-1 Adam on the slice,
//                            write theta / m / v back.
// (A) persistent: prologue | partial store | GRID BARRIER | slice reduce + Adam | GRID BARRIER     -- repeated S times in one launch
// (B) two kernels: [prologue | partial store]  ->  [reduce + Adam, (P+63)/64 workgroups x 1024 threads]   -- 2 S launches
// Grid barrier: XCD-hierarchical, monotonic counters (MI355X_MICROARCH.md "barrier-xcd"): census of workgroups per XCD once,
// lane 0 release fence -> per-XCD arrival counter -> the XCD's last arriver adds to the top counter -> the last XCD publishes the
// epoch into every XCD's generation word -> every workgroup polls its own XCD's word (sc1 load + s_sleep, BOUNDED) -> acquire fence.
//   hipcc --offload-arch=gfx950 -O3 -o persistent_floor persistent_floor.hip ; ./persistent_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Bar {
  unsigned members[8];      // census: workgroups per XCD
  unsigned arrive[8];       // monotonic arrivals per XCD
  unsigned top;             // monotonic arrivals of XCD leaders
  unsigned census;          // flat arrival counter of the census phase
  unsigned timeout;         // set when a bounded spin gave up
  unsigned pad[13];
  unsigned gen[8][32];      // generation word of each XCD on a line of its own
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7;
}

__device__ __forceinline__ bool spin_until(gu32* w, unsigned want, gu32* tmo) {
  for (unsigned spins = 0; spins < (1u << 22); ++spins) {
    if (__hip_atomic_load(w, RLX_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(tmo, 1u, RLX_AGENT);
  return false;
}

struct BarCtx { unsigned xcc, my_members, nx; };

// once per launch: how many workgroups sit on my XCD, how many XCDs take part
__device__ BarCtx bar_census(Bar* b) {
  __shared__ BarCtx ctx;
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    __hip_atomic_fetch_add((gu32*)&b->members[x], 1u, RLX_AGENT);
    __hip_atomic_fetch_add((gu32*)&b->census, 1u, RLX_AGENT);
    spin_until((gu32*)&b->census, gridDim.x, (gu32*)&b->timeout);
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += __hip_atomic_load((gu32*)&b->members[i], RLX_AGENT) ? 1u : 0u;
    ctx.xcc = x; ctx.my_members = __hip_atomic_load((gu32*)&b->members[x], RLX_AGENT); ctx.nx = nx;
  }
  __syncthreads();
  return ctx;
}

// epoch e = 1, 2, ...: every workgroup calls it the same number of times
__device__ __forceinline__ void grid_barrier(Bar* b, const BarCtx& c, unsigned e) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its own stores
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add((gu32*)&b->arrive[c.xcc], 1u, RLX_AGENT);
    if (old + 1 == c.my_members * e) {                       // last workgroup of this XCD
      const unsigned t = __hip_atomic_fetch_add((gu32*)&b->top, 1u, RLX_AGENT);
      if (t + 1 == c.nx * e)                                 // last XCD: release everyone
        for (int i = 0; i < 8; ++i) __hip_atomic_store((gu32*)&b->gen[i][0], e, RLX_AGENT);
    }
    spin_until((gu32*)&b->gen[c.xcc][0], e, (gu32*)&b->timeout);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

extern __shared__ __attribute__((aligned(16))) float lds[];
constexpr int IMG = 57 * 256;          // floats of weight images (57 KB)

// what the fused kernel does outside its tile loop, per step
__device__ __forceinline__ void step_prologue_and_partial(const float* theta, int P, float* partial, float salt) {
  for (int i = threadIdx.x; i < IMG / 4; i += blockDim.x) reinterpret_cast<float4*>(lds)[i] = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += blockDim.x) lds[(i * 7) % IMG] = theta[i];          // scatter into the images
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * P;
  for (int i = threadIdx.x; i < P; i += blockDim.x) out[i] = lds[(i * 7) % IMG] * 1e-3f + salt;   // "gradient" partial
}

// slice [p0, p1) of the parameters: fixed-order sum of the partials (16 groups of lanes, as vn_reduce_kernel) + TF-1 Adam
__device__ __forceinline__ void slice_reduce_adam(const float* partial, int nparts, int P, int p0, int p1, float* theta, float* m,
                                                  float* v, float lr) {
  float* sub = lds + IMG;                                   // [16][64]
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;            // 8 waves: two passes of 8 groups
  for (int base = p0; base < p1; base += 64) {
    const int p = base + lane;
    for (int gg = grp; gg < 16; gg += 8) {
      float acc = 0.f;
      if (p < p1)
        for (int g = gg; g < nparts; g += 16) acc += partial[(size_t)g * P + p];
      sub[gg * 64 + lane] = acc;
    }
    __syncthreads();
    if (grp == 0 && p < p1) {
      float t = sub[lane];
      for (int j = 1; j < 16; ++j) t += sub[j * 64 + lane];
      const float mi = 0.9f * m[p] + 0.1f * t, vi = 0.999f * v[p] + 0.001f * t * t;
      m[p] = mi; v[p] = vi;
      theta[p] = theta[p] - lr * mi / (sqrtf(vi) + 1e-8f);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void k_persistent(Bar* bar, float* theta, float* m, float* v, float* partial, int P, int steps) {
  const BarCtx c = bar_census(bar);
  const int per = (P + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per < P ? blockIdx.x * per : P, p1 = p0 + per < P ? p0 + per : P;
  unsigned e = 0;
  for (int s = 0; s < steps; ++s) {
    step_prologue_and_partial(theta, P, partial, (float)s);
    grid_barrier(bar, c, ++e);
    slice_reduce_adam(partial, gridDim.x, P, p0, p1, theta, m, v, 1e-3f);
    grid_barrier(bar, c, ++e);
  }
}

__global__ __launch_bounds__(512) void k_step(const float* theta, float* partial, int P, float salt) {
  step_prologue_and_partial(theta, P, partial, salt);
}
__global__ __launch_bounds__(1024) void k_reduce(const float* partial, int nparts, int P, float* theta, float* m, float* v) {
  __shared__ float sub[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (p < P)
    for (int g = grp; g < nparts; g += 16) acc += partial[(size_t)g * P + p];
  sub[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && p < P) {
    float t = sub[0][lane];
    for (int j = 1; j < 16; ++j) t += sub[j][lane];
    const float mi = 0.9f * m[p] + 0.1f * t, vi = 0.999f * v[p] + 0.001f * t * t;
    m[p] = mi; v[p] = vi;
    theta[p] = theta[p] - 1e-3f * mi / (sqrtf(vi) + 1e-8f);
  }
}
// barriers alone
__global__ __launch_bounds__(512) void k_barriers(Bar* bar, int n) {
  const BarCtx c = bar_census(bar);
  for (int i = 1; i <= n; ++i) grid_barrier(bar, c, (unsigned)i);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  const size_t ldsb = (size_t)(IMG + 16 * 64) * 4 + 100 * 1024;        // 160 KB class request: one workgroup per CU
  CK(hipFuncSetAttribute((const void*)k_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(159 * 1024)));
  CK(hipFuncSetAttribute((const void*)k_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(159 * 1024)));
  CK(hipFuncSetAttribute((const void*)k_barriers, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(159 * 1024)));
  const size_t lds = ldsb > 159 * 1024 ? 159 * 1024 : ldsb;     // (a few static bytes: the barrier context)
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_persistent, 512, lds));
  printf("%s: %d CUs, occupancy query %d workgroup(s) per CU at 512 threads + %zu B LDS\n", prop.gcnArchName, ncu, occ, lds);
  if (occ < 1) return 1;
  Bar* bar; CK(hipMalloc(&bar, sizeof(Bar)));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int S = 200;
  {
    CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
    hipLaunchKernelGGL(k_barriers, dim3(ncu), dim3(512), lds, s, bar, 10);       // warm-up
    CK(hipStreamSynchronize(s));
    CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_barriers, dim3(ncu), dim3(512), lds, s, bar, 2 * S);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb; CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    printf("grid barrier alone (%d workgroups, XCD census %u %u %u %u %u %u %u %u): %.2f us each%s\n", ncu, hb.members[0], hb.members[1],
           hb.members[2], hb.members[3], hb.members[4], hb.members[5], hb.members[6], hb.members[7], ms / (2 * S) * 1e3,
           hb.timeout ? "  (A SPIN TIMED OUT)" : "");
    if (hb.timeout) return 2;
  }
  for (int P : {921, 7851, 10451}) {
    float *theta, *m, *v, *partial;
    CK(hipMalloc(&theta, P * 4)); CK(hipMalloc(&m, P * 4)); CK(hipMalloc(&v, P * 4)); CK(hipMalloc(&partial, (size_t)ncu * P * 4));
    std::vector<float> h(P, 0.01f);
    auto reset = [&]() { hipMemcpy(theta, h.data(), P * 4, hipMemcpyHostToDevice); hipMemset(m, 0, P * 4); hipMemset(v, 0, P * 4); };
    // (A) one persistent launch
    reset();
    CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
    hipLaunchKernelGGL(k_persistent, dim3(ncu), dim3(512), lds, s, bar, theta, m, v, partial, P, 5);
    CK(hipStreamSynchronize(s));
    reset();
    CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_persistent, dim3(ncu), dim3(512), lds, s, bar, theta, m, v, partial, P, S);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float msA; CK(hipEventElapsedTime(&msA, e0, e1));
    std::vector<float> ta(P); CK(hipMemcpy(ta.data(), theta, P * 4, hipMemcpyDeviceToHost));
    Bar hb; CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    // (B) two launches per step
    reset();
    for (int i = 0; i < 5; ++i) {
      hipLaunchKernelGGL(k_step, dim3(ncu), dim3(512), lds, s, theta, partial, P, (float)i);
      hipLaunchKernelGGL(k_reduce, dim3((P + 63) / 64), dim3(1024), 0, s, partial, ncu, P, theta, m, v);
    }
    CK(hipStreamSynchronize(s));
    reset();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < S; ++i) {
      hipLaunchKernelGGL(k_step, dim3(ncu), dim3(512), lds, s, theta, partial, P, (float)i);
      hipLaunchKernelGGL(k_reduce, dim3((P + 63) / 64), dim3(1024), 0, s, partial, ncu, P, theta, m, v);
    }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float msB; CK(hipEventElapsedTime(&msB, e0, e1));
    std::vector<float> tb(P); CK(hipMemcpy(tb.data(), theta, P * 4, hipMemcpyDeviceToHost));
    int diff = 0;
    for (int i = 0; i < P; ++i) diff += ta[i] != tb[i];
    printf("P = %5d: persistent launch %.2f us per step | two launches per step %.2f us per step | parameters after %d steps differ in %d of %d "
           "entries%s\n", P, msA / S * 1e3, msB / S * 1e3, S, diff, P, hb.timeout ? "  (A SPIN TIMED OUT)" : "");
    hipFree(theta); hipFree(m); hipFree(v); hipFree(partial);
    if (hb.timeout) return 2;
  }
  return 0;
}
