// Microbenchmark + operand-map check: one 50-wide hidden layer of the forward pass (value + one tangent, 16 points per
// wave, 8 waves) in two forms
//   F32   : v_mfma_f32_16x16x4_f32, 3 row tiles x 13 k-steps x {value, tangent}  (the shipped kernel's matrix work
//           without its edge rows), weights as f32 in LDS;
//   SPLIT : every f32 operand cut into three bf16 pieces x = h + m + l (exact: 3 x 8 significand bits), six products
//           hh, hm, mh, hl, lh, mm on v_mfma_f32_16x16x32_bf16 with f32 accumulation -- error of the dropped terms
//           (ml, lm, ll) <= 3 * 2^-24 relative, i.e. fp32-class; bf16 MFMAs run at 16x the f32 MFMA rate and, unlike
//           those, do not share the datapath with the f32 VALU.
// The SPLIT form keeps the shipped kernel's feature <-> accumulator-row mapping (feature f: k-step f/4, lane group f%4,
// row 16(ks>>2) + 4g + (ks&3)), so a layer's accumulator tiles are the next layer's B operand with no lane movement:
// B fragment q (k-steps 8q..8q+7) = the lane's 8 activation values packed pairwise.
// Weight image: 1 KB blocks [piece 3][q 2][row tile 4] of [g 4][row c ^ 12(g&1)][8 bf16]: the row read of the forward
// pass is one conflict-free ds_read_b128 per fragment (the first layout tried, 32-byte rows read with ds_read_b64, was
// merged into ds_read2st64_b64 by the compiler -- half rate, 32-bank modulus, 4-way conflicts: LDS-bound at 2.2x the
// time); the XOR keeps the transposed read of the reverse pass (ds_read_b64_tr_b16 on the same image) 2-way.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/split_layer.hip -o tools/micro/split_layer && tools/micro/split_layer
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int KS = 13, H = 50, WS = 65;
#ifndef VAR
#define VAR 0      // 1: no split arithmetic (pieces = raw bits), 2: no MFMAs, 3: no sigmoid chain either
#endif
__host__ __device__ constexpr int vpos(int ks, int g) { return 16 * (ks >> 2) + 4 * g + (ks & 3); }
__host__ __device__ constexpr int vks(int pos) { return 4 * (pos >> 4) + (pos & 3); }
__host__ __device__ constexpr int vfeat(int pos) { return 4 * vks(pos) + ((pos >> 2) & 3); }

constexpr int BLK = 1024;                      // one (piece, q, row tile) block: [g][c ^ 12(g&1)][8 bf16], read with ds_read_b128
constexpr int IMG_SPLIT = 24 * BLK;            // 24 KB
constexpr int IMG_F32 = 52 * WS * 4;           // f32 image [k row = feature][WS]

__device__ __forceinline__ u32 fu(float x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ float uf(u32 x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ u32 pack_hi(u32 u1, u32 u0) { return __builtin_amdgcn_perm(u1, u0, 0x07060302u); }

// exact three-way split of two f32 values into packed bf16 pairs (truncation: h = top 8 significand bits, ...)
__device__ __forceinline__ void split2(f32x2 x, u32& h, u32& m, u32& l) {
  const u32 u0 = fu(x[0]), u1 = fu(x[1]);
  h = pack_hi(u1, u0);
#ifdef NOPK
  // packed f32 instructions are expensive beside bf16 MFMAs (MI355X_MICROARCH.md, cycle constants): scalar subtracts
  const float r0 = x[0] - uf(u0 & 0xffff0000u), r1 = x[1] - uf(u1 & 0xffff0000u);
  const u32 v0 = fu(r0), v1 = fu(r1);
  m = pack_hi(v1, v0);
  const float s0 = r0 - uf(v0 & 0xffff0000u), s1 = r1 - uf(v1 & 0xffff0000u);
  l = pack_hi(fu(s1), fu(s0));
#else
  const f32x2 r = x - f32x2{uf(u0 & 0xffff0000u), uf(u1 & 0xffff0000u)};
  const u32 v0 = fu(r[0]), v1 = fu(r[1]);
  m = pack_hi(v1, v0);
  const f32x2 s = r - f32x2{uf(v0 & 0xffff0000u), uf(v1 & 0xffff0000u)};
  l = pack_hi(fu(s[1]), fu(s[0]));
#endif
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  if (VAR == 2) { c[0] += uf((a[0] ^ b[0]) & 0x3fffffffu); return c; }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x2 sigmoid2(f32x2 z) {
#ifdef NOPK
  const float e0 = __builtin_amdgcn_exp2f(-1.4426950408889634f * z[0]), e1 = __builtin_amdgcn_exp2f(-1.4426950408889634f * z[1]);
  return f32x2{__builtin_amdgcn_rcpf(1.f + e0), __builtin_amdgcn_rcpf(1.f + e1)};
#endif
  const f32x2 t = z * f32x2{-1.4426950408889634f, -1.4426950408889634f};
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.f, 1.f};
  return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// one layer, SPLIT form.  pv/pt: pre-activations z and tangent zdot of the previous layer (4 accumulator tiles)
__device__ __forceinline__ void layer_split(const f32x4 (&pv)[4], const f32x4 (&pt)[4], const char* img, const float* bias,
                                            int c, int g, f32x4 (&nv)[4], f32x4 (&nt)[4]) {
  u32x4 Ba[2][3], Bq[2][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {                 // pair j = k-steps 2j, 2j+1
    u32 h, m, l, hq, mq, lq;
    if (2 * j < KS) {
      const int t = (2 * j) >> 2, i = (2 * j) & 3;
      f32x2 z = {pv[t][i], pv[t][i + 1]}, zd = {pt[t][i], pt[t][i + 1]};
      f32x2 a = (VAR >= 3) ? z : sigmoid2(z);
      f32x2 q = (VAR >= 3) ? zd : (a - a * a) * zd;
      if (2 * j + 1 >= KS) { a[1] = 0.f; q[1] = 0.f; }
      if (VAR == 1 || VAR == 3 || VAR == 4 || VAR == 5) { h = fu(a[0]); m = fu(a[1]); l = h ^ m; hq = fu(q[0]); mq = fu(q[1]); lq = hq ^ mq; }
      else { split2(a, h, m, l); split2(q, hq, mq, lq); }
      if (VAR == 5) { m = l = h; mq = lq = hq = h; }
    } else {
      h = m = l = hq = mq = lq = 0u;
    }
    Ba[j >> 2][0][j & 3] = h; Ba[j >> 2][1][j & 3] = m; Ba[j >> 2][2][j & 3] = l;
    Bq[j >> 2][0][j & 3] = hq; Bq[j >> 2][1][j & 3] = mq; Bq[j >> 2][2][j & 3] = lq;
  }
  const char* rd = img + (g * 16 + (c ^ (12 * (g & 1)))) * 16;
#ifdef PRIO
  __builtin_amdgcn_s_setprio(PRIO);
#endif
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    nv[mt] = *reinterpret_cast<const f32x4*>(&bias[mt * 16 + g * 4]);
    nt[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u32x4 A[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const u32x4*>(rd + ((p * 2 + q) * 4 + mt) * BLK);
      // small terms first
      nv[mt] = mfma_bf16(A[1], Ba[q][1], nv[mt]);  nt[mt] = mfma_bf16(A[1], Bq[q][1], nt[mt]);
      nv[mt] = mfma_bf16(A[2], Ba[q][0], nv[mt]);  nt[mt] = mfma_bf16(A[2], Bq[q][0], nt[mt]);
      nv[mt] = mfma_bf16(A[0], Ba[q][2], nv[mt]);  nt[mt] = mfma_bf16(A[0], Bq[q][2], nt[mt]);
      nv[mt] = mfma_bf16(A[1], Ba[q][0], nv[mt]);  nt[mt] = mfma_bf16(A[1], Bq[q][0], nt[mt]);
      nv[mt] = mfma_bf16(A[0], Ba[q][1], nv[mt]);  nt[mt] = mfma_bf16(A[0], Bq[q][1], nt[mt]);
      nv[mt] = mfma_bf16(A[0], Ba[q][0], nv[mt]);  nt[mt] = mfma_bf16(A[0], Bq[q][0], nt[mt]);
    }
  }
#ifdef PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// ---- software-pipelined SPLIT form.  A wave's own stream can hide vector work only BETWEEN its bf16 MFMAs (an MFMA
// holds the SIMD's issue port for 8 of its 16 cycles; two waves per SIMD in the block-structured form above measure
// MFMA time + vector time, as if nothing overlapped).  The K loop is ordered q-major, so that
//   region 1: the q = 0 products of layer l (they need k-steps 0..7 = row tiles 0, 1 of layer l-1) run beside the
//             activation + split of row tiles 2, 3 of layer l-1,
//   region 2: the q = 1 products of layer l run beside the activation + split of ITS row tiles 0, 1, which complete
//             after 12 and 24 of the region's 48 MFMAs.
struct PipeState {
  f32x4 zv[4], zt[4];            // pre-activations of the last finished layer (tiles 2, 3 still to be activated)
  u32x4 Ba[2][3], Bq[2][3];      // B fragments of the layer being multiplied
};

__device__ __forceinline__ void act_split_pair(const f32x4 (&zv)[4], const f32x4 (&zt)[4], int j, u32x4 (&Ba)[2][3], u32x4 (&Bq)[2][3]) {
  u32 h, m, l, hq, mq, lq;
  if (2 * j < KS) {
    const int t = (2 * j) >> 2, i = (2 * j) & 3;
    f32x2 z = {zv[t][i], zv[t][i + 1]}, zd = {zt[t][i], zt[t][i + 1]};
    f32x2 a = sigmoid2(z);
#ifdef NOPK
    f32x2 q = {__builtin_fmaf(-a[0], a[0], a[0]) * zd[0], __builtin_fmaf(-a[1], a[1], a[1]) * zd[1]};
#else
    f32x2 q = (a - a * a) * zd;
#endif
    if (2 * j + 1 >= KS) { a[1] = 0.f; q[1] = 0.f; }
    split2(a, h, m, l);
    split2(q, hq, mq, lq);
  } else {
    h = m = l = hq = mq = lq = 0u;
  }
  Ba[j >> 2][0][j & 3] = h; Ba[j >> 2][1][j & 3] = m; Ba[j >> 2][2][j & 3] = l;
  Bq[j >> 2][0][j & 3] = hq; Bq[j >> 2][1][j & 3] = mq; Bq[j >> 2][2][j & 3] = lq;
}

__device__ __forceinline__ void load_A(u32x4 (&A)[3], const char* rd, int q, int mt) {
#pragma unroll
  for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const u32x4*>(rd + ((p * 2 + q) * 4 + mt) * BLK);
}
template <int Q>
__device__ __forceinline__ void mfma_tile(PipeState& S, f32x4 (&nv)[4], f32x4 (&nt)[4], const char* rd, int mt, u32x4 (&A)[3]) {
#ifdef PREFETCH
  u32x4 An[3];
  load_A(An, rd, (mt == 3) ? 1 - Q : Q, (mt + 1) & 3);
#else
  load_A(A, rd, Q, mt);
#endif
  nv[mt] = mfma_bf16(A[1], S.Ba[Q][1], nv[mt]);  nt[mt] = mfma_bf16(A[1], S.Bq[Q][1], nt[mt]);
  nv[mt] = mfma_bf16(A[2], S.Ba[Q][0], nv[mt]);  nt[mt] = mfma_bf16(A[2], S.Bq[Q][0], nt[mt]);
  nv[mt] = mfma_bf16(A[0], S.Ba[Q][2], nv[mt]);  nt[mt] = mfma_bf16(A[0], S.Bq[Q][2], nt[mt]);
  nv[mt] = mfma_bf16(A[1], S.Ba[Q][0], nv[mt]);  nt[mt] = mfma_bf16(A[1], S.Bq[Q][0], nt[mt]);
  nv[mt] = mfma_bf16(A[0], S.Ba[Q][1], nv[mt]);  nt[mt] = mfma_bf16(A[0], S.Bq[Q][1], nt[mt]);
  nv[mt] = mfma_bf16(A[0], S.Ba[Q][0], nv[mt]);  nt[mt] = mfma_bf16(A[0], S.Bq[Q][0], nt[mt]);
#ifdef PREFETCH
#pragma unroll
  for (int p = 0; p < 3; ++p) A[p] = An[p];
#endif
}

#ifndef NVAL
#define NVAL 2
#endif
template <int N>
__device__ __forceinline__ void interleave_pattern() {           // N x { 1 MFMA, NVAL VALU }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, NVAL, 0);
  }
}

__device__ __forceinline__ void layer_pipe(PipeState& S, const char* img, const float* bias, int c, int g, u32x4 (&A)[3]) {
  const char* rd = img + (g * 16 + (c ^ (12 * (g & 1)))) * 16;
  f32x4 nv[4], nt[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    nv[mt] = *reinterpret_cast<const f32x4*>(&bias[mt * 16 + g * 4]);
    nt[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // region 1
#pragma unroll
  for (int j = 4; j < 8; ++j) act_split_pair(S.zv, S.zt, j, S.Ba, S.Bq);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) mfma_tile<0>(S, nv, nt, rd, mt, A);
#ifdef PATTERN
  interleave_pattern<48>();
#endif
  __builtin_amdgcn_sched_barrier(0);
  // region 2
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) mfma_tile<1>(S, nv, nt, rd, mt, A);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { S.zv[mt] = nv[mt]; S.zt[mt] = nt[mt] + f32x4{1.f, 1.f, 1.f, 1.f}; }
#pragma unroll
  for (int j = 0; j < 4; ++j) act_split_pair(S.zv, S.zt, j, S.Ba, S.Bq);
#ifdef PATTERN
  interleave_pattern<48>();
#endif
  __builtin_amdgcn_sched_barrier(0);
}

// one layer, F32 form: image[k row = in feature 4ks+g][column = out position], stride WS (as vn_fused16.hip)
__device__ __forceinline__ void layer_f32(const f32x4 (&pv)[4], const f32x4 (&pt)[4], const float* img, const float* bias,
                                          int c, int g, f32x4 (&nv)[4], f32x4 (&nt)[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    nv[mt] = *reinterpret_cast<const f32x4*>(&bias[mt * 16 + g * 4]);
    nt[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < (KS + 1) / 2; ++j) {
    const int t = (2 * j) >> 2, i = (2 * j) & 3;
    f32x2 z = {pv[t][i], pv[t][i + 1]}, zd = {pt[t][i], pt[t][i + 1]};
    f32x2 a = sigmoid2(z);
    f32x2 q = (a - a * a) * zd;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ks = 2 * j + e;
      if (ks < KS) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const float w = img[(4 * ks + g) * WS + 16 * mt + c];
          nv[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, a[e], nv[mt], 0, 0, 0);
          nt[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, q[e], nt[mt], 0, 0, 0);
        }
      }
    }
  }
}

template <int FORM>
__global__ void __launch_bounds__(512) kern(const char* img_g, const float* bias_g, const float* z0, const float* zd0,
                                            float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int IMG = FORM ? IMG_SPLIT : IMG_F32;
  PipeState S;
  u32x4 Apre[3];
  load_A(Apre, lds + (((threadIdx.x >> 4) & 3) * 16 + ((threadIdx.x & 15) ^ (12 * ((threadIdx.x >> 4) & 1)))) * 16, 0, 0);
  for (int i = threadIdx.x; i < IMG / 4; i += blockDim.x) reinterpret_cast<u32*>(lds)[i] = reinterpret_cast<const u32*>(img_g)[i];
  float* bias = reinterpret_cast<float*>(lds + IMG);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) bias[i] = bias_g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int pt0 = (blockIdx.x * 8 + wave) * 16 + c;              // this lane's point
  f32x4 pv[4], pt[4], nv[4], nt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pv[t][i] = z0[(size_t)pt0 * 64 + 16 * t + 4 * g + i];
      pt[t][i] = zd0[(size_t)pt0 * 64 + 16 * t + 4 * g + i];
    }
#ifdef STAGGER
  if (wave >= 4) for (int i = 0; i < STAGGER; ++i) __builtin_amdgcn_s_sleep(8);        // 64 cycles each
#endif
  for (int it = 0; it < iters; ++it) {
    if (VAR != 4) asm volatile("" ::: "memory");          // keep the weight-fragment reads inside the loop
    if (FORM == 2) {
      if (it == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { S.zv[t] = pv[t]; S.zt[t] = pt[t]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) act_split_pair(S.zv, S.zt, j, S.Ba, S.Bq);
      }
      layer_pipe(S, lds, bias, c, g, Apre);
      continue;
    }
    if (FORM) layer_split(pv, pt, lds, bias, c, g, nv, nt);
    else layer_f32(pv, pt, reinterpret_cast<const float*>(lds), bias, c, g, nv, nt);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pv[t] = nv[t];
      pt[t] = nt[t] + (iters > 1 ? f32x4{1.f, 1.f, 1.f, 1.f} : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
  if (FORM == 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { pv[t] = S.zv[t]; pt[t] = S.zt[t] - (iters > 1 ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{1.f, 1.f, 1.f, 1.f}); }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[(size_t)pt0 * 128 + 16 * t + 4 * g + i] = pv[t][i];
      out[(size_t)pt0 * 128 + 64 + 16 * t + 4 * g + i] = pt[t][i];
    }
}

static unsigned short bf16_trunc(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float bf16_val(unsigned short b) { unsigned u = (unsigned)b << 16; float x; memcpy(&x, &u, 4); return x; }

int main(int argc, char** argv) {
  const int NTHR = argc > 1 ? atoi(argv[1]) : 512;
  const int NB = 256, NPT = NB * 128;
  std::vector<float> W(H * H), b(H), z0((size_t)NPT * 64, 0.f), zd0((size_t)NPT * 64, 0.f);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& w : W) w = 0.4f * rnd();
  for (auto& v : b) v = 0.2f * rnd();
  for (int p = 0; p < NPT; ++p)
    for (int pos = 0; pos < 64; ++pos)
      if (vks(pos) < KS && vfeat(pos) < H) { z0[(size_t)p * 64 + pos] = 2.f * rnd(); zd0[(size_t)p * 64 + pos] = rnd(); }
  // images: W[out][in]
  std::vector<char> img_s(IMG_SPLIT, 0);
  std::vector<float> img_f(52 * WS, 0.f), bias(64, 0.f);
  for (int pos = 0; pos < 64; ++pos) {
    if (vks(pos) >= KS || vfeat(pos) >= H) continue;
    const int fo = vfeat(pos);
    bias[pos] = b[fo];                                    // [tile][g][i] order == position order
    for (int fi = 0; fi < H; ++fi) {
      const float w = W[fo * H + fi];
      img_f[fi * WS + pos] = w;
      const int ks = fi / 4, g = fi % 4, q = ks >> 3, hh = (ks >> 2) & 1, jj = ks & 3;
      float r = w;
      for (int p = 0; p < 3; ++p) {
        const unsigned short piece = bf16_trunc(r);
        r -= bf16_val(piece);
        const int mt = pos >> 4, c = pos & 15;
        *reinterpret_cast<unsigned short*>(&img_s[((p * 2 + q) * 4 + mt) * BLK + (g * 16 + (c ^ (12 * (g & 1)))) * 16 + hh * 8 + jj * 2]) = piece;
      }
    }
  }
  char *d_is, *d_if; float *d_b, *d_z, *d_zd, *d_o;
  hipMalloc(&d_is, IMG_SPLIT); hipMalloc(&d_if, IMG_F32); hipMalloc(&d_b, 256);
  hipMalloc(&d_z, z0.size() * 4); hipMalloc(&d_zd, z0.size() * 4); hipMalloc(&d_o, (size_t)NPT * 128 * 4);
  hipMemcpy(d_is, img_s.data(), IMG_SPLIT, hipMemcpyHostToDevice);
  hipMemcpy(d_if, img_f.data(), IMG_F32, hipMemcpyHostToDevice);
  hipMemcpy(d_b, bias.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(d_z, z0.data(), z0.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_zd, zd0.data(), z0.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> o((size_t)NPT * 128);
  // ---- numerics of ONE layer against fp64
  for (int form = 0; form < 3; ++form) {
    if (form == 2) kern<2><<<NB, NTHR, IMG_SPLIT + 256>>>(d_is, d_b, d_z, d_zd, d_o, 1);
    else if (form) kern<1><<<NB, NTHR, IMG_SPLIT + 256>>>(d_is, d_b, d_z, d_zd, d_o, 1);
    else kern<0><<<NB, NTHR, IMG_F32 + 256>>>(d_if, d_b, d_z, d_zd, d_o, 1);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
    double ev = 0, et = 0, sv = 0, st = 0;
    for (int p = 0; p < 4096; ++p) {
      double a[H], q[H];
      for (int fi = 0; fi < H; ++fi) {
        const int pos = vpos(fi / 4, fi % 4);
        const double s = 1.0 / (1.0 + exp(-(double)z0[(size_t)p * 64 + pos]));
        a[fi] = s; q[fi] = s * (1 - s) * zd0[(size_t)p * 64 + pos];
      }
      for (int fo = 0; fo < (form ? H : 48); ++fo) {
        double v = b[fo], t = 0;
        for (int fi = 0; fi < H; ++fi) { v += (double)W[fo * H + fi] * a[fi]; t += (double)W[fo * H + fi] * q[fi]; }
        const int pos = vpos(fo / 4, fo % 4);
        ev = fmax(ev, fabs(o[(size_t)p * 128 + pos] - v)); sv = fmax(sv, fabs(v));
        et = fmax(et, fabs(o[(size_t)p * 128 + 64 + pos] - t)); st = fmax(st, fabs(t));
      }
    }
    printf("%s one layer vs fp64: value max err %.3e (max |z| %.2f)   tangent max err %.3e (max %.2f)\n",
           form == 2 ? "PIPE " : form ? "SPLIT" : "F32  ", ev, sv, et, st);
  }
  // ---- timing
  const int iters = 2000;
  for (int form = 0; form < 3; ++form) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (form == 2) kern<2><<<NB, NTHR, IMG_SPLIT + 256>>>(d_is, d_b, d_z, d_zd, d_o, iters);
      else if (form) kern<1><<<NB, NTHR, IMG_SPLIT + 256>>>(d_is, d_b, d_z, d_zd, d_o, iters);
      else kern<0><<<NB, NTHR, IMG_F32 + 256>>>(d_if, d_b, d_z, d_zd, d_o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("VAR %d threads %d: ", VAR, NTHR);
    printf("%s %d layers on 256 workgroups x 128 points: %.3f ms -> %.2f us per layer-tile (%.0f cycles at 2.4 GHz)\n",
           form == 2 ? "PIPE " : form ? "SPLIT" : "F32  ", iters, ms, ms * 1e3 / iters, ms * 1e-3 / iters * 2.4e9);
  }
  return 0;
}
