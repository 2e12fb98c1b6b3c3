// Generic gfx950 kernels for the VarNet variational-loss step: any hidden width <= 64, any
// number of quadrature points per test function.  Activations of one 32-point tile live in
// LDS as [feature][column] matrices with 64 columns (32 value columns | 32 tangent columns);
// every layer is a v_mfma_f32_16x16x4_f32 contraction D[feat x col] = W^T[feat x k] . A[k x col].
//
// The math (value + ONE directional tangent, then its reverse pass) is stated in executable
// form in oracle/tangent_ref.py and follows the reference graph TFModel.py:536 (input gradient),
// :653-661 (weak-form integrand), :709 (parameter gradient).
#include "vn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TP = 32;        // points per tile
constexpr int NC = 64;        // columns per tile: value | tangent
constexpr int LDW = 81;       // default LDS row stride (floats): 81 = 17 mod 32 keeps both the row-wise
                              // and the transposed (weight-gradient) accesses conflict-free
constexpr int NTHREADS = 256;

// activation and its derivatives as functions of the activation value a (runtime flag: these kernels are the
// coverage path, not the fast one):  sigma' = a(1-a) | 1-a^2,  sigma''/sigma' = 1-2a | -2a
__device__ __forceinline__ float vn_act(float z, int act) {
  if (act == VN_ACT_TANH) return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * z)) - 1.0f;
  return __builtin_amdgcn_rcpf(1.0f + __expf(-z));
}
__device__ __forceinline__ float vn_d1(float a, int act) { return act == VN_ACT_TANH ? 1.f - a * a : a * (1.f - a); }
__device__ __forceinline__ float vn_d2r(float a, int act) { return act == VN_ACT_TANH ? -2.f * a : 1.f - 2.f * a; }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// D[M x 64] = sum_k a(m,k) * b(k,col).  Wave w produces column tile w (16 columns) for every
// 16-row tile of M (M <= 64).  a/b are bounds-checked by the caller-supplied accessors.
template <class AF, class BF, class SF>
__device__ __forceinline__ void gemm_cols64(int M, int K, int wave, int lane, AF a, BF b, SF st) {
  const int lm = lane & 15, lk = lane >> 4;
  const int ntm = (M + 15) >> 4;
  const int col = wave * 16 + lm;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + lk;
    const float bv = (k < K) ? b(k, col) : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ntm) {
        const int m = t * 16 + lm;
        const float av = (m < M && k < K) ? a(m, k) : 0.f;
        acc[t] = mfma16(av, bv, acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntm) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = t * 16 + 4 * lk + i;
        if (row < M) st(row, col, acc[t][i]);
      }
    }
  }
}

template <int LDS_STRIDE = LDW>
__device__ __forceinline__ void load_tile_inputs(const VnNet& net, const VnRows& sg, long r0,
                                                 float* S0, int tid) {
  const int d_in = net.d_in, dim = net.dim;
  for (int i = tid; i < d_in * NC; i += NTHREADS) {
    const int k = i / NC, c = i % NC;
    const long row = r0 + (c & (TP - 1));
    float v = 0.f;
    if (row < sg.n) {
      if (c < TP) v = sg.X[row * d_in + k];
      else if (sg.G != nullptr && k < dim) v = sg.G[row * dim + k];
    }
    S0[k * LDS_STRIDE + c] = v;
  }
}

// --------------------------------------------------------------------------------------
// forward: rows -> (u, udot)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void vn_generic_fwd_kernel(VnNet net, const float* __restrict__ theta,
                                                                  VnRows sg0, VnRows sg1, long ntiles0,
                                                                  long ntiles) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = net.hmax > net.d_in ? net.hmax : net.d_in;
  float* buf0 = lds;
  float* buf1 = lds + rows * LDW;
  const int L = net.L;

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool first = tile < ntiles0;
    const VnRows& sg = first ? sg0 : sg1;
    const long r0 = (first ? tile : tile - ntiles0) * TP;
    load_tile_inputs(net, sg, r0, buf0, tid);
    __syncthreads();
    float* cur = buf0;
    float* nxt = buf1;
    for (int l = 1; l <= L; ++l) {
      const int Hin = net.H[l - 1], Hout = net.H[l];
      const float* W = theta + net.woff[l];
      const float* bias = theta + net.boff[l];
      gemm_cols64(Hout, Hin, wave, lane,
                  [&](int m, int k) { return W[k * Hout + m]; },
                  [&](int k, int c) { return cur[k * LDW + c]; },
                  [&](int r, int c, float v) { nxt[r * LDW + c] = v; });
      __syncthreads();
      for (int i = tid; i < Hout * TP; i += NTHREADS) {
        const int m = i / TP, c = i % TP;
        const float a = vn_act(nxt[m * LDW + c] + bias[m], net.act);
        const float zd = nxt[m * LDW + TP + c];
        nxt[m * LDW + c] = a;
        nxt[m * LDW + TP + c] = vn_d1(a, net.act) * zd;
      }
      __syncthreads();
      float* t = cur; cur = nxt; nxt = t;
    }
    if (tid < NC) {
      const int HL = net.H[L];
      const float* wo = theta + net.woff[L + 1];
      float acc = 0.f;
      for (int k = 0; k < HL; ++k) acc += wo[k] * cur[k * LDW + tid];
      const long row = r0 + (tid & (TP - 1));
      if (row < sg.n) {
        if (tid < TP) sg.u[row] = acc + theta[net.boff[L + 1]];
        else if (sg.ud != nullptr) sg.ud[row] = acc;
      }
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------
// backward: rows + seeds (ubar, udbar) -> per-workgroup partial parameter gradient
// --------------------------------------------------------------------------------------
// LDW: row stride of the LDS matrices.  81 is the bank-friendly default; 65 (the minimum for 64 columns) is used
// when 81 would not fit 160 KiB (6 layers wider than 61): slower still, but the shape runs.
template <int LDW>
__global__ __launch_bounds__(NTHREADS) void vn_generic_bwd_kernel(VnNet net, const float* __restrict__ theta,
                                                                  VnRows sg0, VnRows sg1, long ntiles0,
                                                                  long ntiles, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int L = net.L;
  const int rows = net.hmax > net.d_in ? net.hmax : net.d_in;
  const int SS = rows * LDW;                 // floats per stored layer
  float* S = lds;                            // S[l] = lds + l*SS, l = 0..L
  float* T = lds + (L + 1) * SS;             // zbar | zdbar of the current layer
  float* sub = T + SS;                       // [TP] ubar
  float* sudb = sub + TP;                    // [TP] udbar

  f32x4 wacc[VN_KMAX_LAYERS][4];
  float bacc[VN_KMAX_LAYERS];
#pragma unroll
  for (int l = 0; l < VN_KMAX_LAYERS; ++l) {
    bacc[l] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) wacc[l][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float woacc = 0.f, boacc = 0.f;

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool first = tile < ntiles0;
    const VnRows& sg = first ? sg0 : sg1;
    const long r0 = (first ? tile : tile - ntiles0) * TP;
    load_tile_inputs<LDW>(net, sg, r0, S, tid);
    if (tid < TP) {
      const long row = r0 + tid;
      sub[tid] = (row < sg.n) ? sg.ubar[row] : 0.f;
      sudb[tid] = (row < sg.n && sg.udbar != nullptr) ? sg.udbar[row] : 0.f;
    }
    __syncthreads();

    // ---- forward, keeping (a_l | zdot_l) of every layer ----
    for (int l = 1; l <= L; ++l) {
      const int Hin = net.H[l - 1], Hout = net.H[l];
      const float* W = theta + net.woff[l];
      const float* bias = theta + net.boff[l];
      const float* cur = S + (l - 1) * SS;
      float* nxt = S + l * SS;
      const bool raw = (l == 1);
      gemm_cols64(Hout, Hin, wave, lane,
                  [&](int m, int k) { return W[k * Hout + m]; },
                  [&](int k, int c) {
                    const float v = cur[k * LDW + c];
                    if (raw || c < TP) return v;
                    const float a = cur[k * LDW + c - TP];
                    return vn_d1(a, net.act) * v;
                  },
                  [&](int r, int c, float v) { nxt[r * LDW + c] = v; });
      __syncthreads();
      for (int i = tid; i < Hout * TP; i += NTHREADS) {
        const int m = i / TP, c = i % TP;
        nxt[m * LDW + c] = vn_act(nxt[m * LDW + c] + bias[m], net.act);
      }
      __syncthreads();
    }

    // ---- output layer: gradients of w_o, b_o and zbar_L ----
    {
      const int HL = net.H[L];
      const float* wo = theta + net.woff[L + 1];
      const float* SL = S + L * SS;
      if (tid < HL) {
        float acc = 0.f;
        for (int c = 0; c < TP; ++c) {
          const float a = SL[tid * LDW + c], zd = SL[tid * LDW + TP + c];
          acc += sub[c] * a + sudb[c] * (vn_d1(a, net.act) * zd);
        }
        woacc += acc;
      }
      if (tid == NTHREADS - 1) {
        float acc = 0.f;
        for (int c = 0; c < TP; ++c) acc += sub[c];
        boacc += acc;
      }
      for (int i = tid; i < HL * TP; i += NTHREADS) {
        const int n = i / TP, c = i % TP;
        const float a = SL[n * LDW + c], zd = SL[n * LDW + TP + c];
        const float sp = vn_d1(a, net.act);
        const float ab = sub[c] * wo[n], adb = sudb[c] * wo[n];
        T[n * LDW + TP + c] = adb * sp;
        T[n * LDW + c] = ab * sp + adb * sp * vn_d2r(a, net.act) * zd;
      }
    }
    __syncthreads();

    // ---- hidden layers, last to first ----
#pragma unroll
    for (int l = VN_KMAX_LAYERS; l >= 1; --l) {
      if (l <= L) {
        const int Hin = net.H[l - 1], Hout = net.H[l];
        const float* W = theta + net.woff[l];
        const float* prev = S + (l - 1) * SS;
        float* dst = S + l * SS;
        const bool raw = (l == 1);
        // bias gradient: row sums of the value columns of T
        if (tid < Hout) {
          float acc = 0.f;
          for (int c = 0; c < TP; ++c) acc += T[tid * LDW + c];
          bacc[l - 1] += acc;
        }
        // weight gradient: G[k][n] += sum_col Aprev[k][col] * T[n][col]
        const int ntn = (Hout + 15) >> 4, ntm = (Hin + 15) >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = wave + 4 * j;
          if (t < ntm * ntn) {
            const int tm = t / ntn, tn = t % ntn;
            const int m = tm * 16 + lm, n = tn * 16 + lm;
            f32x4 acc = wacc[l - 1][j];
            for (int c0 = 0; c0 < NC; c0 += 4) {
              const int c = c0 + lk;
              float av = 0.f, bv = 0.f;
              if (m < Hin) {
                av = prev[m * LDW + c];
                if (!raw && c >= TP) {
                  const float a = prev[m * LDW + c - TP];
                  av = vn_d1(a, net.act) * av;
                }
              }
              if (n < Hout) bv = T[n * LDW + c];
              acc = mfma16(av, bv, acc);
            }
            wacc[l - 1][j] = acc;
          }
        }
        // input gradient of the layer: Abar_{l-1}[k][col] = sum_n W[k][n] * T[n][col]
        if (l > 1) {
          gemm_cols64(Hin, Hout, wave, lane,
                      [&](int m, int k) { return W[m * Hout + k]; },
                      [&](int k, int c) { return T[k * LDW + c]; },
                      [&](int r, int c, float v) { dst[r * LDW + c] = v; });
        }
        __syncthreads();
        if (l > 1) {
          for (int i = tid; i < Hin * TP; i += NTHREADS) {
            const int n = i / TP, c = i % TP;
            const float a = prev[n * LDW + c], zd = prev[n * LDW + TP + c];
            const float sp = vn_d1(a, net.act);
            const float ab = dst[n * LDW + c], adb = dst[n * LDW + TP + c];
            T[n * LDW + TP + c] = adb * sp;
            T[n * LDW + c] = ab * sp + adb * sp * vn_d2r(a, net.act) * zd;
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- write this workgroup's partial gradient (flat parameter layout) ----
  float* out = partial + (long)blockIdx.x * net.P;
#pragma unroll
  for (int l = 1; l <= VN_KMAX_LAYERS; ++l) {
    if (l <= L) {
      const int Hin = net.H[l - 1], Hout = net.H[l];
      const int ntn = (Hout + 15) >> 4, ntm = (Hin + 15) >> 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = wave + 4 * j;
        if (t < ntm * ntn) {
          const int tm = t / ntn, tn = t % ntn;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = tm * 16 + 4 * lk + i, col = tn * 16 + lm;
            if (row < Hin && col < Hout) out[net.woff[l] + row * Hout + col] = wacc[l - 1][j][i];
          }
        }
      }
      if (tid < Hout) out[net.boff[l] + tid] = bacc[l - 1];
    }
  }
  if (tid < net.H[L]) out[net.woff[L + 1] + tid] = woacc;
  if (tid == NTHREADS - 1) out[net.boff[L + 1]] = boacc;
}

// --------------------------------------------------------------------------------------
// weak-form epilogue: (u, udot) -> R_k, lossVec, loss partials, backward seeds
// --------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(NTHREADS) void vn_seed_kernel(VnSeedArgs a) {
  __shared__ float red[4];
  const long k = (long)blockIdx.x * NTHREADS + threadIdx.x;
  float lv = 0.f;
  if (k < a.n_k) {
    const int q = a.integ_num;
    const long base = k * q;
    float R = 0.f;
    for (int p = 0; p < q; ++p) {
      const long r = base + p;
      float t = a.ud[r];
      if (a.time_dependent) t -= a.u[r] * (a.dNtrow ? a.dNtrow[r] : a.fedNt[p]);
      if (a.source) t -= a.source[r] * (a.Nrow ? a.Nrow[r] : a.feN[p]);
      if (a.feW) t *= a.feW[p];
      R += t;
    }
    const float dj = a.detJv ? a.detJv[k] : a.detJ;
    lv = dj * R * R;                                 // detJ applied once: TFModel.py:571-577,661-668
    if (a.lossVec) a.lossVec[k] = lv;
    if (a.ubar) {
      const float s0 = 2.f * a.w2 * dj * R;
      for (int p = 0; p < q; ++p) {
        const long r = base + p;
        const float s = a.feW ? s0 * a.feW[p] : s0;
        a.udbar[r] = s;
        a.ubar[r] = a.time_dependent ? -(a.dNtrow ? a.dNtrow[r] : a.fedNt[p]) * s : 0.f;
      }
    }
  }
  float bc = 0.f, ic = 0.f;
  if (k < a.nB) {
    const float e = a.ub[k] - a.label[k];
    const float e2 = a.biDimVal * e * e;             // TFModel.py:643
    const bool isbc = k < a.bDof;
    if (isbc) bc = e2; else ic = e2;
    if (a.ubar_b) {
      const long nI = a.nB - a.bDof;
      const float cb = 2.f * a.w0 * a.biDimVal / (float)a.bDof;
      const float ci = nI > 0 ? 2.f * a.w1 * a.biDimVal / (float)nI : 0.f;
      a.ubar_b[k] = (isbc ? cb : ci) * e;
    }
  }
  const float s0 = block_sum(lv, red);
  const float s1 = block_sum(bc, red);
  const float s2 = block_sum(ic, red);
  if (threadIdx.x == 0) {
    a.part[blockIdx.x * 3 + 0] = s0;
    a.part[blockIdx.x * 3 + 1] = s1;
    a.part[blockIdx.x * 3 + 2] = s2;
  }
}

// grad[p] = sum over workgroup partials in a fixed order; block 0 also folds the loss partials.
// A block owns 64 consecutive parameters; its 16 waves each sum every 16th partial (coalesced 256-B
// rows, 16 loads in flight per lane), then the 16 sub-sums are added in wave order -> bitwise
// reproducible.
constexpr int RED_GROUPS = 16;
__global__ __launch_bounds__(64 * RED_GROUPS) void vn_reduce_kernel(const float* __restrict__ partial, int nparts,
                                                                  int P, const float* __restrict__ losspart,
                                                                  int nlp, long bDof, long nB, float w0, float w1,
                                                                  float w2, float* __restrict__ gradbuf, VnOptArgs opt) {
  __shared__ float sub[RED_GROUPS][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  // This kernel sits between two training steps whose tile loops may be shorter than its own latency (a step of the
  // [10,20,30] net is 50 us), so every global load is issued before anything waits: the optimizer state of this block's
  // parameters does not depend on the sums and is fetched first, the partials of a lane 16 at a time.
  const bool upd = grp == 0 && p < P && opt.kind >= 0;
  float th = 0.f, mo = 0.f, vo = 0.f;
  if (upd) { th = opt.theta[p]; mo = opt.m[p]; vo = opt.v[p]; }
  float acc = 0.f;
  if (p < P) {
    for (int g0 = grp; g0 < nparts; g0 += RED_GROUPS * 16) {
      float t[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int g = g0 + j * RED_GROUPS;
        t[j] = g < nparts ? partial[(long)g * P + p] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (g0 + j * RED_GROUPS < nparts) acc += t[j];          // same order as a sequential walk g = grp, grp + 16, ...
    }
  }
  sub[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && p < P) {
    float t = sub[0][lane];
#pragma unroll
    for (int j = 1; j < RED_GROUPS; ++j) t += sub[j][lane];
    gradbuf[p] = t;
    if (opt.kind == VN_OPT_ADAM) {                   // same arithmetic as vn_adam_kernel
      const float mi = opt.b1 * mo + (1.f - opt.b1) * t;
      const float vi = opt.b2 * vo + (1.f - opt.b2) * t * t;
      opt.m[p] = mi;
      opt.v[p] = vi;
      opt.theta[p] = th - opt.lr * mi / (sqrtf(vi) + opt.eps);
    } else if (opt.kind == VN_OPT_RMSPROP) {         // same arithmetic as vn_rmsprop_kernel
      const float msi = vo + (t * t - vo) * (1.f - opt.b1);
      const float mi = opt.b2 * mo + opt.lr * t / sqrtf(msi + opt.eps);
      opt.v[p] = msi;
      opt.m[p] = mi;
      opt.theta[p] = th - mi;
    }
  }
  // loss scalars: wave 1 of block 0 folds the per-workgroup partials (lane-strided, four rows per lane in flight, then a
  // fixed shuffle tree, in fp64).  The de-duplicated step hands over one row per 32 test functions (3 381 rows on the bench
  // workload: 14 dependent round trips for one wave, 21 us): beyond 1 024 rows ALL waves of block 0 but the first fold a
  // slice each and wave 1 adds the fifteen sub-sums in wave order.  Either way the order is fixed: bitwise reproducible.
  if (blockIdx.x == 0 && losspart != nullptr) {
    __shared__ double lsub[RED_GROUPS][3];
    const bool wide = nlp > 1024;
    const int nw = wide ? RED_GROUPS - 1 : 1, w = wide ? grp - 1 : 0;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (wide ? grp >= 1 : grp == 1) {
      for (int gb = w * 64 + lane; gb < nlp; gb += 64 * nw * 4) {
        float a[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int g = gb + 64 * nw * j;
#pragma unroll
          for (int c = 0; c < 3; ++c) a[j][c] = g < nlp ? losspart[g * 3 + c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { t0 += (double)a[j][0]; t1 += (double)a[j][1]; t2 += (double)a[j][2]; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        t0 += __shfl_down(t0, o, 64);
        t1 += __shfl_down(t1, o, 64);
        t2 += __shfl_down(t2, o, 64);
      }
      if (wide && lane == 0) { lsub[grp][0] = t0; lsub[grp][1] = t1; lsub[grp][2] = t2; }
    }
    if (wide) {                                       // (block-uniform: nlp is a kernel argument)
      __syncthreads();
      if (grp == 1 && lane == 0) {
        t0 = t1 = t2 = 0.0;
        for (int v = 1; v < RED_GROUPS; ++v) { t0 += lsub[v][0]; t1 += lsub[v][1]; t2 += lsub[v][2]; }
      }
    }
    if (grp == 1 && lane == 0) {
      const double var = t0;
      const double bc = bDof > 0 ? t1 / (double)bDof : 0.0;               // reduce_mean, TFModel.py:645
      const double ic = (nB - bDof) > 0 ? t2 / (double)(nB - bDof) : 0.0; // TFModel.py:648
      const float loss = (float)(w0 * bc + w1 * ic + w2 * var);           // TFModel.py:666
      gradbuf[P + 0] = loss;
      if (opt.loss_acc) opt.loss_acc[0] += loss;
      gradbuf[P + 1] = (float)bc;
      gradbuf[P + 2] = (float)ic;
      gradbuf[P + 3] = (float)var;
    }
  }
}

// TF-1 AdamOptimizer update (restated in oracle/tf1_graph.py::TF1Adam); lr_t computed on host.
__global__ __launch_bounds__(NTHREADS) void vn_adam_kernel(float* __restrict__ theta, float* __restrict__ m,
                                                           float* __restrict__ v, const float* __restrict__ g,
                                                           int P, float lr_t, float b1, float b2, float eps,
                                                           float* __restrict__ loss_acc) {
  const int p = blockIdx.x * NTHREADS + threadIdx.x;
  if (p == 0 && loss_acc) loss_acc[0] += g[P];          // epoch loss = sum of pre-update losses (g[P] = loss)
  if (p < P) {
    const float gi = g[p];
    const float mi = b1 * m[p] + (1.f - b1) * gi;
    const float vi = b2 * v[p] + (1.f - b2) * gi * gi;
    m[p] = mi;
    v[p] = vi;
    theta[p] = theta[p] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// TF-1 RMSPropOptimizer update (ApplyRMSProp, not centered): ms += (g^2 - ms)(1 - rho);
// mom = momentum*mom + lr*g/sqrt(ms + eps); theta -= mom.  Restated in oracle/tf1_graph.py::TF1RMSProp.
__global__ __launch_bounds__(NTHREADS) void vn_rmsprop_kernel(float* __restrict__ theta, float* __restrict__ mom,
                                                              float* __restrict__ ms, const float* __restrict__ g,
                                                              int P, float lr, float rho, float momentum, float eps,
                                                              float* __restrict__ loss_acc) {
  const int p = blockIdx.x * NTHREADS + threadIdx.x;
  if (p == 0 && loss_acc) loss_acc[0] += g[P];
  if (p < P) {
    const float gi = g[p];
    const float msi = ms[p] + (gi * gi - ms[p]) * (1.f - rho);
    const float mi = momentum * mom[p] + lr * gi / sqrtf(msi + eps);
    ms[p] = msi;
    mom[p] = mi;
    theta[p] = theta[p] - mi;
  }
}

}  // namespace

size_t vn_generic_fwd_lds_bytes(const VnNet& net) {
  const int rows = net.hmax > net.d_in ? net.hmax : net.d_in;
  return (size_t)2 * rows * LDW * sizeof(float);
}

static size_t bwd_lds_bytes(const VnNet& net, int ldw) {
  const int rows = net.hmax > net.d_in ? net.hmax : net.d_in;
  return ((size_t)(net.L + 2) * rows * ldw + 2 * TP) * sizeof(float);
}
static int bwd_ldw(const VnNet& net) { return bwd_lds_bytes(net, LDW) <= 160 * 1024 ? LDW : 65; }
size_t vn_generic_bwd_lds_bytes(const VnNet& net) { return bwd_lds_bytes(net, bwd_ldw(net)); }

static inline long tiles_of(long n) { return (n + TP - 1) / TP; }

hipError_t vn_generic_forward(const VnNet& net, const float* theta, VnRows seg0, VnRows seg1, int grid,
                              hipStream_t s) {
  const long nt0 = tiles_of(seg0.n), nt = nt0 + tiles_of(seg1.n);
  if (nt == 0) return hipSuccess;
  if (grid > nt) grid = (int)nt;
  const size_t lds = vn_generic_fwd_lds_bytes(net);
  hipError_t e = hipFuncSetAttribute((const void*)vn_generic_fwd_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(vn_generic_fwd_kernel, dim3(grid), dim3(NTHREADS), lds, s, net, theta, seg0, seg1, nt0, nt);
  return hipGetLastError();
}

hipError_t vn_generic_backward(const VnNet& net, const float* theta, VnRows seg0, VnRows seg1, float* partial,
                               int grid, hipStream_t s) {
  const long nt0 = tiles_of(seg0.n), nt = nt0 + tiles_of(seg1.n);
  const size_t lds = vn_generic_bwd_lds_bytes(net);
  const bool narrow = bwd_ldw(net) != LDW;
  const void* fn = narrow ? (const void*)vn_generic_bwd_kernel<65> : (const void*)vn_generic_bwd_kernel<LDW>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // every workgroup of the grid writes its partial (zeros if it owns no tile)
  if (narrow)
    hipLaunchKernelGGL(vn_generic_bwd_kernel<65>, dim3(grid), dim3(NTHREADS), lds, s, net, theta, seg0, seg1, nt0, nt, partial);
  else
    hipLaunchKernelGGL(vn_generic_bwd_kernel<LDW>, dim3(grid), dim3(NTHREADS), lds, s, net, theta, seg0, seg1, nt0, nt, partial);
  return hipGetLastError();
}

hipError_t vn_seed_launch(const VnSeedArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL(vn_seed_kernel, dim3(grid), dim3(NTHREADS), 0, s, a);
  return hipGetLastError();
}

hipError_t vn_reduce_launch(const float* partial, int nparts, int P, const float* losspart, int nlossparts,
                            long bDof, long nB, float w0, float w1, float w2, float* gradbuf, hipStream_t s,
                            VnOptArgs opt) {
  const int grid = (P + 63) / 64;
  hipLaunchKernelGGL(vn_reduce_kernel, dim3(grid > 0 ? grid : 1), dim3(64 * RED_GROUPS), 0, s, partial, nparts, P,
                     losspart, nlossparts, bDof, nB, w0, w1, w2, gradbuf, opt);
  return hipGetLastError();
}

hipError_t vn_adam_launch(float* theta, float* m, float* v, const float* grad, int P, float lr_t, float b1,
                          float b2, float eps, float* loss_acc, hipStream_t s) {
  const int grid = (P + NTHREADS - 1) / NTHREADS;
  hipLaunchKernelGGL(vn_adam_kernel, dim3(grid), dim3(NTHREADS), 0, s, theta, m, v, grad, P, lr_t, b1, b2, eps, loss_acc);
  return hipGetLastError();
}

hipError_t vn_rmsprop_launch(float* theta, float* mom, float* ms, const float* grad, int P, float lr, float rho,
                             float momentum, float eps, float* loss_acc, hipStream_t s) {
  const int grid = (P + NTHREADS - 1) / NTHREADS;
  hipLaunchKernelGGL(vn_rmsprop_kernel, dim3(grid), dim3(NTHREADS), 0, s, theta, mom, ms, grad, P, lr, rho, momentum, eps, loss_acc);
  return hipGetLastError();
}
