// One-thread-per-point evaluation kernels (float and double): model value, and the strong
// PDE residual with the Laplacian obtained by second-order forward-mode propagation.
// Not on the training hot path: they serve `evaluate` / `residual`
// (VarNet.py:1510-1692; TFModel.py:536-545, 743-754), which the reference runs every
// `saveFreq` epochs on `uniform_input`.  The fp64 form backs BASELINE config 5.
#include "vn_internal.h"

namespace {

constexpr int PT = 64;   // threads per workgroup
constexpr int HM = VN_KMAX_WIDTH > VN_KMAX_DIN ? VN_KMAX_WIDTH : VN_KMAX_DIN;   // per-thread arrays: kernel range only

template <typename T> __device__ __forceinline__ T sig(T z, int act);
template <> __device__ __forceinline__ float sig<float>(float z, int act) {
  return act == VN_ACT_TANH ? tanhf(z) : 1.0f / (1.0f + expf(-z));
}
template <> __device__ __forceinline__ double sig<double>(double z, int act) {
  return act == VN_ACT_TANH ? tanh(z) : 1.0 / (1.0 + exp(-z));
}

template <typename T>
__global__ __launch_bounds__(PT) void vn_pw_forward(VnNet net, const T* __restrict__ theta,
                                                   const T* __restrict__ X, long n, T* __restrict__ u) {
  const long r = (long)blockIdx.x * PT + threadIdx.x;
  if (r >= n) return;
  T a[HM], z[HM];
  for (int k = 0; k < net.d_in; ++k) a[k] = X[r * net.d_in + k];
  for (int l = 1; l <= net.L; ++l) {
    const int Hin = net.H[l - 1], Hout = net.H[l];
    const T* W = theta + net.woff[l];
    const T* b = theta + net.boff[l];
    for (int j = 0; j < Hout; ++j) z[j] = b[j];
    for (int k = 0; k < Hin; ++k) {
      const T ak = a[k];
      for (int j = 0; j < Hout; ++j) z[j] += ak * W[k * Hout + j];
    }
    for (int j = 0; j < Hout; ++j) a[j] = sig<T>(z[j], net.act);
  }
  const T* wo = theta + net.woff[net.L + 1];
  T acc = theta[net.boff[net.L + 1]];
  for (int k = 0; k < net.H[net.L]; ++k) acc += wo[k] * a[k];
  u[r] = acc;
}

// Derivative slots: d1[0..dim-1] = d/dx_d, d1[dim] = d/dt (if time dependent);
// d2[0..dim-1] = d^2/dx_d^2.
constexpr int DMAXS = 3;  // spatial dims supported by the residual

template <typename T>
__global__ __launch_bounds__(PT) void vn_pw_residual(VnNet net, const T* __restrict__ theta,
                                                    const T* __restrict__ X, const T* __restrict__ diff,
                                                    const T* __restrict__ vel, const T* __restrict__ src,
                                                    const T* __restrict__ ddx, int td, long n,
                                                    T* __restrict__ u, T* __restrict__ res) {
  const long r = (long)blockIdx.x * PT + threadIdx.x;
  if (r >= n) return;
  const int dim = net.dim, nd1 = dim + (td ? 1 : 0);
  T a[HM], z[HM];
  T d1[DMAXS + 1][HM], z1[DMAXS + 1][HM];
  T d2[DMAXS][HM], z2[DMAXS][HM];
  for (int k = 0; k < net.d_in; ++k) {
    a[k] = X[r * net.d_in + k];
    for (int d = 0; d < nd1; ++d) d1[d][k] = (k == d) ? T(1) : T(0);
    for (int d = 0; d < dim; ++d) d2[d][k] = T(0);
  }
  for (int l = 1; l <= net.L; ++l) {
    const int Hin = net.H[l - 1], Hout = net.H[l];
    const T* W = theta + net.woff[l];
    const T* b = theta + net.boff[l];
    for (int j = 0; j < Hout; ++j) {
      z[j] = b[j];
      for (int d = 0; d < nd1; ++d) z1[d][j] = T(0);
      for (int d = 0; d < dim; ++d) z2[d][j] = T(0);
    }
    for (int k = 0; k < Hin; ++k) {
      for (int j = 0; j < Hout; ++j) {
        const T w = W[k * Hout + j];
        z[j] += a[k] * w;
        for (int d = 0; d < nd1; ++d) z1[d][j] += d1[d][k] * w;
        for (int d = 0; d < dim; ++d) z2[d][j] += d2[d][k] * w;
      }
    }
    for (int j = 0; j < Hout; ++j) {
      const T s = sig<T>(z[j], net.act);
      const T s1 = net.act == VN_ACT_TANH ? T(1) - s * s : s * (T(1) - s);
      const T s2 = net.act == VN_ACT_TANH ? s1 * (T(-2) * s) : s1 * (T(1) - T(2) * s);
      a[j] = s;
      for (int d = 0; d < dim; ++d) d2[d][j] = s2 * z1[d][j] * z1[d][j] + s1 * z2[d][j];
      for (int d = 0; d < nd1; ++d) d1[d][j] = s1 * z1[d][j];
    }
  }
  const T* wo = theta + net.woff[net.L + 1];
  const int HL = net.H[net.L];
  T val = theta[net.boff[net.L + 1]];
  T g[DMAXS + 1] = {T(0), T(0), T(0), T(0)};
  T lap = T(0);
  for (int k = 0; k < HL; ++k) {
    val += wo[k] * a[k];
    for (int d = 0; d < nd1; ++d) g[d] += wo[k] * d1[d][k];
    for (int d = 0; d < dim; ++d) lap += wo[k] * d2[d][k];
  }
  // TFModel.py:750-754
  T out = td ? -g[dim] : T(0);
  out += diff[r] * lap;
  for (int d = 0; d < dim; ++d) {
    const T dd = ddx ? ddx[r * dim + d] : T(0);
    out -= (vel[r * dim + d] - dd) * g[d];
  }
  if (src) out += src[r];
  if (u) u[r] = val;
  res[r] = out;
}

template <typename T>
hipError_t launch_fwd(const VnNet& net, const T* theta, const T* X, long n, T* u, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const long grid = (n + PT - 1) / PT;
  hipLaunchKernelGGL(vn_pw_forward<T>, dim3((unsigned)grid), dim3(PT), 0, s, net, theta, X, n, u);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_res(const VnNet& net, const T* theta, const T* X, const T* diff, const T* vel, const T* src,
                      const T* ddx, int td, long n, T* u, T* res, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (net.dim > DMAXS) return hipErrorInvalidValue;
  const long grid = (n + PT - 1) / PT;
  hipLaunchKernelGGL(vn_pw_residual<T>, dim3((unsigned)grid), dim3(PT), 0, s, net, theta, X, diff, vel, src, ddx,
                     td, n, u, res);
  return hipGetLastError();
}

}  // namespace

hipError_t vn_pointwise_forward_f32(const VnNet& net, const float* theta, const float* X, long n, float* u,
                                    hipStream_t s) {
  return launch_fwd<float>(net, theta, X, n, u, s);
}
hipError_t vn_pointwise_forward_f64(const VnNet& net, const double* theta, const double* X, long n, double* u,
                                    hipStream_t s) {
  return launch_fwd<double>(net, theta, X, n, u, s);
}
hipError_t vn_pointwise_residual_f32(const VnNet& net, const float* theta, const float* X, const float* diff,
                                     const float* vel, const float* src, const float* ddx, int td, long n,
                                     float* u, float* res, hipStream_t s) {
  return launch_res<float>(net, theta, X, diff, vel, src, ddx, td, n, u, res, s);
}
hipError_t vn_pointwise_residual_f64(const VnNet& net, const double* theta, const double* X, const double* diff,
                                     const double* vel, const double* src, const double* ddx, int td, long n,
                                     double* u, double* res, hipStream_t s) {
  return launch_res<double>(net, theta, X, diff, vel, src, ddx, td, n, u, res, s);
}
