// K6: VAE reparameterisation and KL term.
// Reference: Model/sublayers.py:14-20, Model/cvaetf.py:63-69 (z = eps*exp(0.5*log_var) + mu);
// Train/trainer1.py:23 (KLD = -0.5*sum(1 + lv - mu^2 - exp(lv)) over ALL elements, padded
// source positions included).  HBM-bound elementwise kernels; sums are two-stage, fixed order.
#include "common.h"

namespace {

__device__ __forceinline__ float u01(uint32_t x) {  // (0,1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* mu, const float* lv,
                                                          const float* eps_in, float* eps_out,
                                                          float* z, int64_t n, GctRng rng) {
  const int64_t n4 = (n + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float e[4];
    if (eps_in) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = (i * 4 + j < n) ? eps_in[i * 4 + j] : 0.f;
    } else {
      const uint4 r = gct_philox(rng, (uint32_t)i, (uint32_t)(i >> 32), 0x13198A2Eu, 0x03707344u);
      const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
      sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
      e[0] = r0 * c0; e[1] = r0 * s0; e[2] = r1 * c1; e[3] = r1 * s1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t k = i * 4 + j;
      if (k < n) {
        eps_out[k] = e[j];
        z[k] = e[j] * expf(0.5f * lv[k]) + mu[k];
      }
    }
  }
}

__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* dz, const float* lv,
                                                          const float* eps, const float* dmu_ext,
                                                          const float* dlv_ext, float* dmu,
                                                          float* dlv, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float g = dz[i];
    float a = g, b = 0.5f * g * eps[i] * expf(0.5f * lv[i]);
    if (dmu_ext) a += dmu_ext[i];
    if (dlv_ext) b += dlv_ext[i];
    dmu[i] = a;
    dlv[i] = b;
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = gct_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void kld_partial_kernel(const float* mu, const float* lv,
                                                          float* ws, int64_t n) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float m = mu[i], l = lv[i];
    acc += 1.0f + l - m * m - expf(l);
  }
  const float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void final_sum_kernel(const float* ws, int n, float scale,
                                                        float* out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += ws[i];
  const float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = t * scale;
}

__global__ __launch_bounds__(256) void kld_bwd_kernel(const float* mu, const float* lv,
                                                      const float* gout, float* dmu, float* dlv,
                                                      int64_t n) {
  const float g = gout[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    dmu[i] = g * mu[i];
    dlv[i] = g * 0.5f * (expf(lv[i]) - 1.0f);
  }
}

inline unsigned grid_for(int64_t n, int64_t cap) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

int gct_final_sum(const float* ws, int n, float scale, float* out, hipStream_t st) {
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, ws, n, scale, out);
  GCT_LAUNCH_CHECK("final_sum");
  return GCT_OK;
}

extern "C" int gct_reparam_fwd(const float* mu, const float* log_var, const float* eps_in,
                               float* eps_out, float* z, int64_t n, uint64_t seed, uint32_t site,
                               void* stream) {
  GCT_CHECK_ARG(mu && log_var && eps_out && z && n >= 0, "reparam_fwd: bad args");
  if (n == 0) return GCT_OK;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for((n + 3) / 4, 4096)), dim3(256), 0,
                     (hipStream_t)stream, mu, log_var, eps_in, eps_out, z, n,
                     gct_rng_make(seed, site));
  GCT_LAUNCH_CHECK("reparam_fwd");
  return GCT_OK;
}

extern "C" int gct_reparam_bwd(const float* dz, const float* log_var, const float* eps,
                               const float* dmu_ext, const float* dlv_ext, float* dmu, float* dlv,
                               int64_t n, void* stream) {
  GCT_CHECK_ARG(dz && log_var && eps && dmu && dlv && n >= 0, "reparam_bwd: bad args");
  if (n == 0) return GCT_OK;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0,
                     (hipStream_t)stream, dz, log_var, eps, dmu_ext, dlv_ext, dmu, dlv, n);
  GCT_LAUNCH_CHECK("reparam_bwd");
  return GCT_OK;
}

extern "C" int gct_kld_fwd(const float* mu, const float* log_var, float* out, float* ws, int64_t n,
                           void* stream) {
  GCT_CHECK_ARG(mu && log_var && out && ws && n >= 0, "kld_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = grid_for(n, 1024);
  hipLaunchKernelGGL(kld_partial_kernel, dim3(g), dim3(256), 0, st, mu, log_var, ws, n);
  GCT_LAUNCH_CHECK("kld_partial");
  return gct_final_sum(ws, (int)g, -0.5f, out, st);
}

extern "C" int gct_kld_bwd(const float* mu, const float* log_var, const float* gout, float* dmu,
                           float* dlv, int64_t n, void* stream) {
  GCT_CHECK_ARG(mu && log_var && gout && dmu && dlv && n >= 0, "kld_bwd: bad args");
  if (n == 0) return GCT_OK;
  hipLaunchKernelGGL(kld_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream,
                     mu, log_var, gout, dmu, dlv, n);
  GCT_LAUNCH_CHECK("kld_bwd");
  return GCT_OK;
}
