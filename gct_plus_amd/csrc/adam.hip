// K8: fused Adam over one flat fp32 buffer.
// Reference: torch.optim.Adam as configured in train1.py:116-119 (betas (0.9,0.98), eps 1e-9,
// no weight decay, no amsgrad); step arithmetic follows torch's single-tensor path:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= (lr/(1-b1^t)) * m/denom
// HBM-bound: 16 B read + 12 B written per parameter.
#include <math.h>

#include "common.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v,
                                                   int64_t n4, int64_t n, float b1, float b2,
                                                   float eps, float step_size, float inv_bc2_sqrt,
                                                   float gscale, const int32_t* __restrict__ skip_if_nonzero) {
  if (skip_if_nonzero && *skip_if_nonzero != 0) return;     // a gradient landed where none may (see the header)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    if (e + 3 < n) {
      float4 pp = *reinterpret_cast<float4*>(p + e);
      float4 gg = *reinterpret_cast<const float4*>(g + e);
      float4 mm = *reinterpret_cast<float4*>(m + e);
      float4 vv = *reinterpret_cast<float4*>(v + e);
      float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = G[j] * gscale;
        M[j] = b1 * M[j] + (1.0f - b1) * gr;
        V[j] = b2 * V[j] + (1.0f - b2) * gr * gr;
        P[j] -= step_size * (M[j] / (sqrtf(V[j]) * inv_bc2_sqrt + eps));
      }
      *reinterpret_cast<float4*>(p + e) = pp;
      *reinterpret_cast<float4*>(m + e) = mm;
      *reinterpret_cast<float4*>(v + e) = vv;
    } else {
      for (int64_t k = e; k < n; ++k) {
        const float gr = g[k] * gscale;
        const float mk = b1 * m[k] + (1.0f - b1) * gr;
        const float vk = b2 * v[k] + (1.0f - b2) * gr * gr;
        m[k] = mk; v[k] = vk;
        p[k] -= step_size * (mk / (sqrtf(vk) * inv_bc2_sqrt + eps));
      }
    }
  }
}
}  // namespace

extern "C" int gct_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                     float b1, float b2, float eps, int64_t step, float gscale,
                                     const int32_t* skip_if_nonzero, void* stream) {
  GCT_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "adam_step: bad args");
  GCT_CHECK_ARG(gct_aligned16(p) && gct_aligned16(g) && gct_aligned16(m) && gct_aligned16(v),
                "adam_step: buffers must be 16-B aligned");
  if (n == 0) return GCT_OK;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const int64_t n4 = (n + 3) / 4;
  int64_t grid = (n4 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     v, n4, n, b1, b2, eps, step_size, inv_bc2_sqrt, gscale, skip_if_nonzero);
  GCT_LAUNCH_CHECK("adam_step");
  return GCT_OK;
}

extern "C" int gct_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                             float b1, float b2, float eps, int64_t step, float gscale,
                             void* stream) {
  return gct_adam_step_guarded(p, g, m, v, n, lr, b1, b2, eps, step, gscale, nullptr, stream);
}
