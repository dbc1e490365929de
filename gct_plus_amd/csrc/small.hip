// Tiny dense layer for the property embeddings (Model/vaetf.py:30,75-77, cvaetf.py:26,88-90):
// nn.Linear(n_c, d*n_c) with n_c = 3 -- K is far below one MFMA k-step, M = batch rows.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void small_fwd_kernel(const float* x, const float* w,
                                                        const float* b, float* y, int rows, int K,
                                                        int N) {
  const int64_t total = (int64_t)rows * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / N), n = (int)(i - (int64_t)r * N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(x[(int64_t)r * K + k], w[(int64_t)n * K + k], acc);
    y[i] = acc + (b ? b[n] : 0.f);
  }
}
// thread per (k, n), n fastest => dy reads coalesced; fixed summation order over rows
__global__ __launch_bounds__(256) void small_bwd_kernel(const float* dy, const float* x, float* dw,
                                                        float* db, int rows, int K, int N) {
  const int64_t total = (int64_t)(K + 1) * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / N), n = (int)(i - (int64_t)k * N);
    float acc = 0.f;
    if (k < K) {
      for (int r = 0; r < rows; ++r) acc = fmaf(dy[(int64_t)r * N + n], x[(int64_t)r * K + k], acc);
      dw[(int64_t)n * K + k] = acc;
    } else {
      for (int r = 0; r < rows; ++r) acc += dy[(int64_t)r * N + n];
      if (db) db[n] = acc;
    }
  }
}
}  // namespace

extern "C" int gct_small_linear_fwd(const float* x, const float* w, const float* b, float* y,
                                    int rows, int K, int N, void* stream) {
  GCT_CHECK_ARG(x && w && y && rows >= 0 && K > 0 && N > 0, "small_linear_fwd: bad args");
  if (rows == 0) return GCT_OK;
  int64_t g = ((int64_t)rows * N + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(small_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, w,
                     b, y, rows, K, N);
  GCT_LAUNCH_CHECK("small_linear_fwd");
  return GCT_OK;
}

extern "C" int gct_small_linear_bwd(const float* dy, const float* x, float* dw, float* db, int rows,
                                    int K, int N, void* stream) {
  GCT_CHECK_ARG(dy && x && dw && rows >= 0 && K > 0 && N > 0, "small_linear_bwd: bad args");
  int64_t g = ((int64_t)(K + 1) * N + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(small_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dy, x,
                     dw, db, rows, K, N);
  GCT_LAUNCH_CHECK("small_linear_bwd");
  return GCT_OK;
}
