// K7: final-vocabulary cross-entropy.
// Reference: Train/trainer1.py:21-22  F.cross_entropy(logits.view(-1,V), ys,
// ignore_index=pad, reduction='sum').  V <= 31 in practice: one wave per row (row = 120 B),
// wave-level max / sum-exp; per-block partial sums reduced in fixed order.
#include "common.h"

int gct_final_sum(const float* ws, int n, float scale, float* out, hipStream_t st);

namespace {

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ target, float* ws,
                                                     int64_t rows, int V, int64_t pad_id) {
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t t = target[row];
    if (t == pad_id) continue;  // wave-uniform
    const float* lr = logits + row * V;
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
    mx = gct_wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < V; c += 64) se += expf(lr[c] - mx);
    se = gct_wave_sum(se);
    if (lane == 0 && t >= 0 && t < V) acc += (mx + logf(se)) - lr[t];
  }
  if (lane == 0) sh[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ target,
                                                     const float* gout, float* dlogits,
                                                     int64_t rows, int V, int64_t pad_id) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float g = gout[0];
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t t = target[row];
    const float* lr = logits + row * V;
    float* dr = dlogits + row * V;
    if (t == pad_id) {
      for (int c = lane; c < V; c += 64) dr[c] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
    mx = gct_wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < V; c += 64) se += expf(lr[c] - mx);
    se = gct_wave_sum(se);
    const float inv = 1.0f / se;
    for (int c = lane; c < V; c += 64) {
      float pr = expf(lr[c] - mx) * inv;
      if (c == t) pr -= 1.0f;
      dr[c] = g * pr;
    }
  }
}

}  // namespace

extern "C" int gct_ce_fwd(const float* logits, const int64_t* target, float* out, float* ws,
                          int64_t rows, int V, int64_t pad_id, void* stream) {
  GCT_CHECK_ARG(logits && target && out && ws && rows >= 0 && V > 0, "ce_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (rows + 3) / 4;
  g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)g), dim3(256), 0, st, logits, target, ws, rows,
                     V, pad_id);
  GCT_LAUNCH_CHECK("ce_fwd");
  return gct_final_sum(ws, (int)g, 1.0f, out, st);
}

extern "C" int gct_ce_bwd(const float* logits, const int64_t* target, const float* gout,
                          float* dlogits, int64_t rows, int V, int64_t pad_id, void* stream) {
  GCT_CHECK_ARG(logits && target && gout && dlogits && rows >= 0 && V > 0, "ce_bwd: bad args");
  if (rows == 0) return GCT_OK;
  int64_t g = (rows + 3) / 4;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, logits,
                     target, gout, dlogits, rows, V, pad_id);
  GCT_LAUNCH_CHECK("ce_bwd");
  return GCT_OK;
}
