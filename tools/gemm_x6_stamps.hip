// Diagnostic build of the bf16x6 GEMM kernel with s_memtime stamps per segment (never shipped):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGCT_STAMPS tools/gemm_x6_stamps.hip -o tools/_build/gemm_x6_stamps
#define GCT_LAB_X6P 1
#include "../gct_plus_amd/csrc/capi.hip"
#include "../gct_plus_amd/csrc/gemm.hip"
#include <vector>

int gct_reduce_slabs_seg(const float*, int, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { return 0; }
int gct_reduce_slabs_seg2(const float*, int, int64_t, float*, float*, float*, int64_t, int64_t, const float*, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { return 0; }
int gct_colsum(const float*, const float*, const float*, int64_t, int64_t, int, int, float*, float*, float*, float*, hipStream_t) { return 0; }
int64_t gct_colsum_ws_floats(int64_t, int64_t) { return 0; }

static void run(int64_t M, int K, int N, int epi, int persistent = 0) {
  float *x, *w, *b, *y, *pre;
  uint16_t* wp;
  unsigned long long* st;
  hipMalloc(&x, M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&y, M * N * 4); hipMalloc(&pre, M * N * 4);
  hipMalloc(&wp, (size_t)3 * N * K * 2);
  hipMalloc(&st, 64 * 8 * 8 * 8);
  std::vector<float> h(M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(x, h.data(), M * K * 4, hipMemcpyHostToDevice);
  h.resize((size_t)N * K);
  hipMemcpy(w, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, N * 4);
  gct_split_planes(w, (int64_t)N * K, wp, (int64_t)N * K, nullptr);
  GemmArgs g = {};
  g.M = M; g.N = N; g.K = K;
  g.a = mkseg(x, nullptr, nullptr); g.lda = K; g.a_nper = INT64_MAX / 4;
  g.b = mkseg(w, nullptr, nullptr); g.ldb = K; g.b_nper = N;
  g.bp0 = wp; g.bp_stride = (int64_t)N * K;
  g.c0 = y; g.ldc = N; g.c_nper = N; g.ksplit = K; g.nsplit = 1; g.epi = epi; g.bias0 = b; g.pre = pre; g.resid = pre;
  g.thr = gct_drop_threshold(0.1f); g.keep_scale = 1.f / 0.9f; g.rng = gct_rng_make(1, 2);
  g.stamps = st;
  const int64_t tiles = ((M + 127) / 128) * ((N + 255) / 256);
  hipFuncSetAttribute((const void*)gemm_x6_kernel<X6_FWD>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
  if (persistent) {
    static int32_t* sync = nullptr;
    static float* wsp = nullptr;
    if (!sync) { hipMalloc(&sync, 32 * 256 * 4); hipMemset(sync, 0, 32 * 256 * 4); gct_gemm_set_sync_buffer(sync, 32 * 256 * 4); hipMalloc(&wsp, 64 << 20); }
    gct_gemm_set_persistent(1);
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(st, 0, 64 * 8 * 8 * 8);
      int taken = 0;
      launch_x6p<X6_FWD>(g, 0, wsp, 64 << 20, sync, &taken);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> hs(64 * 8 * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double seg[8] = {0};
    for (int wv = 0; wv < 64 * 8; ++wv) for (int i = 0; i < 8; ++i) seg[i] += (double)hs[wv * 8 + i] / (64 * 8);
    double tot = 0; for (int i = 0; i < 8; ++i) tot += seg[i];
    const double ntile = (double)tiles / 256.0;
    const char* nm[8] = {"steady iterations", "1st iter after end", "2nd iter after end", "3rd iter after end", "partial hand-off", "epilogue halves", "barrier + reload", "-"};
    printf("PERSISTENT M=%ld K=%d N=%d epi=%d: wave lifetime %.0f cycles, %.2f tiles per workgroup, k-tiles %d\n", (long)M, K, N, epi, tot, ntile, K / 32);
    for (int i = 0; i < 7; ++i) printf("   %-20s %9.0f cyc  %5.1f %%   (%.0f per tile)\n", nm[i], seg[i], 100 * seg[i] / tot, seg[i] / ntile);
    hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(pre); hipFree(wp); hipFree(st);
    return;
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st, 0, 64 * 8 * 8 * 8);
    hipLaunchKernelGGL((gemm_x6_kernel<X6_FWD>), dim3((unsigned)tiles), dim3(512), X_LDS_BYTES, 0, g);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> hs(64 * 8 * 8);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  double seg[8] = {0}, segg[2][8] = {{0}};          // all waves; waves 0-3 / 4-7 (the two waves of a SIMD are w and w + 4)
  for (int wv = 0; wv < 64 * 8; ++wv) for (int i = 0; i < 8; ++i) {
    seg[i] += (double)hs[wv * 8 + i] / (64 * 8);
    segg[(wv & 7) >> 2][i] += (double)hs[wv * 8 + i] / (64 * 4);
  }
  double tot = 0; for (int i = 0; i < 8; ++i) tot += seg[i];
  const char* nm[8] = {"prologue", "first half", "barrier", "second half", "pre-epilogue sync", "epilogue(+stores)", "-", "-"};
  printf("M=%ld K=%d N=%d epi=%d: wave lifetime %.0f cycles, k-tiles %d\n", (long)M, K, N, epi, tot, K / 32);
  for (int i = 0; i < 6; ++i)
    printf("   %-18s %9.0f cyc  %5.1f %%   (%.0f per k-tile; waves 0-3: %.0f, waves 4-7: %.0f)\n", nm[i], seg[i], 100 * seg[i] / tot,
           seg[i] / (K / 32), segg[0][i] / (K / 32), segg[1][i] / (K / 32));
  hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(pre); hipFree(wp); hipFree(st);
}

int main() {
  run(40960, 512, 2048, GCT_EPI_BIAS);
  run(40960, 512, 2048, GCT_EPI_GELU_DROP);
  run(40960, 2048, 512, GCT_EPI_DROP_RESID);
  run(40960, 512, 512, GCT_EPI_DROP_RESID);
  run(40960, 512, 2048, GCT_EPI_BIAS, 1);
  run(40960, 512, 2048, GCT_EPI_GELU_DROP, 1);
  run(40960, 2048, 512, GCT_EPI_DROP_RESID, 1);
  run(40960, 512, 512, GCT_EPI_DROP_RESID, 1);
  return 0;
}
