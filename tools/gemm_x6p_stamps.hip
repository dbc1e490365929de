// Diagnostic build of the persistent plane-plane bf16x6 GEMM kernel with s_memtime stamps per segment (never shipped):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGCT_STAMPS tools/gemm_x6p_stamps.hip -o tools/_build/gemm_x6p_stamps
#include "../gct_plus_amd/csrc/capi.hip"
#include "../gct_plus_amd/csrc/gemm.hip"
#include <vector>

int gct_reduce_slabs_seg(const float*, int, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { return 0; }
int gct_reduce_slabs_seg2(const float*, int, int64_t, float*, float*, float*, int64_t, int64_t, const float*, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { return 0; }
int gct_colsum(const float*, const float*, const float*, int64_t, int64_t, int, int, float*, float*, float*, float*, hipStream_t) { return 0; }
int64_t gct_colsum_ws_floats(int64_t, int64_t) { return 0; }

static void run(int64_t M, int K, int N, int epi, int grid, int variant = 0) {
  float *x, *w, *b, *y, *pre;
  uint16_t *wp, *xp;
  unsigned long long* st;
  hipMalloc(&x, M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&y, M * N * 4); hipMalloc(&pre, M * N * 4);
  hipMalloc(&wp, (size_t)3 * N * K * 2);
  hipMalloc(&xp, (size_t)3 * M * K * 2);
  hipMalloc(&st, 64 * 8 * 8 * 8);
  std::vector<float> h(M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(x, h.data(), M * K * 4, hipMemcpyHostToDevice);
  h.resize((size_t)N * K);
  hipMemcpy(w, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, N * 4);
  gct_split_planes(w, (int64_t)N * K, wp, (int64_t)N * K, nullptr);
  gct_split_planes(x, (int64_t)M * K, xp, (int64_t)M * K, nullptr);
  GemmArgs g = {};
  g.M = M; g.N = N; g.K = K;
  g.a = mkseg(x, nullptr, nullptr); g.lda = K; g.a_nper = INT64_MAX / 4;
  g.b = mkseg(w, nullptr, nullptr); g.ldb = K; g.b_nper = N;
  g.bp0 = wp; g.bp_stride = (int64_t)N * K;
  g.ap0 = xp; g.ap_stride = (int64_t)M * K;
  g.c0 = y; g.ldc = N; g.c_nper = N; g.ksplit = K; g.nsplit = 1; g.epi = epi; g.bias0 = b; g.pre = pre; g.resid = pre;
  g.thr = gct_drop_threshold(0.1f); g.keep_scale = 1.f / 0.9f; g.rng = gct_rng_make(1, 2);
  g.stamps = st;
  g.stagger = variant;
  const int64_t tiles = ((M + 127) / 128) * ((N + 255) / 256);
  hipFuncSetAttribute((const void*)gemm_x6p_kernel<X6_FWD>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st, 0, 64 * 8 * 8 * 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((gemm_x6p_kernel<X6_FWD>), dim3((unsigned)(tiles < grid ? tiles : grid)), dim3(512), X_LDS_BYTES, 0, g);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> hs(64 * 8 * 8);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  double seg[8] = {0};
  for (int wv = 0; wv < 64 * 8; ++wv) for (int i = 0; i < 8; ++i) seg[i] += (double)hs[wv * 8 + i] / (64 * 8);
  double tot = 0; for (int i = 0; i < 8; ++i) tot += seg[i];
  const double tpw = (double)tiles / (tiles < grid ? tiles : grid);     // output tiles per workgroup
#ifdef GCT_STAMPS_EPI
  const char* nm[8] = {"tile start (wait K0)", "main loop", "tile-end sync", "next tile setup + DMA", "acc -> LDS (half 0)", "tail half 0", "acc -> LDS (half 1)", "tail half 1 + barrier"};
#else
  const char* nm[8] = {"tile start (wait K0)", "first half", "mid barrier", "second half", "tile-end sync", "epilogue", "post-epi barrier", "final store drain"};
#endif
  printf("variant %d  M=%ld K=%d N=%d epi=%d grid=%d: %.1f us, %.1f TF-eq; wave lifetime %.0f cycles, %.2f tiles per workgroup, k-tiles %d\n", variant, (long)M, K, N, epi, grid,
         ms * 1e3, 2.0 * M * K * N / (ms * 1e-3) / 1e12, tot, tpw, K / 32);
  for (int i = 0; i < 8; ++i) printf("   %-22s %9.0f cyc  %5.1f %%   (%.0f per tile, %.0f per k-tile)\n", nm[i], seg[i], 100 * seg[i] / tot, seg[i] / tpw, seg[i] / tpw / (K / 32));
  hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(pre); hipFree(wp); hipFree(xp); hipFree(st);
}

int main() {
  run(40960, 512, 1536, GCT_EPI_BIAS, 256, 0);          // 1920 tiles: 7.5 rounds
  run(40960, 512, 2048, GCT_EPI_GELU_DROP, 256, 0);     // 2560 tiles: 10 rounds
  run(40960 * 2, 512, 512, GCT_EPI_DROP_RESID, 256, 0);  // 1280 tiles: 5 rounds
  return 0;
}
