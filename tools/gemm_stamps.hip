// Diagnostic build of the GEMM kernel with s_memtime stamps per loop segment (never shipped):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGCT_STAMPS tools/gemm_stamps.hip -o gpurun_out/gemm_stamps
// Prints the share of wave time per segment for the FFN-1 forward shape and 4096^3.
#include "../gct_plus_amd/csrc/capi.hip"
#include "../gct_plus_amd/csrc/gemm.hip"
#include <vector>

int gct_reduce_slabs_seg(const float*, int, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { return 0; }
int gct_reduce_slabs_seg2(const float*, int, int64_t, float*, float*, float*, int64_t, int64_t, const float*, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { return 0; }
int gct_colsum(const float*, const float*, const float*, int64_t, int64_t, int, int, float*, float*, float*, float*, hipStream_t) { return 0; }
int64_t gct_colsum_ws_floats(int64_t, int64_t) { return 0; }

static void run(int64_t M, int K, int N, int extra_lds) {
  float *x, *w, *b, *y;
  unsigned long long* st;
  hipMalloc(&x, M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&y, M * N * 4);
  hipMalloc(&st, 64 * 4 * 8 * 8);
  std::vector<float> h(M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(x, h.data(), M * K * 4, hipMemcpyHostToDevice);
  h.resize((size_t)N * K);
  hipMemcpy(w, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, N * 4);
  GemmArgs g = {};
  g.M = M; g.N = N; g.K = K;
  g.a = mkseg(x, nullptr, nullptr); g.lda = K; g.a_nper = INT64_MAX / 4;
  g.b = mkseg(w, nullptr, nullptr); g.ldb = K; g.b_nper = N;
  g.c0 = y; g.ldc = N; g.c_nper = N; g.ksplit = K; g.nsplit = 1; g.epi = GCT_EPI_BIAS; g.bias0 = b;
  g.stamps = st;
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (extra_lds) hipFuncSetAttribute((const void*)gemm_f32_fast_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st, 0, 64 * 4 * 8 * 8);
    hipLaunchKernelGGL((gemm_f32_fast_kernel<true, true>), dim3((unsigned)tiles), dim3(256), extra_lds, 0, g);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> hs(64 * 4 * 8);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  double seg[8] = {0};
  for (int wv = 0; wv < 64 * 4; ++wv) for (int i = 0; i < 8; ++i) seg[i] += (double)hs[wv * 8 + i] / (64 * 4);
  double tot = 0; for (int i = 0; i < 8; ++i) tot += seg[i];
  const char* nm[8] = {"prologue", "Q0 (+gloads)", "Q1+Q2", "Q3 (+ldsw)", "barrier", "-", "epilogue", "-"};
  printf("M=%ld K=%d N=%d extra_lds=%d: wave lifetime %.0f cycles, k-tiles %d\n", (long)M, K, N, extra_lds, tot, K / 32);
  for (int i = 0; i < 7; ++i) printf("   %-12s %9.0f cyc  %5.1f %%   (%.0f per k-tile)\n", nm[i], seg[i], 100 * seg[i] / tot, seg[i] / (K / 32));
  hipFree(x); hipFree(w); hipFree(b); hipFree(y); hipFree(st);
}

int main() {
  run(40960, 512, 2048, 0);
  run(40960, 512, 2048, 40000);
  run(4096, 4096, 4096, 0);
  run(4096, 4096, 4096, 40000);
  return 0;
}
