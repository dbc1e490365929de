// GEMM laboratory (never shipped): the library's own nn.Linear entry points at the training shapes, HIP-event timed,
// built as ONE translation unit so that diagnostic -D switches reach the kernels:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DGCT_LAB_...] tools/gemm_lab.hip -o tools/_build/gemm_lab
//   ./gemm_lab [rows=40960] [reps=9] [filter]
// -DGCT_LAB_NO_EPI_MATH  : the fused epilogues store the raw accumulators (no bias / GELU / Philox / residual reads)
// -DGCT_LAB_NO_EPI_STORE : epilogue arithmetic kept, stores dropped
// -DGCT_LAB_NO_EPI       : the whole per-wave epilogue dropped
#define GCT_LAB_X6P 1
#include "../gct_plus_amd/csrc/capi.hip"
#include "../gct_plus_amd/csrc/gemm.hip"
#include "../gct_plus_amd/csrc/reduce.hip"
#include <algorithm>
#include <math.h>
#include <string>
#include <vector>

static float* dalloc(size_t n, float scale, unsigned seed) {
  float* p;
  if (hipMalloc(&p, n * 4) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 8388608.f - 1.f) * scale; }
  (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
  return p;
}

template <class F>
static double timeit(F fn, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) fn();
  (void)hipDeviceSynchronize();
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0, 0);
    fn();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e3;
}

// A/B: alternate the two variants launch by launch, median of each
template <class F>
static void timeab(F fn, int reps, double* out2) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) { fn(v); fn(v); }
  (void)hipDeviceSynchronize();
  std::vector<float> ts[2];
  for (int r = 0; r < reps; ++r)
    for (int v = 0; v < 2; ++v) {
      (void)hipEventRecord(e0, 0);
      fn(v);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      ts[v].push_back(ms);
    }
  for (int v = 0; v < 2; ++v) { std::sort(ts[v].begin(), ts[v].end()); out2[v] = ts[v][ts[v].size() / 2] * 1e3; }
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 40960;
  const int reps = argc > 2 ? atoi(argv[2]) : 9;
  const std::string filt = argc > 3 ? argv[3] : "";
  const int variant = argc > 4 ? atoi(argv[4]) : 1;      // 1: persistent stream-K launch, 2: one wave per SIMD (forward only)
  struct Shape { const char* name; int K, nper, nseg; };
  const Shape shapes[] = {{"qkv", 512, 512, 3}, {"out", 512, 512, 1}, {"ffn1", 512, 2048, 1}, {"ffn2", 2048, 512, 1}};
  const size_t maxn = (size_t)M * 2048;
  float* x = dalloc(maxn, 1.f, 1);
  float* y = dalloc(maxn, 1.f, 2);
  float* pre = dalloc(maxn, 1.f, 3);
  float* resid = dalloc(maxn, 1.f, 4);
  float* dx = dalloc(maxn, 1.f, 5);
  float* w = dalloc((size_t)2048 * 512 * 3, 0.04f, 6);
  float* dw = dalloc((size_t)2048 * 512 * 3, 0.f, 7);
  float* b = dalloc(4096, 0.1f, 8);
  float* db = dalloc(4096, 0.f, 9);
  uint16_t* wp; (void)hipMalloc(&wp, (size_t)3 * 2048 * 512 * 3 * 2);
  const int64_t wsb = 1ll << 30;
  float* ws; (void)hipMalloc(&ws, wsb);
  int32_t* sync; (void)hipMalloc(&sync, 32 * 256 * 4); (void)hipMemset(sync, 0, 32 * 256 * 4);
  gct_gemm_set_sync_buffer(sync, 32 * 256 * 4);
  float* yref = dalloc(maxn, 0.f, 11);
  float* pre_ref = dalloc(maxn, 0.f, 12);
  // max |a - b| over n floats (host side; small enough at these sizes)
  auto maxdiff = [&](const float* a, const float* bb, size_t n) {
    std::vector<float> ha(n), hb(n);
    (void)hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), bb, n * 4, hipMemcpyDeviceToHost);
    double m = 0; size_t bad = 0;
    for (size_t i = 0; i < n; ++i) { const double d = fabs((double)ha[i] - hb[i]); if (d > m) m = d; if (!(d <= 1e-3)) ++bad; }
    if (bad) printf("      !!! %zu elements differ by more than 1e-3\n", bad);
    return m;
  };
  printf("rows %ld, median of %d, mode %s\n", (long)M, reps, gct_gemm_get_mode() == GCT_GEMM_BF16X6 ? "bf16x6" : "f32");
  for (const Shape& s : shapes) {
    if (!filt.empty() && filt.find(s.name) == std::string::npos) continue;
    const int K = s.K, nper = s.nper, nseg = s.nseg, N = nper * nseg;
    const int64_t pstride = (int64_t)N * K;
    gct_split_planes(w, pstride, wp, pstride, nullptr);
    const float* w1 = nseg > 1 ? w + (size_t)nper * K : nullptr;
    const float* w2 = nseg > 2 ? w + (size_t)2 * nper * K : nullptr;
    const float* b1 = nseg > 1 ? b + nper : nullptr;
    const float* b2 = nseg > 2 ? b + 2 * nper : nullptr;
    float* y1 = nseg > 1 ? y + nper : nullptr;
    float* y2 = nseg > 2 ? y + 2 * nper : nullptr;
    const double fl = 2.0 * M * K * N;
    struct Case { const char* nm; int epi; float p; };
    const Case fc[] = {{"bias", GCT_EPI_BIAS, 0.f}, {"gelu_drop.1", GCT_EPI_GELU_DROP, 0.1f}, {"drop_resid.1", GCT_EPI_DROP_RESID, 0.1f}};
    for (const Case& c : fc) {
      if (c.epi != GCT_EPI_BIAS && nseg > 1) continue;
      double tt[2];
      (void)hipMemset(y, 0xff, (size_t)M * N * 4); (void)hipMemset(yref, 0xff, (size_t)M * N * 4);
      timeab([&](int pers) {
        gct_gemm_set_persistent(pers ? variant : 0);
        float* yy = pers ? y : yref; float* pp = pers ? pre : pre_ref;
        float* yy1 = nseg > 1 ? yy + nper : nullptr; float* yy2 = nseg > 2 ? yy + 2 * nper : nullptr;
        int rc = gct_linear_fwd_p(x, K, M, K, w, w1, w2, K, wp, pstride, b, b1, b2, nseg, nper, yy, yy1, yy2, N, c.epi, resid, pp, c.p,
                                  3, 1, ws, wsb, nullptr, nullptr);
        if (rc) { printf("fwd failed: %s\n", gct_last_error()); exit(1); }
      }, reps, tt);
      const double md = maxdiff(y, yref, (size_t)M * N);
      printf("%-5s fwd   %-13s M=%ld K=%d N=%d: %8.1f us %6.1f TF | persistent %8.1f us %6.1f TF (%+5.1f %%) maxdiff %.2e\n", s.name, c.nm,
             (long)M, K, N, tt[0], fl / tt[0] / 1e6, tt[1], fl / tt[1] / 1e6, 100 * (tt[1] / tt[0] - 1), md);
    }
    const Case dc[] = {{"store", GCT_DEPI_STORE, 0.f}, {"gelu_bwd.1", GCT_DEPI_GELU_BWD, 0.1f}};
    for (const Case& c : dc) {
      if (c.epi != GCT_DEPI_STORE && nseg > 1) continue;
      double tt[2];
      (void)hipMemset(dx, 0xff, (size_t)M * K * 4); (void)hipMemset(yref, 0xff, (size_t)M * K * 4);
      timeab([&](int pers) {
        gct_gemm_set_persistent(pers ? variant : 0);
        float* dd = pers ? dx : yref;
        int rc = gct_linear_dgrad_p(resid, nseg > 1 ? resid + nper : nullptr, nseg > 2 ? resid + 2 * nper : nullptr, N, M, nseg, nper, w, w1, w2,
                                    K, wp, pstride, K, dd, K, c.epi, pre_ref, c.p, 3, 1, ws, wsb, nullptr, 0, nullptr);
        if (rc) { printf("dgrad failed: %s\n", gct_last_error()); exit(1); }
      }, reps, tt);
      const double md = maxdiff(dx, yref, (size_t)M * K);
      printf("%-5s dgrad %-13s M=%ld K'=%d N'=%d: %8.1f us %6.1f TF | persistent %8.1f us %6.1f TF (%+5.1f %%) maxdiff %.2e\n", s.name, c.nm,
             (long)M, N, K, tt[0], fl / tt[0] / 1e6, tt[1], fl / tt[1] / 1e6, 100 * (tt[1] / tt[0] - 1), md);
    }
    {
      float* dw1 = nseg > 1 ? dw + (size_t)nper * K : nullptr;
      float* dw2 = nseg > 2 ? dw + (size_t)2 * nper * K : nullptr;
      float* db1 = nseg > 1 ? db + nper : nullptr;
      float* db2 = nseg > 2 ? db + 2 * nper : nullptr;
      const double t = timeit([&] {
        int rc = gct_linear_wgrad(resid, nseg > 1 ? resid + nper : nullptr, nseg > 2 ? resid + 2 * nper : nullptr, N, M, nseg, nper, x, K, K, dw, dw1, dw2, K, db, db1, db2, ws, nullptr);
        if (rc) { printf("wgrad failed: %s\n", gct_last_error()); exit(1); }
      }, reps);
      printf("%-5s wgrad (+reduce)       M=%ld: %8.1f us %6.1f TF\n", s.name, (long)M, t, fl / t / 1e6);
    }
    fflush(stdout);
  }
  return 0;
}
