// Probe: does staggering the tile boundaries of the 256 one-per-CU workgroups hide the epilogue's HBM burst?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/burst_probe.hip -o tools/_build/burst_probe
// Every workgroup (512 threads, 144 KB of LDS so that exactly one fits a CU) repeats `iters` times:
//   compute phase : `nmfma` v_mfma_f32_16x16x32_bf16 per wave (registers only) + optionally 64 B per thread and
//                   "K-tile" of L2-resident loads (the operand stream of the real GEMM)
//   store phase   : `wkb` KB of non-temporal float4 stores per workgroup (+ optionally `rkb` KB of loads),
//                   fresh addresses every iteration
// with the workgroups started in phase (mode 0) or delayed by a share of one period according to
//   mode 1: their index inside the XCD (blockIdx >> 3), mode 2: their XCD (blockIdx & 7), mode 3: blockIdx / grid.
// Not on the product path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  float* out;            // store target, [iters][grid][wkb KB]
  const float* rd;       // epilogue read source (same layout), nullptr: none
  const float* l2buf;    // small buffer for the compute-phase operand stream (nullptr: none)
  int64_t l2_floats;
  int iters, nmfma, wkb, rkb, mode;
  int64_t period_cycles;  // one iteration's length in s_memtime cycles (for the stagger)
  unsigned long long* stamps;   // [grid][2]: cycles in compute / store phases of wave 0
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void burst_kernel(const Args a) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x;
  const unsigned nb = gridDim.x, b = blockIdx.x;
  // stagger
  double frac = 0.0;
  if (a.mode == 1) frac = (double)((b >> 3) % 32u) / 32.0;
  else if (a.mode == 2) frac = (double)(b & 7u) / 8.0;
  else if (a.mode == 3) frac = (double)b / (double)nb;
  else if (a.mode == 4) frac = (double)(((b >> 3) * 8u + (b & 7u) * 37u) % 256u) / 256.0;   // scrambled
  if (frac > 0.0) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long want = (unsigned long long)(frac * (double)a.period_cycles);
    while (__builtin_readcyclecounter() - t0 < want) __builtin_amdgcn_s_sleep(32);
  }
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)((tid * 7 + i) % 13 - 6); fb[i] = (__bf16)(float)((tid * 3 + i) % 11 - 5); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float4 sink = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned long long tc = 0, ts = 0;
  const int64_t per_wg = (int64_t)a.wkb * 256;            // floats per workgroup and iteration
  for (int it = 0; it < a.iters; ++it) {
    unsigned long long t0 = __builtin_readcyclecounter();
    // ---- compute phase: 96 MFMAs per "K-tile", one operand fetch per thread and K-tile
    const int nkt = a.nmfma / 96;
    for (int kt = 0; kt < nkt; ++kt) {
      if (a.l2buf) {
        const int64_t o = (((int64_t)b * 977 + kt * 131 + it * 17) * 512 + tid) * 16 % (a.l2_floats - 16);
        const float4 v0 = *reinterpret_cast<const float4*>(a.l2buf + (o & ~3ll));
        sink.x += v0.x; sink.y += v0.w;
      }
#pragma unroll
      for (int m = 0; m < 24; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[q], 0, 0, 0);
      }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    // ---- store phase
    float* dst = a.out + ((int64_t)it * nb + b) * per_wg;
    const float* src = a.rd ? a.rd + ((int64_t)it * nb + b) * per_wg : nullptr;
    const int nst = (int)(per_wg / 4 / 512);               // float4 stores per thread
    float4 r[16];
    if (src) {
      const int nrd = a.rkb * 256 / 4 / 512;
      for (int s = 0; s < nrd && s < 16; ++s) r[s] = *reinterpret_cast<const float4*>(src + ((int64_t)s * 512 + tid) * 4);
      for (int s = 0; s < nrd && s < 16; ++s) { sink.x += r[s].x; sink.z += r[s].z; }
    }
    for (int s = 0; s < nst; ++s) {
      f32x4 v = {acc[s & 3][0] + sink.x, acc[s & 3][1], acc[s & 3][2], acc[s & 3][3] + (float)s};
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + ((int64_t)s * 512 + tid) * 4));
    }
    __syncthreads();
    unsigned long long t2 = __builtin_readcyclecounter();
    tc += t1 - t0; ts += t2 - t1;
  }
  if (tid == 0 && a.stamps) { a.stamps[2 * b] = tc; a.stamps[2 * b + 1] = ts; }
  if (sink.x == 123.456f) lds[tid] = 1;   // keep the loads alive
}

static double run(Args a, int grid, int reps, double* comp_cyc, double* store_cyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int LDS = 144 * 1024;
  hipFuncSetAttribute((const void*)burst_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  std::vector<float> ts;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(burst_kernel, dim3(grid), dim3(512), LDS, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r) ts.push_back(ms);
  }
  std::vector<unsigned long long> st(2 * grid);
  hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, s = 0;
  for (int i = 0; i < grid; ++i) { c += (double)st[2 * i]; s += (double)st[2 * i + 1]; }
  *comp_cyc = c / grid / a.iters; *store_cyc = s / grid / a.iters;
  double best = 1e30;
  for (float t : ts) if (t < best) best = t;
  return best * 1e3;   // us
}

int main(int argc, char** argv) {
  const int grid = 256, iters = 8;
  const int nmfma = argc > 1 ? atoi(argv[1]) : 1536;      // per wave and iteration (16 K-tiles of 96)
  Args a = {};
  const int64_t out_floats = (int64_t)iters * grid * 256 * 256;      // up to 256 KB per workgroup and iteration
  hipMalloc(&a.out, out_floats * 4);
  float* rd; hipMalloc(&rd, out_floats * 4); hipMemset(rd, 0, out_floats * 4);
  float* l2; const int64_t l2f = 6 * 1024 * 1024 / 4; hipMalloc(&l2, l2f * 4); hipMemset(l2, 0, l2f * 4);
  hipMalloc(&a.stamps, 2 * grid * 8);
  a.iters = iters; a.nmfma = nmfma; a.l2_floats = l2f;
  printf("grid %d, %d iterations, %d MFMA per wave and iteration\n", grid, iters, nmfma);
  for (int stream = 0; stream < 2; ++stream) {
    a.l2buf = stream ? l2 : nullptr;
    // pure compute reference
    a.wkb = 0; a.rkb = 0; a.rd = nullptr; a.mode = 0; a.period_cycles = 0;
    double cc, sc;
    const double t_comp = run(a, grid, 3, &cc, &sc);
    printf("operand stream %s: compute only %.1f us (%.0f cycles per iteration)\n", stream ? "on " : "off", t_comp, cc);
    const int cases[3][2] = {{128, 0}, {256, 0}, {128, 128}};
    for (int ci = 0; ci < 3; ++ci) {
      a.wkb = cases[ci][0]; a.rkb = cases[ci][1]; a.rd = a.rkb ? rd : nullptr;
      double period = 0;
      for (int mode = 0; mode <= 4; ++mode) {
        a.mode = mode;
        a.period_cycles = (int64_t)period;
        const double t = run(a, grid, 3, &cc, &sc);
        if (mode == 0) period = cc + sc;
        // a staggered launch runs one period longer by construction: report per-iteration cycles as well
        printf("  write %3d KB read %3d KB mode %d: %8.1f us  (compute %7.0f + store phase %6.0f cycles per iteration; "
               "store phase = %.2f us)\n", a.wkb, a.rkb, mode, t, cc, sc, sc / (cc + sc) * (t / iters));
      }
    }
  }
  return 0;
}
