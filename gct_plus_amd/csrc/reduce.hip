// Deterministic reductions and small utility kernels (all HBM-bound).
//   gct_reduce_slabs      : dst[i] = sum_s slabs[s][i]     (split-K / partial-sum tails)
//   gct_colsum            : db[n] = sum_m dY[m][n]         (bias gradients; Linear backward)
//   gct_add, gct_copy_rows, gct_dropout_bwd
#include "common.h"

namespace {

// dst[e] = sum_s slabs[s][e].  Block = CW float4-columns x SL slab-lanes (CW*SL = 256): lane sl
// sums slabs sl, sl+SL, ... (4 independent loads in flight), the SL partials are combined
// through LDS in fixed order => deterministic.  SL is picked from n so that even a 2 KB
// destination (norm / bias partials, up to 1024 slabs) is reduced by many lanes in parallel.
template <int SL>
__device__ __forceinline__ void reduce_slabs_body(float4* red, const float* __restrict__ slabs, int nslab,
                                                  int64_t stride, float* d0, float* d1, float* d2, int64_t nper,
                                                  int64_t n4, int accumulate, unsigned bid, unsigned nblk) {
  constexpr int CW = 256 / SL;
  const int c = threadIdx.x % CW, sl = threadIdx.x / CW;
  for (int64_t base = (int64_t)bid * CW; base < n4; base += (int64_t)nblk * CW) {
    const int64_t i = base + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      const float* src = slabs + i * 4;
      int s = sl;
      for (; s + 3 * SL < nslab; s += 4 * SL) {
        const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)s * stride);
        const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)(s + SL) * stride);
        const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)(s + 2 * SL) * stride);
        const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)(s + 3 * SL) * stride);
        a.x += (v0.x + v1.x) + (v2.x + v3.x);
        a.y += (v0.y + v1.y) + (v2.y + v3.y);
        a.z += (v0.z + v1.z) + (v2.z + v3.z);
        a.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; s < nslab; s += SL) {
        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)s * stride);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    __syncthreads();
    red[threadIdx.x] = a;
    __syncthreads();
    if (sl == 0 && i < n4) {
      for (int k = 1; k < SL; ++k) {
        const float4 v = red[k * CW + c];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      const int64_t e = i * 4;
      const bool g1 = e >= nper, g2 = e >= 2 * nper;
      float* d = (g2 ? d2 : (g1 ? d1 : d0)) + (e - (g2 ? 2 * nper : (g1 ? nper : 0)));
      if (accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(d);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      *reinterpret_cast<float4*>(d) = a;
    }
  }
}

template <int SL>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs,
                                                           int nslab, int64_t stride, float* d0,
                                                           float* d1, float* d2, int64_t nper,
                                                           int64_t n4, int accumulate) {
  __shared__ float4 red[256];
  reduce_slabs_body<SL>(red, slabs, nslab, stride, d0, d1, d2, nper, n4, accumulate, blockIdx.x, gridDim.x);
}

// two reductions in one launch (the weight slabs and the bias slabs of a split-K wgrad): blocks [0, ga) take
// region A with 4 slab lanes, the rest region B with 64 (a 2 KB destination still gets many lanes)
struct ReduceRegion {
  const float* slabs;
  int64_t stride;
  float *d0, *d1, *d2;
  int64_t nper, n4;
};
__global__ __launch_bounds__(256) void reduce_slabs2_kernel(const ReduceRegion a, const ReduceRegion b, int nslab,
                                                            unsigned ga) {
  __shared__ float4 red[256];
  if (blockIdx.x < ga)
    reduce_slabs_body<4>(red, a.slabs, nslab, a.stride, a.d0, a.d1, a.d2, a.nper, a.n4, 0, blockIdx.x, ga);
  else
    reduce_slabs_body<64>(red, b.slabs, nslab, b.stride, b.d0, b.d1, b.d2, b.nper, b.n4, 0, blockIdx.x - ga,
                          gridDim.x - ga);
}

// Deferred reductions.  A layer's backward pass ends 5-7 weight-gradient GEMMs and 2-3 Norm backward kernels with a slab
// reduction each (74 + 32 launches of 5-10 us per step, each at 0.43 of the HBM rate because it is all ramp and tail).
// Between gct_reduce_defer_begin() and gct_reduce_defer_end() every float4-shaped reduction of this file is recorded
// instead of launched, and gct_reduce_defer_flush() runs all recorded ones as ONE launch (same lanes, same summation
// order per region: bit-identical results).  The caller keeps every recorded job's slabs intact until the flush
// (gct_plus_amd/ops.py hands out distinct workspace regions while deferral is on).  Thread-local: the recording thread
// is the one that flushes (autograd's backward thread).
struct ReduceJob {
  const float* slabs;
  float *d0, *d1, *d2;
  int64_t stride, nper, n4;
  int32_t nslab, sl;          // sl: slab lanes of the region (4 / 16 / 64)
};
constexpr int RJ_MAX = 24;    // jobs per launch (kernel-argument space)
struct ReduceJobs {
  ReduceJob j[RJ_MAX];
  uint32_t first[RJ_MAX + 1]; // first block of each job
  int32_t n;
};
__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceJobs a) {
  __shared__ float4 red[256];
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < a.n; ++i)
    if (blockIdx.x >= a.first[i]) k = i;
  const ReduceJob& j = a.j[k];
  const unsigned bid = blockIdx.x - a.first[k], nblk = a.first[k + 1] - a.first[k];
  if (j.sl == 4) reduce_slabs_body<4>(red, j.slabs, j.nslab, j.stride, j.d0, j.d1, j.d2, j.nper, j.n4, 0, bid, nblk);
  else if (j.sl == 16) reduce_slabs_body<16>(red, j.slabs, j.nslab, j.stride, j.d0, j.d1, j.d2, j.nper, j.n4, 0, bid, nblk);
  else reduce_slabs_body<64>(red, j.slabs, j.nslab, j.stride, j.d0, j.d1, j.d2, j.nper, j.n4, 0, bid, nblk);
}

thread_local bool t_defer = false;
thread_local int t_njobs = 0;
thread_local ReduceJob t_jobs[4 * RJ_MAX];

inline unsigned rj_blocks(const ReduceJob& j) {
  const int cw = 256 / j.sl;
  int64_t g = (j.n4 + cw - 1) / cw;
  const int64_t cap = j.sl == 4 ? 8192 : 1024;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
int flush_jobs(hipStream_t st) {
  for (int base = 0; base < t_njobs; base += RJ_MAX) {
    ReduceJobs a;
    a.n = t_njobs - base < RJ_MAX ? t_njobs - base : RJ_MAX;
    unsigned nb = 0;
    for (int i = 0; i < a.n; ++i) {
      a.j[i] = t_jobs[base + i];
      a.first[i] = nb;
      nb += rj_blocks(a.j[i]);
    }
    for (int i = a.n; i <= RJ_MAX; ++i) a.first[i] = nb;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3(nb), dim3(256), 0, st, a);
  }
  t_njobs = 0;
  GCT_LAUNCH_CHECK("reduce_multi");
  return GCT_OK;
}
// records the job (true) or tells the caller to launch it now (deferral off / table full)
inline bool defer_job(const float* slabs, int nslab, int64_t stride, float* d0, float* d1, float* d2, int64_t nper,
                      int64_t n4, int sl) {
  if (!t_defer || t_njobs >= 4 * RJ_MAX) return false;
  t_jobs[t_njobs++] = ReduceJob{slabs, d0, d1, d2, stride, nper, n4, nslab, sl};
  return true;
}

static int launch_reduce(const float* slabs, int nslab, int64_t stride, float* d0, float* d1,
                         float* d2, int64_t nper, int64_t n4, int accumulate, hipStream_t st) {
  auto grid = [&](int cw) {
    int64_t g = (n4 + cw - 1) / cw;
    return dim3((unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)));
  };
  const int sl = (n4 >= 16384 || nslab <= 4) ? 4 : ((n4 >= 1024 || nslab <= 16) ? 16 : 64);
  if (!accumulate && defer_job(slabs, nslab, stride, d0, d1, d2, nper, n4, sl)) return 0;
  if (accumulate && t_njobs > 0) flush_jobs(st);     // an accumulation may target a recorded destination: keep the order
  if (n4 >= 16384 || nslab <= 4)
    hipLaunchKernelGGL(reduce_slabs_kernel<4>, grid(64), dim3(256), 0, st, slabs, nslab, stride, d0,
                       d1, d2, nper, n4, accumulate);
  else if (n4 >= 1024 || nslab <= 16)
    hipLaunchKernelGGL(reduce_slabs_kernel<16>, grid(16), dim3(256), 0, st, slabs, nslab, stride,
                       d0, d1, d2, nper, n4, accumulate);
  else
    hipLaunchKernelGGL(reduce_slabs_kernel<64>, grid(4), dim3(256), 0, st, slabs, nslab, stride, d0,
                       d1, d2, nper, n4, accumulate);
  return 0;
}

__global__ __launch_bounds__(256) void reduce_slabs_scalar_kernel(const float* __restrict__ slabs,
                                                                  int nslab, int64_t stride,
                                                                  float* d0, float* d1, float* d2,
                                                                  int64_t nper, int64_t n,
                                                                  int accumulate) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    float a = slabs[e];
    for (int s = 1; s < nslab; ++s) a += slabs[(int64_t)s * stride + e];
    const int q = (int)(e >= nper) + (int)(e >= 2 * nper);
    float* d = (q == 0 ? d0 : (q == 1 ? d1 : d2)) + (e - q * nper);
    *d = accumulate ? *d + a : a;
  }
}

// partial[chunk][n] = sum over the chunk's rows of y_seg(n)[m][n % nper]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* y0, const float* y1,
                                                             const float* y2, int64_t ld,
                                                             int64_t M, int64_t nper, int64_t N,
                                                             int64_t rows_per_chunk,
                                                             float* partial) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int q = (int)(n >= nper) + (int)(n >= 2 * nper);
  const float* y = (q == 0 ? y0 : (q == 1 ? y1 : y2)) + (n - q * nper);
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < M ? r0 + rows_per_chunk : M;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int64_t r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += y[r * ld];
    a1 += y[(r + 1) * ld];
    a2 += y[(r + 2) * ld];
    a3 += y[(r + 3) * ld];
  }
  for (; r < r1; ++r) a0 += y[r * ld];
  partial[(int64_t)blockIdx.y * N + n] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* y,
                                                  int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      const float4 u = *reinterpret_cast<const float4*>(a + i);
      const float4 v = *reinterpret_cast<const float4*>(b + i);
      *reinterpret_cast<float4*>(y + i) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    } else {
      for (int64_t j = i; j < n; ++j) y[j] = a[j] + b[j];
    }
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* src, int64_t src_rpb,
                                                        int64_t src_off, float* dst,
                                                        int64_t dst_rpb, int64_t dst_off,
                                                        int64_t rows, int64_t rpb, int cols,
                                                        int accumulate) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i - r * cols;
    const int64_t b = r / rpb, l = r - b * rpb;
    const float v = src[(b * src_rpb + src_off + l) * cols + c];
    float* d = dst + (b * dst_rpb + dst_off + l) * cols + c;
    *d = accumulate ? *d + v : v;
  }
}

// dy[row][col] = keep(row,col) ? dout*scale : 0 ; thread = one column x 4 rows
// qmap (nullable): the rows are a quad compaction (csrc/liverows.hip) -- compact quad gq carries the dropout
// coordinates of original quad qmap[gq] (negative: padding, zeros)
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* dout, float* dy,
                                                          int64_t rows, int cols, uint32_t thr,
                                                          float scale, GctRng rng, const int32_t* __restrict__ qmap) {
  const int64_t ngroups = (rows + 3) / 4;
  const int64_t total = ngroups * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t gq = i / cols;
    const int col = (int)(i - gq * cols);
    const int64_t oq = qmap ? (int64_t)qmap[gq] : gq;
    const uint4 bits = gct_drop_bits(rng, (uint32_t)(oq < 0 ? 0 : oq), (uint32_t)col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = gq * 4 + e;
      if (row < rows) {
        const int64_t o = row * cols + col;
        dy[o] = gct_drop_keep(bits, e, (uint32_t)col, thr) ? dout[o] * scale : 0.f;
      }
    }
  }
}

// flags[t] = 1 iff rows [32 t, 32 t + 32) of x[rows][cols] hold a non-zero element
__global__ __launch_bounds__(256) void row_tile_flags_kernel(const float* __restrict__ x, int64_t ld, int64_t rows,
                                                             int cols, uint8_t* __restrict__ flags) {
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int c4 = cols / 4;
  int nz = 0;
  for (int i = threadIdx.x; i < 32 * c4; i += 256) {
    const int r = i / c4, c = i - r * c4;
    if (r0 + r < rows) {
      const float4 v = *reinterpret_cast<const float4*>(x + (r0 + r) * ld + c * 4);
      nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
  }
  nz = __syncthreads_or(nz);
  if (threadIdx.x == 0) flags[blockIdx.x] = nz ? 1 : 0;
}

// list[0..count) = ascending indices t with flags[t] != 0 (one workgroup, deterministic order)
__global__ __launch_bounds__(1024) void compact_flags_kernel(const uint8_t* __restrict__ flags, int ntiles,
                                                             int32_t* __restrict__ list, int32_t* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t0 = 0; t0 < ntiles; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    const int f = (t < ntiles && flags[t]) ? 1 : 0;
    const unsigned long long m = __ballot(f);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (f) list[off + before] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = base;
}

inline unsigned grid_for(int64_t work_items, int block = 256, int64_t cap = 4096) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

}  // namespace

int gct_reduce_slabs_seg(const float* slabs, int nslab, int64_t stride, float* d0, float* d1,
                         float* d2, int64_t nper_elems, int64_t n, hipStream_t st);

// weight slabs (A) + bias slabs (B) of one split-K wgrad in ONE launch; falls back to two launches when a
// region is not float4-shaped
int gct_reduce_slabs_seg2(const float* sa, int nslab, int64_t stride_a, float* a0, float* a1, float* a2,
                          int64_t nper_a, int64_t na, const float* sb, int64_t stride_b, float* b0, float* b1,
                          float* b2, int64_t nper_b, int64_t nb, hipStream_t st) {
  auto vec = [](const float* s, int64_t stride, float* d0, float* d1, float* d2, int64_t nper, int64_t n) {
    return n > 0 && (n % 4 == 0) && (nper % 4 == 0) && (stride % 4 == 0) && gct_aligned16(s) && gct_aligned16(d0) &&
           (!d1 || gct_aligned16(d1)) && (!d2 || gct_aligned16(d2));
  };
  if (!vec(sa, stride_a, a0, a1, a2, nper_a, na) || !vec(sb, stride_b, b0, b1, b2, nper_b, nb)) {
    int rc = gct_reduce_slabs_seg(sa, nslab, stride_a, a0, a1, a2, nper_a, na, st);
    if (rc) return rc;
    return gct_reduce_slabs_seg(sb, nslab, stride_b, b0, b1, b2, nper_b, nb, st);
  }
  if (t_defer && t_njobs + 2 <= 4 * RJ_MAX) {        // same lanes as the launch below: 4 for the weights, 64 for the bias
    defer_job(sa, nslab, stride_a, a0, a1, a2, nper_a, na / 4, 4);
    defer_job(sb, nslab, stride_b, b0, b1, b2, nper_b, nb / 4, 64);
    return GCT_OK;
  }
  const ReduceRegion ra = {sa, stride_a, a0, a1, a2, nper_a, na / 4}, rb = {sb, stride_b, b0, b1, b2, nper_b, nb / 4};
  int64_t ga = (ra.n4 + 63) / 64, gb = (rb.n4 + 3) / 4;
  if (ga > 8192) ga = 8192;
  if (gb > 1024) gb = 1024;
  hipLaunchKernelGGL(reduce_slabs2_kernel, dim3((unsigned)(ga + gb)), dim3(256), 0, st, ra, rb, nslab, (unsigned)ga);
  GCT_LAUNCH_CHECK("reduce_slabs2");
  return GCT_OK;
}

int gct_reduce_slabs_seg(const float* slabs, int nslab, int64_t stride, float* d0, float* d1,
                         float* d2, int64_t nper_elems, int64_t n, hipStream_t st) {
  if (n <= 0) return GCT_OK;
  const bool vec = (n % 4 == 0) && (nper_elems % 4 == 0) && (stride % 4 == 0) &&
                   gct_aligned16(slabs) && gct_aligned16(d0) && (!d1 || gct_aligned16(d1)) &&
                   (!d2 || gct_aligned16(d2));
  if (vec)
    launch_reduce(slabs, nslab, stride, d0, d1, d2, nper_elems, n / 4, 0, st);
  else
    hipLaunchKernelGGL(reduce_slabs_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, st, slabs,
                       nslab, stride, d0, d1, d2, nper_elems, n, 0);
  GCT_LAUNCH_CHECK("reduce_slabs");
  return GCT_OK;
}

static int colsum_chunks(int64_t M) {
  int64_t c = (M + 127) / 128;
  if (c > 256) c = 256;
  if (c < 1) c = 1;
  return (int)c;
}

int64_t gct_colsum_ws_floats(int64_t M, int64_t N) { return (int64_t)colsum_chunks(M) * N; }

int gct_colsum(const float* y0, const float* y1, const float* y2, int64_t ld, int64_t M, int nseg,
               int nper, float* d0, float* d1, float* d2, float* ws, hipStream_t st) {
  const int64_t N = (int64_t)nseg * nper;
  const int chunks = colsum_chunks(M);
  const int64_t rpc = (M + chunks - 1) / chunks;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)chunks);
  hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, st, y0, y1, y2, ld, M,
                     (int64_t)nper, N, rpc > 0 ? rpc : 1, ws);
  GCT_LAUNCH_CHECK("colsum_partial");
  // never deferred: linear_wgrad lends this pass the workspace that its GEMM overwrites right afterwards
  const bool keep = t_defer;
  t_defer = false;
  const int rc = gct_reduce_slabs_seg(ws, chunks, N, d0, d1, d2, nper, N, st);
  t_defer = keep;
  return rc;
}

extern "C" int64_t gct_rowred_ws_bytes(int64_t rows, int64_t cols) {
  return (gct_colsum_ws_floats(rows, cols) + 1024 * cols) * (int64_t)sizeof(float) + 256;
}

extern "C" int gct_reduce_slabs(const float* slabs, int nslab, int64_t stride, float* dst,
                                int64_t n, int accumulate, void* stream) {
  GCT_CHECK_ARG(slabs && dst && nslab >= 1 && n >= 0, "reduce_slabs: bad args");
  if (n == 0) return GCT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t big = INT64_MAX / 4;
  const bool vec = (n % 4 == 0) && (stride % 4 == 0) && gct_aligned16(slabs) && gct_aligned16(dst);
  if (vec)
    launch_reduce(slabs, nslab, stride, dst, dst, dst, big, n / 4, accumulate, st);
  else
    hipLaunchKernelGGL(reduce_slabs_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, st, slabs,
                       nslab, stride, dst, dst, dst, big, n, accumulate);
  GCT_LAUNCH_CHECK("reduce_slabs");
  return GCT_OK;
}

extern "C" int gct_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  GCT_CHECK_ARG(a && b && y && n >= 0, "add: bad args");
  if (n == 0) return GCT_OK;
  GCT_CHECK_ARG(gct_aligned16(a) && gct_aligned16(b) && gct_aligned16(y), "add: unaligned");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a,
                     b, y, n);
  GCT_LAUNCH_CHECK("add");
  return GCT_OK;
}

extern "C" int gct_copy_rows(const float* src, int64_t src_rpb, int64_t src_off, float* dst,
                             int64_t dst_rpb, int64_t dst_off, int64_t rows, int64_t rpb, int cols,
                             int accumulate, void* stream) {
  GCT_CHECK_ARG(src && dst && rows >= 0 && rpb > 0 && cols > 0, "copy_rows: bad args");
  if (rows == 0) return GCT_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * cols)), dim3(256), 0,
                     (hipStream_t)stream, src, src_rpb, src_off, dst, dst_rpb, dst_off, rows, rpb,
                     cols, accumulate);
  GCT_LAUNCH_CHECK("copy_rows");
  return GCT_OK;
}

extern "C" int gct_dropout_bwd(const float* dout, float* dy, int64_t rows, int cols, float p,
                               uint64_t seed, uint32_t site, const int32_t* quad_map, void* stream) {
  GCT_CHECK_ARG(dout && dy && rows >= 0 && cols > 0 && p >= 0.f && p < 1.f, "dropout_bwd: bad args");
  GCT_CHECK_ARG(!quad_map || rows % 4 == 0, "dropout_bwd: compacted rows come in quads");
  if (rows == 0) return GCT_OK;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(((rows + 3) / 4) * cols)), dim3(256), 0,
                     (hipStream_t)stream, dout, dy, rows, cols, gct_drop_threshold(p),
                     1.0f / (1.0f - p), gct_rng_make(seed, site), quad_map);
  GCT_LAUNCH_CHECK("dropout_bwd");
  return GCT_OK;
}

extern "C" int gct_nonzero_row_tiles(const float* x, int64_t ld, int64_t rows, int cols, int32_t* list,
                                     int32_t* count, uint8_t* flags_ws, void* stream) {
  GCT_CHECK_ARG(x && list && count && flags_ws && rows >= 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 &&
                    gct_aligned16(x),
                "nonzero_row_tiles: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = (rows + 31) / 32;
  GCT_CHECK_ARG(ntiles <= INT32_MAX, "nonzero_row_tiles: too many rows");
  if (ntiles > 0) {
    hipLaunchKernelGGL(row_tile_flags_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, x, ld, rows, cols, flags_ws);
    GCT_LAUNCH_CHECK("row_tile_flags");
  }
  hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)flags_ws, (int)ntiles, list,
                     count);
  GCT_LAUNCH_CHECK("compact_flags");
  return GCT_OK;
}

/* Deferred slab reductions (see reduce_multi_kernel): between begin and end every float4-shaped slab reduction issued by
 * this thread through the library (weight-gradient slabs, bias sums, Norm alpha / bias partials) is recorded instead of
 * launched; flush runs the recorded ones as one launch on `stream` and keeps recording; end flushes and stops. */
extern "C" int gct_reduce_defer_begin(void) {
  t_defer = true;
  return GCT_OK;
}
extern "C" int gct_reduce_defer_flush(void* stream) {
  if (t_njobs == 0) return GCT_OK;
  return flush_jobs((hipStream_t)stream);
}
extern "C" int gct_reduce_defer_end(void* stream) {
  t_defer = false;
  if (t_njobs == 0) return GCT_OK;
  return flush_jobs((hipStream_t)stream);
}
extern "C" int gct_reduce_defer_pending(void) { return t_njobs; }
