// Live rows of the decoder backward.
//
// Under the reference's loss (Train/trainer1.py:21-22: cross-entropy with ignore_index = pad, summed) the
// gradient entering the decoder is exactly zero on the rows of padded target positions -- 56 % of the rows at
// MOSES-like lengths.  A row that is zero on entry stays zero through every layer below it ONLY IF no live
// query row attends to it (then its dK / dV are zero too): every row-wise op (Norm, Linear, GELU, dropout)
// maps a zero row to a zero row, attention gives zero dQ rows for zero dO rows, but a dead row that is a
// VISIBLE KEY of a live query receives dK / dV.  With the reference's causal, right-padded trg_mask the
// property holds; model.forward(trg_mask=...) is a public contract, so it is CHECKED here on the device from
// the gradient and the mask actually used, and every shortcut built on it (weight-gradient GEMMs over live
// token tiles, the compacted decoder backward) is taken only when the check passes:
//
//   live[b,t]     = row (b,t) of g has a non-zero element
//   violation     = exists (b,i,j): live[b,i] && !live[b,j] && mask[b,i,j] != 0      (dead key seen by a live query)
//                   or a live row whose mask row is all zero next to any dead row     (uniform attention over ALL keys)
//   non-prefix    = exists (b,t):   !live[b,t] && live[b,t+1]                        (live rows are not 0..n_b-1)
//
// gct_live_rows also emits what the shortcuts consume:
//   * the list of 32-row token tiles that hold a live row (ALL tiles when the check fails, so a consumer of the
//     tile list needs no host round trip to stay exact);
//   * the COMPACTION MAP of the decoder backward.  Rows are compacted in aligned groups of 4 ("quads": rows 4q..4q+3
//     of the [B*T] row space), because every dropout site draws one Philox value per 4 rows x 2 columns -- keeping
//     quads intact keeps that cost, a kernel only needs the ORIGINAL quad index of each compact quad:
//        quad_list[i] = original quad of compact quad i (ascending; padded with -1 to a multiple of 32 quads),
//        compact row 4i + e  <->  original row 4 * quad_list[i] + e,
//        cstart[b]    = compact row of (b, t = 0); the live rows of a sample are contiguous in both spaces, so
//                       (b, t) sits at compact row cstart[b] + t for t < n_b when the live rows are the prefix 0..n_b-1.
//     About 8 % more rows than an exact row compaction at MOSES-like lengths (two partial quads per sample).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void live_flags_kernel(const float* __restrict__ g, int64_t ld, int T, int cols,
                                                         const uint8_t* __restrict__ mask, int64_t mask_sb,
                                                         int64_t mask_sq, uint8_t* __restrict__ live,
                                                         int32_t* __restrict__ n_b, int32_t* __restrict__ info) {
  extern __shared__ uint8_t lv[];                 // [T]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool vec = (cols % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) & 15u) == 0);
  const int c4 = cols / 4;
  for (int t = wave; t < T; t += 4) {             // one wave per row
    const float* row = g + ((int64_t)b * T + t) * ld;
    int nz = 0;
    if (vec) {
      for (int c = lane; c < c4; c += 64) {
        const float4 v = *reinterpret_cast<const float4*>(row + c * 4);
        nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
      }
    } else {
      for (int c = lane; c < cols; c += 64) nz |= (row[c] != 0.f);
    }
    nz = __any(nz);
    if (lane == 0) lv[t] = (uint8_t)(nz ? 1 : 0);
  }
  __syncthreads();
  int cnt = 0, viol = 0, nonpre = 0;
  for (int t = tid; t < T; t += 256) {
    cnt += lv[t];
    live[(int64_t)b * T + t] = lv[t];
    if (t + 1 < T && !lv[t] && lv[t + 1]) nonpre = 1;
  }
  if (mask == nullptr) {
    // everything visible: any dead row next to any live row is a visible dead key
    int anyl = 0, anyd = 0;
    for (int t = tid; t < T; t += 256) { anyl |= lv[t]; anyd |= !lv[t]; }
    anyl = __syncthreads_or(anyl);
    anyd = __syncthreads_or(anyd);
    viol = (anyl && anyd) ? 1 : 0;
  } else {
    const uint8_t* mb = mask + (int64_t)b * mask_sb;
    for (int idx = tid; idx < T * T; idx += 256) {
      const int i = idx / T, j = idx - i * T;
      if (lv[i] && !lv[j] && mb[(int64_t)i * mask_sq + j]) viol = 1;
    }
    // a live row that sees NO key attends uniformly to every key (masked_fill(-1e9) on the whole row), dead ones
    // included: their dV is non-zero
    int anyd = 0;
    for (int t = tid; t < T; t += 256) anyd |= !lv[t];
    anyd = __syncthreads_or(anyd);
    if (anyd)
      for (int i = tid; i < T; i += 256)
        if (lv[i]) {
          int seen = 0;
          for (int j = 0; j < T && !seen; ++j) seen = mb[(int64_t)i * mask_sq + j] != 0;
          if (!seen) viol = 1;
        }
    viol = __syncthreads_or(viol);
  }
  cnt = (int)gct_wave_sum((float)cnt);
  __shared__ int wc[4];
  if (lane == 0) wc[wave] = cnt;
  nonpre = __syncthreads_or(nonpre);
  if (tid == 0) {
    const int n = wc[0] + wc[1] + wc[2] + wc[3];
    n_b[b] = n;
    atomicAdd(&info[0], n);
    if (viol) atomicAdd(&info[1], 1);
    if (nonpre) atomicAdd(&info[2], 1);
  }
}

// one workgroup: quad_list = live quads ascending, padded with -1 to a multiple of 32; qrank[q] = rank of quad q
// among the live ones (scratch); cstart[b]; info[4] = compact rows (4 x padded quads), info[5] = live quads
__global__ __launch_bounds__(1024) void live_quads_kernel(const uint8_t* __restrict__ live, int B, int T,
                                                          int32_t* __restrict__ quad_list, int32_t* __restrict__ qrank,
                                                          int32_t* __restrict__ cstart, int32_t* __restrict__ info) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t M = (int64_t)B * T;
  const int Q = (int)((M + 3) / 4);
  if (tid == 0) base = 0;
  __syncthreads();
  for (int q0 = 0; q0 < Q; q0 += 1024) {
    const int q = q0 + tid;
    int f = 0;
    if (q < Q)
      for (int e = 0; e < 4; ++e) {
        const int64_t r = (int64_t)q * 4 + e;
        if (r < M && live[r]) f = 1;
      }
    const unsigned long long m = __ballot(f);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (q < Q) qrank[q] = off + before;        // for a dead quad: the rank the next live quad will get (unused)
    if (f) quad_list[off + before] = q;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      base += tot;
    }
    __syncthreads();
  }
  const int total = base, padded = (total + 31) & ~31;
  for (int k = total + tid; k < padded; k += 1024) quad_list[k] = -1;
  __threadfence_block();
  __syncthreads();
  for (int b = tid; b < B; b += 1024) {
    const int64_t r0 = (int64_t)b * T;
    cstart[b] = 4 * qrank[r0 >> 2] + (int)(r0 & 3);
  }
  if (tid == 0) {
    info[4] = 4 * padded;
    info[5] = total;
  }
}

// dst[i][:] = src[4 * quad_list[i / 4] + i % 4][:] (zero for padding quads / rows beyond M); float4 columns
__global__ __launch_bounds__(256) void gather_quads_kernel(const float* __restrict__ src, int64_t ld, int64_t M,
                                                           const int32_t* __restrict__ quad_list, int64_t nrows,
                                                           int c4, float* __restrict__ dst, int64_t ldd) {
  const int64_t total = nrows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / c4;
    const int c = (int)(i - row * c4);
    const int q = quad_list[row >> 2];
    const int64_t r = (int64_t)q * 4 + (row & 3);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q >= 0 && r < M) v = *reinterpret_cast<const float4*>(src + r * ld + c * 4);
    *reinterpret_cast<float4*>(dst + row * ldd + c * 4) = v;
  }
}
// dst[4 * quad_list[i / 4] + i % 4][:] = src[i][:] for the valid compact rows (dst pre-zeroed by the caller)
__global__ __launch_bounds__(256) void scatter_quads_kernel(const float* __restrict__ src, int64_t ld,
                                                            const int32_t* __restrict__ quad_list, int64_t nrows,
                                                            int c4, float* __restrict__ dst, int64_t ldd, int64_t M) {
  const int64_t total = nrows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / c4;
    const int c = (int)(i - row * c4);
    const int q = quad_list[row >> 2];
    const int64_t r = (int64_t)q * 4 + (row & 3);
    if (q >= 0 && r < M) *reinterpret_cast<float4*>(dst + r * ldd + c * 4) = *reinterpret_cast<const float4*>(src + row * ld + c * 4);
  }
}

// dst[4 * quad_list[i / 4] + i % 4][:] += src[i][:] (compact quads are disjoint: no atomics)
__global__ __launch_bounds__(256) void scatter_add_quads_kernel(const float* __restrict__ src, int64_t ld,
                                                                const int32_t* __restrict__ quad_list, int64_t nrows,
                                                                int c4, float* __restrict__ dst, int64_t ldd, int64_t M) {
  const int64_t total = nrows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / c4;
    const int c = (int)(i - row * c4);
    const int q = quad_list[row >> 2];
    const int64_t r = (int64_t)q * 4 + (row & 3);
    if (q >= 0 && r < M) {
      float4* d = reinterpret_cast<float4*>(dst + r * ldd + c * 4);
      const float4 a = *d, b = *reinterpret_cast<const float4*>(src + row * ld + c * 4);
      *d = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
  }
}

// zero the rows of a compact buffer that belong to no sample's live prefix: [cstart[b] + n_b[b], cstart[b + 1]) for every
// sample (the padded rows that travel with a live quad) and everything behind the last sample up to `nrows` (the -1
// padding quads).  A handful of rows per sample instead of a fill of the whole buffer.
// Blocks [0, B): sample b's gap.  The LAST sample's gap runs to `nrows` (the padding quads and the slack rows behind the
// compact rows: a few hundred rows, where every other gap is at most three) -- it is shared by the ZG_TAIL extra blocks
// [B, B + ZG_TAIL) instead of being one workgroup's serial loop (that block alone took 10 us per launch, 36 launches a step).
constexpr int ZG_TAIL = 48;
__global__ __launch_bounds__(256) void zero_gap_rows_kernel(float* __restrict__ buf, int64_t ld, int c4,
                                                            const int32_t* __restrict__ cstart,
                                                            const int32_t* __restrict__ n_b, int B, int64_t nrows) {
  int b = blockIdx.x;
  int64_t r0, r1;
  if (b < B - 1) {
    r0 = (int64_t)cstart[b] + n_b[b];
    r1 = (int64_t)cstart[b + 1];
  } else {
    const int part = b < B ? 0 : b - B + 1;                 // block B - 1 and the ZG_TAIL extra ones share the tail
    const int64_t t0 = (int64_t)cstart[B - 1] + n_b[B - 1];
    const int64_t len = nrows > t0 ? nrows - t0 : 0, per = (len + ZG_TAIL) / (ZG_TAIL + 1);
    r0 = t0 + part * per;
    r1 = r0 + per < nrows ? r0 + per : nrows;
  }
  for (int64_t row = r0; row < r1; ++row) {
    float* p = buf + row * ld;
    for (int c = threadIdx.x; c < c4; c += 256) *reinterpret_cast<float4*>(p + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// the same two maps for any column count / alignment (the vocabulary head's 28-31 logits per row): one element per thread
__global__ __launch_bounds__(256) void gather_quads_scalar_kernel(const float* __restrict__ src, int64_t ld, int64_t M,
                                                                  const int32_t* __restrict__ quad_list, int64_t nrows,
                                                                  int cols, float* __restrict__ dst, int64_t ldd) {
  const int64_t total = nrows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / cols;
    const int c = (int)(i - row * cols);
    const int q = quad_list[row >> 2];
    const int64_t r = (int64_t)q * 4 + (row & 3);
    dst[row * ldd + c] = (q >= 0 && r < M) ? src[r * ld + c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void scatter_quads_scalar_kernel(const float* __restrict__ src, int64_t ld,
                                                                   const int32_t* __restrict__ quad_list, int64_t nrows,
                                                                   int cols, float* __restrict__ dst, int64_t ldd,
                                                                   int64_t M) {
  const int64_t total = nrows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / cols;
    const int c = (int)(i - row * cols);
    const int q = quad_list[row >> 2];
    const int64_t r = (int64_t)q * 4 + (row & 3);
    if (q >= 0 && r < M) dst[r * ldd + c] = src[row * ld + c];
  }
}

// flags[t] = 32-row tile t holds a live row, or the dead-key check failed (then every tile is listed)
__global__ __launch_bounds__(256) void live_tile_flags_kernel(const uint8_t* __restrict__ live, int64_t rows,
                                                              const int32_t* __restrict__ info,
                                                              uint8_t* __restrict__ flags, int ntiles) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  int f = info[1] != 0;
  for (int r = 0; r < 32 && !f; ++r) {
    const int64_t row = (int64_t)t * 32 + r;
    if (row < rows && live[row]) f = 1;
  }
  flags[t] = (uint8_t)f;
}

__global__ __launch_bounds__(1024) void live_compact_tiles_kernel(const uint8_t* __restrict__ flags, int ntiles,
                                                                  int32_t* __restrict__ list, int32_t* __restrict__ count,
                                                                  int32_t* __restrict__ info) {
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
  if (threadIdx.x == 0) {
    *count = base;
    info[3] = base;
  }
}

// counter += number of rows r with live[r] == 0 that hold a non-zero element (one wave per row)
__global__ __launch_bounds__(256) void dead_rows_nonzero_kernel(const float* __restrict__ g, int64_t ld, int64_t rows,
                                                                int cols, const uint8_t* __restrict__ live,
                                                                int32_t* __restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows || live[row]) return;
  const float* p = g + row * ld;
  int nz = 0;
  for (int c = lane; c < cols; c += 64) nz |= (p[c] != 0.f);
  if (__any(nz) && lane == 0) atomicAdd(counter, 1);
}

}  // namespace

extern "C" int gct_dead_rows_nonzero(const float* g, int64_t ld, int64_t rows, int cols, const uint8_t* live,
                                     int32_t* counter, void* stream) {
  GCT_CHECK_ARG(g && live && counter && rows >= 0 && cols > 0 && ld >= cols, "dead_rows_nonzero: bad args");
  if (rows == 0) return GCT_OK;
  hipLaunchKernelGGL(dead_rows_nonzero_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, ld,
                     rows, cols, live, counter);
  GCT_LAUNCH_CHECK("dead_rows_nonzero");
  return GCT_OK;
}

extern "C" int gct_live_rows(const float* g, int64_t ld, int B, int T, int cols, const uint8_t* mask,
                             int64_t mask_sb, int64_t mask_sq, uint8_t* live, int32_t* n_b, int32_t* info,
                             int32_t* cstart, int32_t* quad_list, int32_t* qrank_ws, int32_t* tile_list,
                             int32_t* tile_count, uint8_t* tile_flags_ws, void* stream) {
  GCT_CHECK_ARG(g && live && n_b && info && B >= 0 && T > 0 && cols > 0 && ld >= cols, "live_rows: bad args");
  GCT_CHECK_ARG(T <= 4096, "live_rows: T = %d unsupported", T);
  GCT_CHECK_ARG((tile_list == nullptr) == (tile_count == nullptr) && (tile_list == nullptr || tile_flags_ws),
                "live_rows: tile_list / tile_count / tile_flags_ws go together");
  GCT_CHECK_ARG((cstart == nullptr) == (quad_list == nullptr) && (cstart == nullptr) == (qrank_ws == nullptr),
                "live_rows: cstart / quad_list / qrank_ws go together");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(info, 0, 8 * sizeof(int32_t), st);
  if (e != hipSuccess) {
    gct_set_error("live_rows: memset failed: %s", hipGetErrorString(e));
    return GCT_ERR_HIP;
  }
  if (B > 0) {
    hipLaunchKernelGGL(live_flags_kernel, dim3((unsigned)B), dim3(256), (size_t)((T + 15) & ~15), st, g, ld, T, cols,
                       mask, mask_sb, mask_sq, live, n_b, info);
    GCT_LAUNCH_CHECK("live_flags");
  }
  if (quad_list) {
    hipLaunchKernelGGL(live_quads_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)live, B, T, quad_list, qrank_ws,
                       cstart, info);
    GCT_LAUNCH_CHECK("live_quads");
  }
  if (tile_list) {
    const int64_t rows = (int64_t)B * T;
    const int ntiles = (int)((rows + 31) / 32);
    if (ntiles > 0) {
      hipLaunchKernelGGL(live_tile_flags_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, st,
                         (const uint8_t*)live, rows, (const int32_t*)info, tile_flags_ws, ntiles);
      GCT_LAUNCH_CHECK("live_tile_flags");
    }
    hipLaunchKernelGGL(live_compact_tiles_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)tile_flags_ws, ntiles,
                       tile_list, tile_count, info);
    GCT_LAUNCH_CHECK("live_compact_tiles");
  }
  return GCT_OK;
}

// Key rows of a key-padding mask [B][Lk]: live = mask != 0; n_b; info[0] live keys, info[2] samples whose visible
// keys are not the prefix 0..n_b-1, info[6] samples without a visible key.  One wave per sample.
__global__ __launch_bounds__(256) void key_flags_kernel(const uint8_t* __restrict__ mask, int64_t sb, int B, int Lk,
                                                        uint8_t* __restrict__ live, int32_t* __restrict__ n_b,
                                                        int32_t* __restrict__ info) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const uint8_t* m = mask + (int64_t)b * sb;
  int cnt = 0, nonpre = 0;
  for (int k = lane; k < Lk; k += 64) {
    const int f = m[k] != 0;
    live[(int64_t)b * Lk + k] = (uint8_t)f;
    cnt += f;
    if (k + 1 < Lk && !f && m[k + 1] != 0) nonpre = 1;
  }
  cnt = (int)gct_wave_sum((float)cnt);
  nonpre = __any(nonpre);
  if (lane == 0) {
    n_b[b] = cnt;
    atomicAdd(&info[0], cnt);
    if (nonpre) atomicAdd(&info[2], 1);
    if (cnt == 0) atomicAdd(&info[6], 1);
  }
}

extern "C" int gct_key_rows(const uint8_t* mask, int64_t mask_sb, int B, int Lk, uint8_t* live, int32_t* n_b,
                            int32_t* info, int32_t* cstart, int32_t* quad_list, int32_t* qrank_ws, void* stream) {
  GCT_CHECK_ARG(mask && live && n_b && info && cstart && quad_list && qrank_ws && B >= 0 && Lk > 0 && mask_sb >= Lk,
                "key_rows: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(info, 0, 8 * sizeof(int32_t), st);
  if (e != hipSuccess) {
    gct_set_error("key_rows: memset failed: %s", hipGetErrorString(e));
    return GCT_ERR_HIP;
  }
  if (B > 0) {
    hipLaunchKernelGGL(key_flags_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, mask, mask_sb, B, Lk, live, n_b,
                       info);
    GCT_LAUNCH_CHECK("key_flags");
  }
  hipLaunchKernelGGL(live_quads_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)live, B, Lk, quad_list, qrank_ws,
                     cstart, info);
  GCT_LAUNCH_CHECK("live_quads");
  return GCT_OK;
}

extern "C" int gct_gather_quads(const float* src, int64_t ld, int64_t M, const int32_t* quad_list, int64_t nrows,
                                int cols, float* dst, int64_t ldd, void* stream) {
  GCT_CHECK_ARG(src && quad_list && dst && nrows >= 0 && nrows % 4 == 0 && cols > 0 && ld >= cols && ldd >= cols,
                "gather_quads: bad args");
  if (nrows == 0) return GCT_OK;
  if (cols % 4 != 0 || ld % 4 != 0 || ldd % 4 != 0 || !gct_aligned16(src) || !gct_aligned16(dst)) {
    int64_t gs = (nrows * cols + 255) / 256;
    if (gs > 8192) gs = 8192;
    hipLaunchKernelGGL(gather_quads_scalar_kernel, dim3((unsigned)gs), dim3(256), 0, (hipStream_t)stream, src, ld, M,
                       quad_list, nrows, cols, dst, ldd);
    GCT_LAUNCH_CHECK("gather_quads (scalar)");
    return GCT_OK;
  }
  const int64_t work = nrows * (cols / 4);
  int64_t grid = (work + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(gather_quads_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, ld, M, quad_list,
                     nrows, cols / 4, dst, ldd);
  GCT_LAUNCH_CHECK("gather_quads");
  return GCT_OK;
}

extern "C" int gct_scatter_quads(const float* src, int64_t ld, const int32_t* quad_list, int64_t nrows, int cols,
                                 float* dst, int64_t ldd, int64_t M, void* stream) {
  GCT_CHECK_ARG(src && quad_list && dst && nrows >= 0 && nrows % 4 == 0 && cols > 0 && ld >= cols && ldd >= cols,
                "scatter_quads: bad args");
  if (nrows == 0) return GCT_OK;
  if (cols % 4 != 0 || ld % 4 != 0 || ldd % 4 != 0 || !gct_aligned16(src) || !gct_aligned16(dst)) {
    int64_t gs = (nrows * cols + 255) / 256;
    if (gs > 8192) gs = 8192;
    hipLaunchKernelGGL(scatter_quads_scalar_kernel, dim3((unsigned)gs), dim3(256), 0, (hipStream_t)stream, src, ld,
                       quad_list, nrows, cols, dst, ldd, M);
    GCT_LAUNCH_CHECK("scatter_quads (scalar)");
    return GCT_OK;
  }
  const int64_t work = nrows * (cols / 4);
  int64_t grid = (work + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(scatter_quads_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, ld, quad_list,
                     nrows, cols / 4, dst, ldd, M);
  GCT_LAUNCH_CHECK("scatter_quads");
  return GCT_OK;
}

extern "C" int gct_scatter_add_quads(const float* src, int64_t ld, const int32_t* quad_list, int64_t nrows, int cols,
                                     float* dst, int64_t ldd, int64_t M, void* stream) {
  GCT_CHECK_ARG(src && quad_list && dst && nrows >= 0 && nrows % 4 == 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 &&
                    ldd % 4 == 0 && gct_aligned16(src) && gct_aligned16(dst),
                "scatter_add_quads: bad args");
  if (nrows == 0) return GCT_OK;
  const int64_t work = nrows * (cols / 4);
  int64_t grid = (work + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(scatter_add_quads_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, ld, quad_list,
                     nrows, cols / 4, dst, ldd, M);
  GCT_LAUNCH_CHECK("scatter_add_quads");
  return GCT_OK;
}

extern "C" int gct_zero_gap_rows(float* buf, int64_t ld, int cols, const int32_t* cstart, const int32_t* n_b, int B,
                                 int64_t nrows, void* stream) {
  GCT_CHECK_ARG(buf && cstart && n_b && B >= 0 && nrows >= 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && gct_aligned16(buf),
                "zero_gap_rows: bad args");
  if (B == 0) return GCT_OK;
  hipLaunchKernelGGL(zero_gap_rows_kernel, dim3((unsigned)(B + ZG_TAIL)), dim3(256), 0, (hipStream_t)stream, buf, ld, cols / 4,
                     cstart, n_b, B, nrows);
  GCT_LAUNCH_CHECK("zero_gap_rows");
  return GCT_OK;
}
