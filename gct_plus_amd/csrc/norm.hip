// K2: the reference's Norm (Model/modules.py:80-95):
//   y = alpha * (x - mean) / (std_unbiased + eps) + bias        (eps added to the STD)
// One wave per row: a 512-wide row is 8 floats per lane (2 x 16-B loads), statistics by
// wave-level shuffles, nothing goes through LDS.  HBM-bound: fwd reads x, writes y (8 B/elem);
// bwd reads dy, x (+dres), writes dx (12-16 B/elem); dalpha/dbias leave as per-block partials
// reduced deterministically by gct_reduce_slabs.
#include "common.h"

namespace {

constexpr int MAXC = 8;  // float4 chunks per lane => d <= 8*256 = 2048

template <int NC>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ bias, float* y,
                                                       float* mean_out, float* rstd_out,
                                                       int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float4 a[NC], b[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < d) {
      a[c] = *reinterpret_cast<const float4*>(alpha + col);
      b[c] = *reinterpret_cast<const float4*>(bias + col);
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += nwaves) {
    const float* xr = x + row * d;
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < d) {
        v[c] = *reinterpret_cast<const float4*>(xr + col);
        s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
      }
    }
    const float mean = gct_wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < d) {
        v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean;
        ss += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
      }
    }
    const float var = gct_wave_sum(ss) / (float)(d - 1);
    const float rstd = 1.0f / (sqrtf(var) + eps);
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    float* yr = y + row * d;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < d) {
        float4 o;
        o.x = a[c].x * (v[c].x * rstd) + b[c].x;
        o.y = a[c].y * (v[c].y * rstd) + b[c].y;
        o.z = a[c].z * (v[c].z * rstd) + b[c].z;
        o.w = a[c].w * (v[c].w * rstd) + b[c].w;
        *reinterpret_cast<float4*>(yr + col) = o;
      }
    }
  }
}

// dx_i = r*(g_i - mean(g)) - c_i * (sum_j g_j c_j) * r^2 / (sigma*(d-1))  [+ dres_i]
// with g = dy*alpha, c = x-mean, sigma = 1/r - eps.   partial[blk][0][:] = sum dy*c*r,
// partial[blk][1][:] = sum dy.
template <int NC>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const float* __restrict__ dy,
                                                       const float* __restrict__ x,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ mean_in,
                                                       const float* __restrict__ rstd_in,
                                                       const float* __restrict__ dres, float* dx,
                                                       float* partial, int64_t rows, int d,
                                                       float eps, const int32_t* __restrict__ qmap,
                                                       int64_t src_rows, float* dropo, uint32_t thr, float dscale,
                                                       GctRng rng) {
  // dropo (nullable): a second output, dropout_bwd(dx) with the mask of the dropout the PRECEDING sub-layer applied
  // to its output (the next consumer of this gradient) -- saves that sub-layer's separate dropout_bwd pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float4 a[NC], pa[NC], pb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + lane * 4;
    a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < d) a[c] = *reinterpret_cast<const float4*>(alpha + col);
    pa[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    pb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // a wave takes whole QUADS of rows: the dropout bits of the second output cover 4 rows x 2 columns per Philox call,
  // so one set of calls serves the four rows of a quad
  const int64_t nquads = (rows + 3) >> 2;
  for (int64_t quad = (int64_t)blockIdx.x * 4 + wave; quad < nquads; quad += nwaves) {
   uint4 db[NC][2];
   if (dropo) {
     const int64_t oq = qmap ? (int64_t)qmap[quad] : quad;
#pragma unroll
     for (int c = 0; c < NC; ++c) {
       const int col = c * 256 + lane * 4;
       db[c][0] = gct_drop_bits(rng, (uint32_t)(oq < 0 ? 0 : oq), (uint32_t)col);
       db[c][1] = gct_drop_bits(rng, (uint32_t)(oq < 0 ? 0 : oq), (uint32_t)col + 2);
     }
   }
   for (int e4 = 0; e4 < 4; ++e4) {
    const int64_t row = quad * 4 + e4;
    if (row >= rows) break;
    // qmap: dy / dres / dx are quad-compacted (csrc/liverows.hip); x, mean, rstd stay in the forward's row space
    // (src_rows > 0) or are compacted the same way (src_rows == 0: the forward ran on the compact rows)
    int64_t srow = row;
    if (qmap) {
      const int q = qmap[row >> 2];
      if (src_rows > 0) srow = (int64_t)q * 4 + (row & 3);
      if (q < 0 || (src_rows > 0 && srow >= src_rows)) {            // padding row: its gradient is zero
        float* dr0 = dx + row * d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int col = c * 256 + lane * 4;
          if (col < d) {
            *reinterpret_cast<float4*>(dr0 + col) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dropo) *reinterpret_cast<float4*>(dropo + row * d + col) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        continue;
      }
    }
    const float mean = mean_in[srow], r = rstd_in[srow];
    const float* xr = x + srow * d;
    const float* gr = dy + row * d;
    float4 cx[NC], g[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      cx[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      g[c] = cx[c];
      if (col < d) {
        float4 xv = *reinterpret_cast<const float4*>(xr + col);
        const float4 dv = *reinterpret_cast<const float4*>(gr + col);
        xv.x -= mean; xv.y -= mean; xv.z -= mean; xv.w -= mean;
        cx[c] = xv;
        pa[c].x += dv.x * xv.x * r; pa[c].y += dv.y * xv.y * r;
        pa[c].z += dv.z * xv.z * r; pa[c].w += dv.w * xv.w * r;
        pb[c].x += dv.x; pb[c].y += dv.y; pb[c].z += dv.z; pb[c].w += dv.w;
        g[c] = make_float4(dv.x * a[c].x, dv.y * a[c].y, dv.z * a[c].z, dv.w * a[c].w);
        s1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
        s2 += (g[c].x * xv.x + g[c].y * xv.y) + (g[c].z * xv.z + g[c].w * xv.w);
      }
    }
    s1 = gct_wave_sum(s1) / (float)d;
    s2 = gct_wave_sum(s2);
    const float sigma = 1.0f / r - eps;
    const float k2 = sigma > 0.f ? s2 * r * r / (sigma * (float)(d - 1)) : 0.f;
    float* dr = dx + row * d;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < d) {
        float4 o;
        o.x = r * (g[c].x - s1) - cx[c].x * k2;
        o.y = r * (g[c].y - s1) - cx[c].y * k2;
        o.z = r * (g[c].z - s1) - cx[c].z * k2;
        o.w = r * (g[c].w - s1) - cx[c].w * k2;
        if (dres) {
          const float4 e = *reinterpret_cast<const float4*>(dres + row * d + col);
          o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        *reinterpret_cast<float4*>(dr + col) = o;
        if (dropo) {
          // mask coordinates are the forward's (row, col); row & 3 == e4 in both row spaces (quads are aligned)
          const int e = e4;
          const uint4 b0 = db[c][0], b1 = db[c][1];
          float4 m;
          m.x = gct_drop_keep(b0, e, (uint32_t)col, thr) ? o.x * dscale : 0.f;
          m.y = gct_drop_keep(b0, e, (uint32_t)col + 1, thr) ? o.y * dscale : 0.f;
          m.z = gct_drop_keep(b1, e, (uint32_t)col + 2, thr) ? o.z * dscale : 0.f;
          m.w = gct_drop_keep(b1, e, (uint32_t)col + 3, thr) ? o.w * dscale : 0.f;
          *reinterpret_cast<float4*>(dropo + row * d + col) = m;
        }
      }
    }
   }
  }
  // combine the 4 waves' column partials through LDS, chunk by chunk (fixed order)
  __shared__ float4 comb[4][64];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * 256 + lane * 4;
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
      comb[wave][lane] = which == 0 ? pa[c] : pb[c];
      __syncthreads();
      if (wave == 0 && col < d) {
        float4 t = comb[0][lane];
        for (int w = 1; w < 4; ++w) {
          t.x += comb[w][lane].x; t.y += comb[w][lane].y;
          t.z += comb[w][lane].z; t.w += comb[w][lane].w;
        }
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + which) * d + col) = t;
      }
    }
  }
}

inline int norm_blocks(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

int gct_reduce_slabs_seg(const float* slabs, int nslab, int64_t stride, float* d0, float* d1,
                         float* d2, int64_t nper_elems, int64_t n, hipStream_t st);

#define NORM_DISPATCH(KERNEL, ...)                                                      \
  do {                                                                                  \
    const int nc = (d + 255) / 256;                                                     \
    if (nc <= 1) hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, st, __VA_ARGS__);  \
    else if (nc <= 2) hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else if (nc <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, st, __VA_ARGS__);          \
  } while (0)

extern "C" int gct_norm_fwd(const float* x, const float* alpha, const float* bias, float* y,
                            float* mean, float* rstd, int64_t rows, int d, float eps,
                            void* stream) {
  GCT_CHECK_ARG(x && alpha && bias && y && mean && rstd, "norm_fwd: null pointer");
  GCT_CHECK_ARG(rows >= 0 && d >= 4 && d % 4 == 0 && d <= 256 * MAXC, "norm_fwd: d=%d unsupported", d);
  GCT_CHECK_ARG(gct_aligned16(x) && gct_aligned16(y) && gct_aligned16(alpha) && gct_aligned16(bias),
                "norm_fwd: pointers must be 16-B aligned");
  if (rows == 0) return GCT_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4 > 8192 ? 8192 : (rows + 3) / 4));
  NORM_DISPATCH(norm_fwd_kernel, x, alpha, bias, y, mean, rstd, rows, d, eps);
  GCT_LAUNCH_CHECK("norm_fwd");
  return GCT_OK;
}

extern "C" int gct_norm_bwd(const float* dy, const float* x, const float* alpha, const float* mean,
                            const float* rstd, const float* dres, float* dx, float* dalpha,
                            float* dbias, float* ws, int64_t rows, int d, float eps,
                            const int32_t* quad_map, int64_t src_rows, float* drop_out, float p, uint64_t seed,
                            uint32_t site, void* stream) {
  GCT_CHECK_ARG(dy && x && alpha && mean && rstd && dx && dalpha && dbias && ws,
                "norm_bwd: null pointer");
  GCT_CHECK_ARG(!drop_out || (gct_aligned16(drop_out) && p >= 0.f && p < 1.f), "norm_bwd: bad dropout output");
  GCT_CHECK_ARG(!quad_map || (rows % 4 == 0 && src_rows >= 0), "norm_bwd: compacted rows come in quads");
  GCT_CHECK_ARG(rows >= 0 && d >= 4 && d % 4 == 0 && d <= 256 * MAXC, "norm_bwd: d=%d unsupported", d);
  GCT_CHECK_ARG(gct_aligned16(dy) && gct_aligned16(x) && gct_aligned16(dx) && gct_aligned16(alpha) &&
                    gct_aligned16(ws) && (!dres || gct_aligned16(dres)),
                "norm_bwd: pointers must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = norm_blocks(rows);
  dim3 grid((unsigned)nblk);
  NORM_DISPATCH(norm_bwd_kernel, dy, x, alpha, mean, rstd, dres, dx, ws, rows, d, eps, quad_map, src_rows, drop_out,
                gct_drop_threshold(p), 1.0f / (1.0f - p), gct_rng_make(seed, site));
  GCT_LAUNCH_CHECK("norm_bwd");
  // partial layout [blk][2][d]: one slab per block, destinations dalpha | dbias
  return gct_reduce_slabs_seg(ws, nblk, (int64_t)2 * d, dalpha, dbias, nullptr, d, (int64_t)2 * d, st);
}
