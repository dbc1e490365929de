// K1: token embedding lookup + property-token concat + positional encoding + dropout.
// Reference: Model/modules.py:108-110 (nn.Embedding), :134-144 (x*sqrt(d) + pe, dropout),
// Model/vaetf.py:35-39 / Model/cvaetf.py:38-41 (cond rows concatenated in FRONT, they get PE too).
// HBM-bound: the table (<= 31 rows x 2 KB) and pe stay in L2; traffic = the [B*L][d] output.
// Backward: per-vocab-row segmented reduction in LDS (V <= 64 rows), partial slabs per row
// chunk, deterministic slab reduction -- no atomics.
#include "common.h"

namespace {

// thread = (4 consecutive rows) x (4 consecutive columns): float4 everywhere, two Philox calls per 16 elements,
// 32-bit index arithmetic (the 64-bit divisions of the first version cost more than the memory traffic:
// 55 us for 84 MB).  Requires d % 4 == 0 and B * L * d < 2^31 (checked on the host).
__global__ __launch_bounds__(256) void embed_pe_fwd_kernel(const int64_t* __restrict__ tok,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ cond,
                                                           const float* __restrict__ pe, float* out,
                                                           int B, int S, int n_c, int d, int vocab,
                                                           float scale, uint32_t thr,
                                                           float keep_scale, GctRng rng) {
  const int L = S + n_c, rows = B * L, d4 = d >> 2;
  const int total = ((rows + 3) >> 2) * d4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int gq = i / d4, col = (i - gq * d4) * 4;
    uint4 b0 = make_uint4(~0u, ~0u, ~0u, ~0u), b1 = b0;
    if (thr) {
      b0 = gct_drop_bits(rng, (uint32_t)gq, (uint32_t)col);
      b1 = gct_drop_bits(rng, (uint32_t)gq, (uint32_t)col + 2);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = gq * 4 + e;
      if (row >= rows) break;
      const int b = row / L, l = row - b * L;
      float4 v;
      if (l < n_c) {
        v = *reinterpret_cast<const float4*>(cond + ((int64_t)b * n_c + l) * d + col);
      } else {
        int64_t t = tok[(int64_t)b * S + (l - n_c)];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);  // clamp: never fault on a bad id
        v = *reinterpret_cast<const float4*>(table + t * d + col);
      }
      const float4 q = *reinterpret_cast<const float4*>(pe + (int64_t)l * d + col);
      v.x = v.x * scale + q.x; v.y = v.y * scale + q.y; v.z = v.z * scale + q.z; v.w = v.w * scale + q.w;
      v.x = gct_drop_keep(b0, e, (uint32_t)col, thr) ? v.x * keep_scale : 0.f;
      v.y = gct_drop_keep(b0, e, (uint32_t)col + 1, thr) ? v.y * keep_scale : 0.f;
      v.z = gct_drop_keep(b1, e, (uint32_t)col + 2, thr) ? v.z * keep_scale : 0.f;
      v.w = gct_drop_keep(b1, e, (uint32_t)col + 3, thr) ? v.w * keep_scale : 0.f;
      *reinterpret_cast<float4*>(out + (int64_t)row * d + col) = v;
    }
  }
}

// grid = (ceil(d/256), chunks); block = 256 columns; LDS acc[vocab][256]
__global__ __launch_bounds__(256) void embed_pe_bwd_kernel(const float* __restrict__ dout,
                                                           const int64_t* __restrict__ tok,
                                                           float* dcond, float* partial, int B,
                                                           int S, int n_c, int d, int vocab,
                                                           float gscale, uint32_t thr, GctRng rng,
                                                           int64_t groups_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [vocab][256]
  const int tx = threadIdx.x, col = blockIdx.x * 256 + tx;
  for (int v = 0; v < vocab; ++v) acc[v * 256 + tx] = 0.f;
  const int L = S + n_c;
  const int64_t rows = (int64_t)B * L, ngroups = (rows + 3) / 4;
  const int64_t g0 = (int64_t)blockIdx.y * groups_per_chunk;
  const int64_t g1 = g0 + groups_per_chunk < ngroups ? g0 + groups_per_chunk : ngroups;
  if (col < d) {
    // the gradient rows and token ids of group gq + 1 are requested before group gq is accumulated (the LDS
    // read-modify-write chain of a group hides behind the next group's HBM latency; without this the loop ran one
    // dependent load after the other: 140 us for 85 MB)
    float cur[4], nxt[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t ctok[4], ntok[4] = {0, 0, 0, 0};
    auto fetch = [&](int64_t gq, float (&v)[4], int64_t (&t)[4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (int)gq * 4 + e;
        const int rr = row < (int)rows ? row : (int)rows - 1;
        v[e] = dout[(int64_t)rr * d + col];
        const int b = rr / L, l = rr - b * L;
        t[e] = l < n_c ? 0 : tok[(int64_t)b * S + (l - n_c)];
      }
    };
    if (g0 < g1) fetch(g0, cur, ctok);
    for (int64_t gq = g0; gq < g1; ++gq) {
      if (gq + 1 < g1) fetch(gq + 1, nxt, ntok);
      uint4 bits = make_uint4(~0u, ~0u, ~0u, ~0u);
      if (thr) bits = gct_drop_bits(rng, (uint32_t)gq, (uint32_t)col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (int)gq * 4 + e;
        if (row < (int)rows) {
          const float g = gct_drop_keep(bits, e, (uint32_t)col, thr) ? cur[e] * gscale : 0.f;
          const int b = row / L, l = row - b * L;
          if (l < n_c) {
            dcond[((int64_t)b * n_c + l) * d + col] = g;
          } else {
            int64_t t = ctok[e];
            t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
            acc[t * 256 + tx] += g;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cur[e] = nxt[e];
        ctok[e] = ntok[e];
      }
    }
    float* p = partial + (int64_t)blockIdx.y * vocab * d;
    for (int v = 0; v < vocab; ++v) p[(int64_t)v * d + col] = acc[v * 256 + tx];
  }
}

inline int embed_chunks(int64_t rows) {
  int64_t c = (rows / 4 + 15) / 16;      // >= 16 groups of 4 rows per chunk; 2 x 512 workgroups at B x S = 41k rows
  if (c > 512) c = 512;
  if (c < 1) c = 1;
  return (int)c;
}

}  // namespace

extern "C" int64_t gct_embed_ws_bytes(int B, int S, int d, int vocab) {
  return (int64_t)embed_chunks((int64_t)B * (S + 8)) * vocab * d * (int64_t)sizeof(float) + 256;
}

extern "C" int gct_embed_pe_fwd(const int64_t* tok, const float* table, const float* cond,
                                const float* pe, float* out, int B, int S, int n_c, int d,
                                int vocab, float scale, float p, uint64_t seed, uint32_t site,
                                void* stream) {
  // S == 0: pure "x*scale + pe, dropout" over the cond rows (standalone PositionalEncoding)
  GCT_CHECK_ARG(pe && out && B >= 0 && S >= 0 && d > 0 && vocab > 0 && n_c >= 0 && S + n_c > 0,
                "embed_pe_fwd: bad args");
  GCT_CHECK_ARG(S == 0 || (tok && table), "embed_pe_fwd: tok/table missing");
  GCT_CHECK_ARG(n_c == 0 || cond, "embed_pe_fwd: cond rows requested without a cond buffer");
  GCT_CHECK_ARG(p >= 0.f && p < 1.f, "embed_pe_fwd: dropout p out of range");
  GCT_CHECK_ARG(d % 4 == 0 && (int64_t)B * (S + n_c) * d < (1ll << 31) && gct_aligned16(out) && gct_aligned16(pe) &&
                    (!table || gct_aligned16(table)) && (!cond || gct_aligned16(cond)),
                "embed_pe_fwd: d %% 4 == 0, 16-B aligned buffers and B*L*d < 2^31 required");
  if (B == 0) return GCT_OK;
  const int64_t total = (((int64_t)B * (S + n_c) + 3) / 4) * (d / 4);
  int64_t grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     tok, table, cond, pe, out, B, S, n_c, d, vocab, scale, gct_drop_threshold(p),
                     1.0f / (1.0f - p), gct_rng_make(seed, site));
  GCT_LAUNCH_CHECK("embed_pe_fwd");
  return GCT_OK;
}

int gct_reduce_slabs_seg(const float* slabs, int nslab, int64_t stride, float* d0, float* d1,
                         float* d2, int64_t nper_elems, int64_t n, hipStream_t st);

extern "C" int gct_embed_pe_bwd(const float* dout, const int64_t* tok, float* dtable, float* dcond,
                                float* ws, int B, int S, int n_c, int d, int vocab, float scale,
                                float p, uint64_t seed, uint32_t site, void* stream) {
  GCT_CHECK_ARG(dout && ws && B >= 0 && S >= 0 && d > 0 && vocab > 0 && n_c >= 0 && S + n_c > 0,
                "embed_pe_bwd: bad args");
  GCT_CHECK_ARG(S == 0 || (tok && dtable), "embed_pe_bwd: tok/dtable missing");
  GCT_CHECK_ARG(n_c == 0 || dcond, "embed_pe_bwd: dcond missing");
  GCT_CHECK_ARG(vocab <= 64, "embed_pe_bwd: vocab %d > 64 unsupported (LDS table)", vocab);
  GCT_CHECK_ARG(p >= 0.f && p < 1.f, "embed_pe_bwd: dropout p out of range");
  GCT_CHECK_ARG((int64_t)B * (S + n_c) + 4 < (1ll << 31), "embed_pe_bwd: B * L < 2^31 required");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * (S + n_c);
  const int chunks = embed_chunks((int64_t)B * (S + 8));
  const int64_t ngroups = (rows + 3) / 4;
  int64_t gpc = (ngroups + chunks - 1) / chunks;
  if (gpc < 1) gpc = 1;
  dim3 grid((unsigned)((d + 255) / 256), (unsigned)chunks);
  hipLaunchKernelGGL(embed_pe_bwd_kernel, grid, dim3(256), (size_t)vocab * 256 * sizeof(float), st,
                     dout, tok, dcond, ws, B, S, n_c, d, vocab, scale / (1.0f - p),
                     gct_drop_threshold(p), gct_rng_make(seed, site), gpc);
  GCT_LAUNCH_CHECK("embed_pe_bwd");
  if (S == 0) return GCT_OK;
  const int64_t n = (int64_t)vocab * d;
  return gct_reduce_slabs_seg(ws, chunks, n, dtable, nullptr, nullptr, n, n, st);
}
