// KV-cached autoregressive decode helpers (reference
// Inference/sampling_tool.py:140-184 re-runs the WHOLE decoder on ys[:, :i+1] every step).
//   gct_attn_decode  : one query row per (sample, head) against cached keys/values
//   gct_select_token : softmax over the vocabulary + greedy / multinomial choice, appends the
//                      token, updates the key-valid flags and the per-sample finished mask
// Both are tiny and HBM/latency-bound; they exist so a whole decode step is a fixed kernel
// chain with no host round trip (graph-capturable).
#include "common.h"

namespace {

// one wave per (b, h); 16 lanes per key for the scores (4 keys per pass), lanes over d for P.V.
// pos (nullable): DEVICE-side step counter -- the cache holds Lc = cache_off + *pos keys, and the step's own key /
// value (knew / vnew, row b of a [n][ldn] step buffer) are the last key: they are appended to the caches here
// (row Lc) and scored from registers, so ONE captured graph serves every step of the decode loop.
template <int DK>
__global__ __launch_bounds__(256) void attn_decode_kernel(
    const float* __restrict__ q, int64_t ldq, float* __restrict__ k, float* __restrict__ v,
    int64_t kv_row, int64_t kv_batch, const uint8_t* __restrict__ valid, int64_t valid_sb,
    float* __restrict__ o, int64_t ldo, int n, int H, int Lc_host, float scale,
    const int32_t* __restrict__ pos, int cache_off, const float* __restrict__ knew,
    const float* __restrict__ vnew, int64_t ldn, const int32_t* __restrict__ klen) {
  __shared__ float sc[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= (int64_t)n * H) return;
  const int b = (int)(pair / H), h = (int)(pair - (int64_t)b * H);
  // klen (nullable, without pos): the keys of sample b beyond klen[b] are masked (`valid` says so too) and b sees at
  // least one key, so they weigh exactly 0: their cache rows are never read (the padded latent positions of the
  // cross-attention memory are most of it at MOSES-like lengths)
  const int Lold = pos ? cache_off + *pos : (klen ? klen[b] : Lc_host);          // keys already in the cache
  const int Lc = pos ? Lold + 1 : Lold;
  const float* qp = q + (int64_t)b * ldq + h * DK;
  float* kp = k + (int64_t)b * kv_batch + h * DK;
  float* vp = v + (int64_t)b * kv_batch + h * DK;
  const uint8_t* vl = valid ? valid + (int64_t)b * valid_sb : nullptr;
  constexpr int LPK = DK / 4;        // lanes per key (float4 each)
  constexpr int KPP = 64 / LPK;      // keys per pass
  const int sub = lane % LPK, kslot = lane / LPK;
  const float4 qv = *reinterpret_cast<const float4*>(qp + sub * 4);
  float vlast = 0.f;
  if (pos) {
    // this step's key: score it from the step buffer and append key / value to the caches
    const float* kn = knew + (int64_t)b * ldn + h * DK;
    const float* vn = vnew + (int64_t)b * ldn + h * DK;
    float s = 0.f;
    if (lane < LPK) {
      const float4 kv4 = *reinterpret_cast<const float4*>(kn + lane * 4);
      s = (qv.x * kv4.x + qv.y * kv4.y) + (qv.z * kv4.z + qv.w * kv4.w);   // sub == lane here
      *reinterpret_cast<float4*>(kp + (int64_t)Lold * kv_row + lane * 4) = kv4;
      *reinterpret_cast<float4*>(vp + (int64_t)Lold * kv_row + lane * 4) = *reinterpret_cast<const float4*>(vn + lane * 4);
    }
#pragma unroll
    for (int off = LPK / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      s *= scale;
      if (vl && vl[Lold] == 0) s = -1e9f;
      sc[wave][Lold] = s;
    }
    if (lane < DK) vlast = vn[lane];
  }
  float m = -INFINITY;
  for (int j0 = 0; j0 < Lold; j0 += KPP) {
    const int j = j0 + kslot;
    float s = 0.f;
    if (j < Lold) {
      const float4 kv4 = *reinterpret_cast<const float4*>(kp + (int64_t)j * kv_row + sub * 4);
      s = (qv.x * kv4.x + qv.y * kv4.y) + (qv.z * kv4.z + qv.w * kv4.w);
    }
#pragma unroll
    for (int off = LPK / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (j < Lold && sub == 0) {
      s *= scale;
      if (vl && vl[j] == 0) s = -1e9f;            // masked_fill(mask == 0, -1e9)
      sc[wave][j] = s;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < Lc; j += 64) m = fmaxf(m, sc[wave][j]);
  m = gct_wave_max(m);
  float l = 0.f;
  for (int j = lane; j < Lc; j += 64) {
    const float e = expf(sc[wave][j] - m);
    sc[wave][j] = e;
    l += e;
  }
  l = gct_wave_sum(l);
  const float inv = 1.0f / l;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (lane < DK) {
    float acc = 0.f;
    for (int j = 0; j < Lold; ++j) acc = fmaf(sc[wave][j] * inv, vp[(int64_t)j * kv_row + lane], acc);
    if (pos) acc = fmaf(sc[wave][Lold] * inv, vlast, acc);
    o[(int64_t)b * ldo + h * DK + lane] = acc;
  }
}

// Cross-attention of one decode step over the LATENT memory itself.  The cross-attention keys / values of the
// reference are projections of e_j = fc_z(z_j): k_{j,h} = G_h z_j + c_h, v_{j,h} = H_h z_j + d_h with G_h = W_k,h W_z
// and H_h = W_v,h W_z (64 x latent).  A score q_h . k_{j,h} is (G_h^T q_h) . z_j plus a term that is the same for
// every key of the row (softmax-invariant), and sum_j p_j v_{j,h} = H_h (sum_j p_j z_j) + d_h.  With G_h^T folded into
// the query projection and H_h into the output projection (host side, once per sequence), a step reads the z rows of
// a sample ONCE for all heads -- latent x 4 B per key instead of 2 x d_model x 4 B (8 x less at latent 128, d 512).
// One workgroup per sample: z rows -> LDS, scores for (head, key) pairs, softmax per head, context (head, latent).
// The n_c condition rows of a cond2lat memory are not functions of z: they keep explicit per-head keys / values
// (shifted by -c_h / -d_h on the host so that the two kinds of key share one softmax).
__global__ __launch_bounds__(256) void attn_decode_z_kernel(
    const float* __restrict__ q, int64_t ldq, int qoff, const float* __restrict__ z, int64_t z_batch, int lat, int Le,
    const float* __restrict__ ckv, int64_t ckv_batch, int64_t ld_ckv, int nc, int d, const uint8_t* __restrict__ valid,
    int64_t valid_sb, const int32_t* __restrict__ klen, float* __restrict__ out, int64_t ldo, int ooff, int H, int dk,
    float scale) {
  extern __shared__ __attribute__((aligned(16))) float zl[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Lk = nc + Le, zs_ld = lat + 4, lat4 = lat >> 2, Lp = (Lk + 3) & ~3;
  float* zs = zl;                              // [Le][lat + 4]
  float* qs = zl + (size_t)Le * zs_ld;         // [H][lat]
  float* ps = qs + (size_t)H * lat;            // [H][Lp]
  // keys beyond klen[b] are masked and the sample sees a key: weight exactly 0, never read
  const int Lvis = klen ? (klen[b] < Lk ? klen[b] : Lk) : Lk;
  const int ncv = nc < Lvis ? nc : Lvis, nz = Lvis - ncv;
  const float* zb = z + (int64_t)b * z_batch;
  const float* qb = q + (int64_t)b * ldq;
  for (int i = tid; i < nz * lat4; i += 256) {
    const int row = i / lat4, c4 = i - row * lat4;
    *reinterpret_cast<float4*>(zs + row * zs_ld + c4 * 4) = *reinterpret_cast<const float4*>(zb + (int64_t)row * lat + c4 * 4);
  }
  for (int i = tid; i < H * lat4; i += 256)
    *reinterpret_cast<float4*>(qs + i * 4) = *reinterpret_cast<const float4*>(qb + qoff + i * 4);
  __syncthreads();
  const uint8_t* vl = valid ? valid + (int64_t)b * valid_sb : nullptr;
  for (int i = tid; i < H * nz; i += 256) {          // consecutive lanes: consecutive keys of one head
    const int h = i / nz, j = i - h * nz;
    const float* zr = zs + j * zs_ld;
    const float* qh = qs + h * lat;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int c = 0; c < lat; c += 4) {
      const float4 a = *reinterpret_cast<const float4*>(qh + c), k4 = *reinterpret_cast<const float4*>(zr + c);
      s0 = fmaf(a.x, k4.x, s0); s1 = fmaf(a.y, k4.y, s1); s2 = fmaf(a.z, k4.z, s2); s3 = fmaf(a.w, k4.w, s3);
    }
    float sc = ((s0 + s1) + (s2 + s3)) * scale;
    if (vl && vl[ncv + j] == 0) sc = -1e9f;          // masked_fill(mask == 0, -1e9)
    ps[h * Lp + ncv + j] = sc;
  }
  for (int i = tid; i < H * ncv; i += 256) {         // explicit condition rows: q_h . k'_{j,h} over the head dim
    const int h = i / ncv, j = i - h * ncv;
    const float* kr = ckv + (int64_t)b * ckv_batch + (int64_t)j * ld_ckv + h * dk;
    const float* qh = qb + h * dk;
    float sc = 0.f;
    for (int c = 0; c < dk; ++c) sc = fmaf(qh[c], kr[c], sc);
    sc *= scale;
    if (vl && vl[j] == 0) sc = -1e9f;
    ps[h * Lp + j] = sc;
  }
  __syncthreads();
  for (int h = wave; h < H; h += 4) {                // softmax of one head per wave
    float* pr = ps + h * Lp;
    float m = -INFINITY;
    for (int j = lane; j < Lvis; j += 64) m = fmaxf(m, pr[j]);
    m = gct_wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Lvis; j += 64) {
      const float e = expf(pr[j] - m);
      pr[j] = e;
      l += e;
    }
    l = gct_wave_sum(l);
    const float inv = 1.0f / l;
    for (int j = lane; j < Lvis; j += 64) pr[j] *= inv;
  }
  __syncthreads();
  float* ob = out + (int64_t)b * ldo;
  for (int i = tid; i < H * lat4; i += 256) {        // context of head h in the latent space: sum_j p_j z_j
    const int h = i / lat4, c4 = i - h * lat4;
    const float* pr = ps + h * Lp + ncv;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < nz; ++j) {
      const float w = pr[j];
      const float4 z4 = *reinterpret_cast<const float4*>(zs + j * zs_ld + c4 * 4);
      acc.x = fmaf(w, z4.x, acc.x); acc.y = fmaf(w, z4.y, acc.y); acc.z = fmaf(w, z4.z, acc.z); acc.w = fmaf(w, z4.w, acc.w);
    }
    *reinterpret_cast<float4*>(ob + ooff + h * lat + c4 * 4) = acc;
  }
  if (nc > 0) {
    for (int i = tid; i < d; i += 256) {             // condition rows: sum_j p_j v'_{j,h}, head-major like the reference
      const int h = i / dk;
      const float* pr = ps + h * Lp;
      float acc = 0.f;
      for (int j = 0; j < ncv; ++j) acc = fmaf(pr[j], ckv[(int64_t)b * ckv_batch + (int64_t)j * ld_ckv + d + i], acc);
      ob[i] = acc;
    }
  }
}

// x[b][:] = table[ys[b][*pos]] * scale + pe[pe_off + *pos][:]   (Embeddings + PositionalEncoding of ONE position)
__global__ __launch_bounds__(256) void decode_embed_kernel(const int64_t* __restrict__ ys, int64_t ld_ys,
                                                           const int32_t* __restrict__ pos, int pe_off,
                                                           const float* __restrict__ table, int vocab,
                                                           const float* __restrict__ pe, float* __restrict__ out,
                                                           int n, int d, float scale) {
  const int p = *pos;
  const int64_t total = (int64_t)n * (d / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / (d / 4)), c = (int)(i - (int64_t)b * (d / 4));
    int64_t tok = ys[(int64_t)b * ld_ys + p];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const float4 e = *reinterpret_cast<const float4*>(table + tok * d + c * 4);
    const float4 q = *reinterpret_cast<const float4*>(pe + (int64_t)(pe_off + p) * d + c * 4);
    *reinterpret_cast<float4*>(out + (int64_t)b * d + c * 4) =
        make_float4(e.x * scale + q.x, e.y * scale + q.y, e.z * scale + q.z, e.w * scale + q.w);
  }
}

__global__ void decode_advance_kernel(int32_t* pos) { *pos += 1; }

__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// one wave per sample row.  pos_dev (nullable): device-side step counter -- the token is written at ys[.., *pos_dev + 1]
// and valid[.., valid_off + *pos_dev + 1]; seed_dev (nullable) replaces the by-value seed of the multinomial draw
__global__ __launch_bounds__(256) void select_token_kernel(const float* __restrict__ logits, int V,
                                                           int64_t* ys, int64_t ld_ys, int pos_host,
                                                           uint8_t* valid, int64_t valid_sb,
                                                           uint8_t* done, float* probs_out, int n,
                                                           int mode, int64_t pad_id, int64_t eos_id,
                                                           GctRng rng, const int32_t* __restrict__ pos_dev,
                                                           int valid_off, const uint64_t* __restrict__ seed_dev) {
  const int pos = pos_dev ? *pos_dev + 1 : pos_host;
  if (seed_dev) rng = gct_rng_make(*seed_dev, 0xDEC0DEu);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  const float* lr = logits + (int64_t)row * V;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lr[c]);
  mx = gct_wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < V; c += 64) se += expf(lr[c] - mx);
  se = gct_wave_sum(se);
  const float inv = 1.0f / se;
  int best = 0;
  if (mode == 0) {
    // greedy: first index of the maximum probability (torch.max semantics)
    float bp = -1.f;
    int bi = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
      const float p = expf(lr[c] - mx) * inv;
      if (probs_out) probs_out[(int64_t)row * V + c] = p;
      if (p > bp) { bp = p; bi = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float op = __shfl_xor(bp, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (op > bp || (op == bp && oi < bi)) { bp = op; bi = oi; }
    }
    best = bi;
  } else {
    // multinomial: inverse CDF with one Philox uniform per (row, position)
    const uint4 r = gct_philox(rng, (uint32_t)row, (uint32_t)pos, 0x452821E6u, 0x38D01377u);
    const float u = u01_open(r.x);
    float cum = 0.f;
    int pick = V - 1;
    bool found = false;
    for (int c0 = 0; c0 < V; c0 += 64) {
      const int c = c0 + lane;
      float p = c < V ? expf(lr[c] - mx) * inv : 0.f;
      if (probs_out && c < V) probs_out[(int64_t)row * V + c] = p;
      float incl = p;                                   // inclusive scan over the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      const bool hit = !found && c < V && (cum + incl) > u;
      const unsigned long long ball = __ballot(hit);
      if (ball && !found) {
        pick = c0 + (int)__builtin_ctzll(ball);
        found = true;
      }
      cum += __shfl(incl, 63, 64);
    }
    best = pick;
  }
  if (lane == 0) {
    ys[(int64_t)row * ld_ys + pos] = best;
    if (valid) valid[(int64_t)row * valid_sb + valid_off + pos] = (best != pad_id) ? 1 : 0;
    if (done && best == eos_id) done[row] = 1;
  }
}

}  // namespace

extern "C" int gct_attn_decode(const float* q, int64_t ldq, float* k, float* v,
                               int64_t kv_row, int64_t kv_batch, const uint8_t* valid,
                               int64_t valid_sb, float* o, int64_t ldo, int n, int H, int Lc, int dk,
                               float scale, const int32_t* pos, int cache_off, const float* knew,
                               const float* vnew, int64_t ldn, const int32_t* klen, void* stream) {
  GCT_CHECK_ARG(q && k && v && o && n >= 0 && H > 0 && Lc >= 0 && Lc <= 256, "attn_decode: bad args");
  GCT_CHECK_ARG(!(klen && pos), "attn_decode: klen is for a fixed cache (cross-attention), not the device-position form");
  GCT_CHECK_ARG(pos || Lc > 0, "attn_decode: no keys");
  GCT_CHECK_ARG(!pos || (knew && vnew && ldn % 4 == 0 && gct_aligned16(knew) && gct_aligned16(vnew) && cache_off >= 0),
                "attn_decode: the device-position form needs this step's key / value rows");
  GCT_CHECK_ARG(dk == 16 || dk == 32 || dk == 64, "attn_decode: head dim %d unsupported", dk);
  GCT_CHECK_ARG(ldq % 4 == 0 && kv_row % 4 == 0 && kv_batch % 4 == 0 && gct_aligned16(q) &&
                    gct_aligned16(k) && gct_aligned16(v),
                "attn_decode: operands must be 16-B aligned");
  if (n == 0) return GCT_OK;
  const int64_t pairs = (int64_t)n * H;
  dim3 grid((unsigned)((pairs + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dk == 64) hipLaunchKernelGGL(attn_decode_kernel<64>, grid, block, 0, st, q, ldq, k, v, kv_row, kv_batch, valid, valid_sb, o, ldo, n, H, Lc, scale, pos, cache_off, knew, vnew, ldn, klen);
  else if (dk == 32) hipLaunchKernelGGL(attn_decode_kernel<32>, grid, block, 0, st, q, ldq, k, v, kv_row, kv_batch, valid, valid_sb, o, ldo, n, H, Lc, scale, pos, cache_off, knew, vnew, ldn, klen);
  else hipLaunchKernelGGL(attn_decode_kernel<16>, grid, block, 0, st, q, ldq, k, v, kv_row, kv_batch, valid, valid_sb, o, ldo, n, H, Lc, scale, pos, cache_off, knew, vnew, ldn, klen);
  GCT_LAUNCH_CHECK("attn_decode");
  return GCT_OK;
}

extern "C" int gct_attn_decode_z(const float* q, int64_t ldq, int qoff, const float* z, int64_t z_batch, int lat, int Le,
                                 const float* ckv, int64_t ckv_batch, int64_t ld_ckv, int nc, const uint8_t* valid,
                                 int64_t valid_sb, const int32_t* klen, float* out, int64_t ldo, int ooff, int n, int H,
                                 int dk, float scale, void* stream) {
  GCT_CHECK_ARG(q && z && out && n >= 0 && H > 0 && dk > 0 && Le >= 0 && nc >= 0 && nc + Le > 0 && nc + Le <= 259,
                "attn_decode_z: bad args");
  GCT_CHECK_ARG(lat > 0 && lat % 4 == 0 && lat <= 128 && Le <= 256, "attn_decode_z: latent size %d / %d rows unsupported", lat, Le);
  GCT_CHECK_ARG(nc == 0 || (ckv && ld_ckv >= 2 * (int64_t)H * dk), "attn_decode_z: condition rows without their keys / values");
  GCT_CHECK_ARG(ldq % 4 == 0 && qoff % 4 == 0 && ooff % 4 == 0 && ldo % 4 == 0 && z_batch % 4 == 0 && gct_aligned16(q) &&
                    gct_aligned16(z) && gct_aligned16(out),
                "attn_decode_z: operands must be 16-B aligned");
  if (n == 0) return GCT_OK;
  const size_t lds = ((size_t)Le * (lat + 4) + (size_t)H * lat + (size_t)H * ((nc + Le + 3) & ~3)) * sizeof(float);
  GCT_CHECK_ARG(lds <= 160 * 1024, "attn_decode_z: %zu bytes of LDS needed", lds);
  static size_t lds_set = 0;
  if (lds > 48 * 1024 && lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_decode_z_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) {
      gct_set_error("attn_decode_z: cannot reserve LDS: %s", hipGetErrorString(e));
      return GCT_ERR_HIP;
    }
    lds_set = 160 * 1024;
  }
  hipLaunchKernelGGL(attn_decode_z_kernel, dim3((unsigned)n), dim3(256), lds, (hipStream_t)stream, q, ldq, qoff, z, z_batch,
                     lat, Le, ckv, ckv_batch, ld_ckv, nc, H * dk, valid, valid_sb, klen, out, ldo, ooff, H, dk, scale);
  GCT_LAUNCH_CHECK("attn_decode_z");
  return GCT_OK;
}

extern "C" int gct_decode_embed(const int64_t* ys, int64_t ld_ys, const int32_t* pos, int pe_off, const float* table,
                                int vocab, const float* pe, float* out, int n, int d, float scale, void* stream) {
  GCT_CHECK_ARG(ys && pos && table && pe && out && n >= 0 && d > 0 && d % 4 == 0 && vocab > 0 && pe_off >= 0 &&
                    gct_aligned16(table) && gct_aligned16(pe) && gct_aligned16(out),
                "decode_embed: bad args");
  if (n == 0) return GCT_OK;
  const int64_t work = (int64_t)n * (d / 4);
  hipLaunchKernelGGL(decode_embed_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ys,
                     ld_ys, pos, pe_off, table, vocab, pe, out, n, d, scale);
  GCT_LAUNCH_CHECK("decode_embed");
  return GCT_OK;
}

extern "C" int gct_decode_advance(int32_t* pos, void* stream) {
  GCT_CHECK_ARG(pos, "decode_advance: null pointer");
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos);
  GCT_LAUNCH_CHECK("decode_advance");
  return GCT_OK;
}

extern "C" int gct_select_token(const float* logits, int V, int64_t* ys, int64_t ld_ys, int pos,
                                uint8_t* valid, int64_t valid_sb, uint8_t* done, float* probs_out,
                                int n, int mode, int64_t pad_id, int64_t eos_id, uint64_t seed,
                                const int32_t* pos_dev, int valid_off, const uint64_t* seed_dev, void* stream) {
  GCT_CHECK_ARG(logits && ys && V > 0 && n >= 0 && pos >= 0 && (mode == 0 || mode == 1) && valid_off >= 0,
                "select_token: bad args");
  if (n == 0) return GCT_OK;
  hipLaunchKernelGGL(select_token_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, logits, V, ys, ld_ys, pos, valid, valid_sb, done, probs_out,
                     n, mode, pad_id, eos_id, gct_rng_make(seed, 0xDEC0DEu), pos_dev, valid_off, seed_dev);
  GCT_LAUNCH_CHECK("select_token");
  return GCT_OK;
}
