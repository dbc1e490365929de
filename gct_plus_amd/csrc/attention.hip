// K4: multi-head attention core, forward and backward, L <= 208.
// Reference: Model/sublayers.py:29-41 attention():  softmax(q k^T / sqrt(dk) masked_fill(mask==0,
// -1e9)) -> dropout on the probabilities -> . v ; head split/merge of sublayers.py:64-69 is
// folded into the addressing (q/k/v are read in place from the fused projection buffer, o is
// written heads-merged).  The reference's positional table allows any L <= 200 (Model/modules.py:117).
//
// Machine mapping (gfx950).  All products run on v_mfma_f32_16x16x4_f32 (exact fp32).
//  * The score tile is computed TRANSPOSED (S^T = K Q^T): its accumulator layout has the key
//    index on the registers and the query on the lane -- exactly the B-operand layout of the
//    following P.V product (which sums over keys).  Probabilities never leave registers and
//    never touch HBM; a softmax row is reduced over 4 regs x tiles in-lane plus two shuffles.
//  * Masks arrive PACKED (gct_attn_mask_pack: one bit per key, 8 words per query row, packed once per
//    forward and shared by all layers and heads, plus one word of visible key tiles per 16-row query tile).
//    A lane holds the words of its own query row in registers; visibility of a 16x16 tile, "row has a
//    visible key" and the per-element mask bit are bit operations on them.
//  * 16x16 score tiles without a visible (q,k) are skipped -- wave-uniform, and only when every query
//    row of the tile has at least one visible key, so the masked_fill(-1e9) semantics (uniform row
//    when everything is masked) stay exact.  Dropout keep bits: one Philox call per (query, 8 keys) in
//    the lane that owns them, only for tiles that are computed.
//  * The backward recomputes probabilities from the saved log-sum-exp (no [B,H,L,L] tensor).
// Two kernel families:
//  * L_k <= 96 (every shipped configuration): BARRIER-FREE kernels, one wave per work item, no workgroup
//    cooperation, a non-persistent grid the dispatcher balances: attn_fwd_direct_kernel (item = pair x query
//    tile), attn_bwd_dq_kernel (pair x query tile -> dQ + per-row scalars / keep bits / visited tiles in a
//    workspace), attn_bwd_dkv_kernel (pair x key tile -> dK, dV).  Operands reach a wave coalesced and are turned
//    into the row-per-lane MFMA fragments in wave-private LDS tiles (WaveTile); a pair's K / V are re-read by its
//    query tiles from L2, not HBM (measured HBM traffic = algorithmic bytes +1 %).  Register use is kept low on
//    purpose (71 / 118 / 120 VGPRs): what hides the memory latency here is the number of resident waves.
//  * 96 < L <= 208: the LDS kernels attn_fwd_kernel / attn_bwd_kernel -- one (batch, head) pair at a time in a
//    persistent 6-wave workgroup's LDS (K and V [L][dk+4] fp32, software-pipelined over the pairs; the backward in
//    two phases that reuse one LDS region).  GCT_ATTN_FWD_LDS=1 / GCT_ATTN_BWD_LDS=1 select them for any length.
// Roofline: HBM-bound on q,k,v,o (+ gradients): 0.57 / 0.51 of it fwd / bwd at B=512;
// 4.L.d flop/token ~ 3 % of the step's flops.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ATT_THREADS = 384;       // 6 waves: one per query / key tile at L <= 96
constexpr int MASK_W = 8;              // packed mask: 8 x 32 bits per query row
constexpr int L_MAX = 208;             // 13 tiles of 16: covers the reference's 200-row positional table + 3 conditions

struct AttnArgs {
  // strides are 32-bit (checked on the host): 64-bit strides cost SGPRs this kernel does not have to spare -- they
  // were spilled into VGPR lanes (v_readlane on every use)
  const float *q, *k, *v;
  int ldq, ldk, ldv;
  const uint32_t* mbits;               // packed mask rows (nullable = everything visible)
  int mb_sb, mb_sq;                    // strides in words: per batch, per query row (0: key-padding mask)
  // visible key tiles per (batch, 16-row query tile), precomputed by gct_attn_mask_pack (nullable: computed from the
  // mask rows by every wave -- a dependent load + ~80 VALU before the first K request can go out)
  const uint32_t* tbits;
  int tb_sb, tb_su;                    // strides in words: per batch, per query tile (0: key-padding mask)
  float* o;
  int ldo;
  float* lse;
  float* probs;
  // backward
  const float *o_in, *dout, *lse_in;
  float *dq, *dk, *dv;
  int lddq, lddk, lddv;
  // compacted decoder backward (csrc/liverows.hip): dout / dq rows of sample b start at cstart[b] and only the first
  // nlive[b] query rows exist; kv_compact: dk / dv (self-attention) live in the same compact rows
  const int32_t *cstart, *nlive;
  int kv_compact;
  // the FORWARD ran on the compact query rows as well (decoder forward over the live rows): in the forward q and o,
  // in the backward q and o_in, live at rows cstart[b] .. cstart[b] + nlive[b] like dout / dq
  int qo_compact;
  // compacted keys / values (cross-attention over the encoder memory without its padded rows): the K / V rows of
  // sample b start at kstart[b] and only its first klen[b] keys exist (the rest are masked keys: zero rows in LDS);
  // the backward's dk / dv live in the same rows
  const int32_t *kstart, *klen;
  // direct backward (two launches): per query row {lse, delta, masked score, dO row non-zero}, the dropout keep
  // bits in the packed-mask layout, one word of visited key tiles per (pair, query tile) -- written by the dQ kernel, read by
  // the dK / dV kernel
  float4* ws_meta;
  uint32_t* ws_keep;
  uint32_t* ws_use;                    // per (pair, query tile): the key tiles it computed (0: dead tile)
  int B, H, Lq, Lk, npairs;
  float scale, keep_scale;
  uint32_t thr;
  GctRng rng;
#ifdef GCT_STAMPS
  unsigned long long* stamps;
#endif
};

// Diagnostic build only (tools/attn_stamps.hip, -DGCT_STAMPS): s_memtime at the pipeline's seams, summed per wave.
#ifdef GCT_STAMPS
#define ASTAMP_DECL unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory")
#define ASTAMP(i)                                                                      \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    unsigned long long t__;                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    seg[i] += t__ - tprev;                                                             \
    tprev = t__;                                                                       \
  } while (0)
#define ASTAMP_OUT                                                                     \
  if (a.stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 64)                          \
    for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * (ATT_THREADS / 64) + (threadIdx.x >> 6)) * 8 + i] = seg[i]
// direct kernels: one record per wave (= item), items [GCT_STAMP_ITEM0, GCT_STAMP_ITEM0 + 4096)
#define GCT_STAMP_ITEM0 8192
#define DSTAMP_OUT_AT(item, rec0)                                                      \
  if (a.stamps && (threadIdx.x & 63) == 0 && (item) >= GCT_STAMP_ITEM0 && (item) < GCT_STAMP_ITEM0 + 4096)  \
    for (int i = 0; i < 8; ++i) a.stamps[((size_t)((item) - GCT_STAMP_ITEM0 + (rec0))) * 8 + i] = seg[i]
#define DSTAMP_OUT(item) DSTAMP_OUT_AT(item, 0)
#else
#define ASTAMP_DECL
#define ASTAMP(i)
#define ASTAMP_OUT
#define DSTAMP_OUT(item)
#define DSTAMP_OUT_AT(item, rec0)
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Staging of a [B][L][ld] head slice into LDS [LP][SD] (zero padded) is split in a load half (registers) and a
// store half, so the loads of the NEXT pair can be in flight during the MFMAs of this one.  The load half is
// branch-free and does not touch the loaded values (rows beyond L re-read row L-1; they are zeroed by the store
// half): anything that consumes a loaded register makes the wave wait for it AND for every older load (vmcnt is
// in-order), which would turn the prefetch into a stall.
template <int DK, int NT>
struct StageOff {           // BYTE offsets of a thread's chunks inside a pair's [L][ld] slice (pair independent):
  static constexpr int C = DK / 4;                                 // unsigned 32 bit => global_load with a scalar
  static constexpr int ITERS = (16 * NT * C + ATT_THREADS - 1) / ATT_THREADS;    // base and a 32-bit vector offset
  uint32_t off[ITERS];
  __device__ __forceinline__ void init(int ld, int L, int tid) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * ATT_THREADS;
      const int r = idx / C, c = idx - r * C;
      off[it] = (uint32_t)(((r < L ? r : L - 1) * ld + c * 4) * 4);
    }
  }
};
template <int DK, int NT>
struct Stage {
  static constexpr int SD = DK + 4, C = DK / 4;
  static constexpr int ITERS = StageOff<DK, NT>::ITERS;
  float4 v[ITERS];
  __device__ __forceinline__ void load(const float* src, int ld, int b, int h, int L, const StageOff<DK, NT>& o) {
    const float* base = src + (int64_t)b * L * ld + h * DK;      // wave-uniform
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      v[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + o.off[it]);
  }
  __device__ __forceinline__ void load_rows(const float* src, int ld, int64_t row0, int h, const StageOff<DK, NT>& o) {
    const float* base = src + row0 * ld + h * DK;                 // wave-uniform
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      v[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + o.off[it]);
  }
  __device__ __forceinline__ void store(float* dst, int L, int LP, int tid) const {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * ATT_THREADS;
      const int r = idx / C, c = idx - r * C;
      if (r < LP)
        *reinterpret_cast<float4*>(dst + r * SD + c * 4) = r < L ? v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// Operand fragments for products that contract over the head dimension (S = Q K^T, dP = dO V^T):
// lane group g owns dk indices [g*DK/4, (g+1)*DK/4), so a lane's whole fragment of one row is
// NDT consecutive float4 -- NDT 16-byte reads feed DK/4 MFMAs (any k permutation is legal when
// both operands use it).
template <int NDT>
__device__ __forceinline__ void row_frag(float4 (&f)[NDT], const float* lds, int row, int g) {
  constexpr int SD = 16 * NDT + 4;
#pragma unroll
  for (int j = 0; j < NDT; ++j)
    f[j] = *reinterpret_cast<const float4*>(lds + row * SD + g * 4 * NDT + 4 * j);
}
// The same fragment straight from global memory (row of a [B][L][ld] head slice).  Rows beyond L re-read row
// L-1 (finite values; every consumer discards or masks what such rows produce) -- no select on the loaded data,
// see Stage.
template <int NDT>
__device__ __forceinline__ void row_frag_global(float4 (&f)[NDT], const float* src, int ld, int b, int h, int L,
                                                int row, int g) {
  const int rr = row < L ? row : L - 1;
  const char* base = reinterpret_cast<const char*>(src + (int64_t)b * L * ld + h * 16 * NDT);   // wave-uniform
  const uint32_t off = (uint32_t)((rr * ld + g * 4 * NDT) * 4);                                   // per lane
#pragma unroll
  for (int j = 0; j < NDT; ++j) f[j] = *reinterpret_cast<const float4*>(base + off + 16 * j);
}
template <int NDT>
__device__ __forceinline__ void row_frag_rows(float4 (&f)[NDT], const float* src, int ld, int64_t row0, int h, int L,
                                              int row, int g) {
  const int rr = row < L ? row : (L > 0 ? L - 1 : 0);
  const char* base = reinterpret_cast<const char*>(src + row0 * ld + h * 16 * NDT);              // wave-uniform
  const uint32_t off = (uint32_t)((rr * ld + g * 4 * NDT) * 4);
#pragma unroll
  for (int j = 0; j < NDT; ++j) f[j] = *reinterpret_cast<const float4*>(base + off + 16 * j);
}
template <int NDT>
__device__ __forceinline__ f32x4 dot_frag(const float4 (&a)[NDT], const float4 (&b)[NDT], f32x4 acc) {
#pragma unroll
  for (int j = 0; j < NDT; ++j) {
    acc = mfma16(a[j].x, b[j].x, acc);
    acc = mfma16(a[j].y, b[j].y, acc);
    acc = mfma16(a[j].z, b[j].z, acc);
    acc = mfma16(a[j].w, b[j].w, acc);
  }
  return acc;
}

// bits of word w (keys 32w .. 32w+31) that are inside [0, Lk)
__device__ __forceinline__ uint32_t range_word(int w, int Lk) {
  const int n = Lk - 32 * w;
  return n >= 32 ? 0xffffffffu : (n > 0 ? ((1u << n) - 1u) : 0u);
}

// packed mask words of query row q as stored (rows beyond Lq re-read row Lq-1); call only with a.mbits != null
template <int MW>
__device__ __forceinline__ void mask_row_raw(uint4 (&raw)[(MW + 3) / 4], const AttnArgs& a, int b, int q) {
  const int qq = q < a.Lq ? q : a.Lq - 1;
  const char* base = reinterpret_cast<const char*>(a.mbits + (int64_t)b * a.mb_sb);              // wave-uniform
  const uint32_t off = (uint32_t)(qq * a.mb_sq * 4);
#pragma unroll
  for (int w4 = 0; w4 < (MW + 3) / 4; ++w4) raw[w4] = *reinterpret_cast<const uint4*>(base + off + 16 * w4);
}
// ... and their use: all ones without a mask or beyond Lq
template <int MW>
__device__ __forceinline__ void mask_row_use(uint32_t (&mw)[MW], const uint4 (&raw)[(MW + 3) / 4], const AttnArgs& a,
                                             int q) {
  const bool on = a.mbits && q < a.Lq;
#pragma unroll
  for (int w4 = 0; w4 < (MW + 3) / 4; ++w4) {
    if (4 * w4 + 0 < MW) mw[4 * w4 + 0] = on ? raw[w4].x : 0xffffffffu;
    if (4 * w4 + 1 < MW) mw[4 * w4 + 1] = on ? raw[w4].y : 0xffffffffu;
    if (4 * w4 + 2 < MW) mw[4 * w4 + 2] = on ? raw[w4].z : 0xffffffffu;
    if (4 * w4 + 3 < MW) mw[4 * w4 + 3] = on ? raw[w4].w : 0xffffffffu;
  }
}

// bit t set => key tile t must be computed for the query tile whose row `q` this lane holds (mw = its mask words)
template <int MW>
__device__ __forceinline__ uint32_t tiles_for_q(const uint32_t (&mw)[MW], int q, int Lq, int Lk, int nkt) {
  uint32_t rowvis = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) rowvis |= mw[w] & range_word(w, Lk);
  const bool real = q < Lq;
  const bool ok = __all(!real || rowvis != 0);      // every real row of the tile sees a key
  uint32_t use = 0;
#pragma unroll
  for (int t = 0; t < 2 * MW; ++t)
    if (t < nkt) {
      const uint32_t nib = ((mw[t >> 1] & range_word(t >> 1, Lk)) >> ((t & 1) * 16)) & 0xffffu;
      if (!ok || __any(real && nib != 0)) use |= 1u << t;
    }
  return __builtin_amdgcn_readfirstlane(use);
}

// Dropout keep bits of one query row for the key tiles in `use`: bit 4t + r <=> key 16t + 4g + r is kept.  One
// Philox call serves the lane's 4 keys of TWO adjacent tiles: key r of tile t takes the 16-bit lane (t & 1) * 4 + r of
// philox(row, (t >> 1) * 4 + g) and is kept iff that lane >= thr >> 16 (P(drop) = floor(p * 2^16) / 2^16, the
// convention of the GEMM epilogues' dropout).  The loop is kept ROLLED: one call's registers live at a time.
template <int NT>
__device__ __forceinline__ uint64_t keep_bits_row(const AttnArgs& a, uint32_t grow, uint32_t use, int g) {
  uint64_t keep = ~0ull;
  if (a.thr) {
    keep = 0;
    const uint32_t th = a.thr >> 16;
#pragma unroll 1
    for (int t2 = 0; t2 < (NT + 1) / 2; ++t2)
      if ((use >> (2 * t2)) & 3u) {
        const uint4 bits = gct_philox(a.rng, grow, (uint32_t)(4 * t2 + g), 0xA4093822u, 0x299F31D0u);
        const uint32_t byte = ((bits.x & 0xffffu) >= th ? 1u : 0u) | ((bits.x >> 16) >= th ? 2u : 0u) |
                              ((bits.y & 0xffffu) >= th ? 4u : 0u) | ((bits.y >> 16) >= th ? 8u : 0u) |
                              ((bits.z & 0xffffu) >= th ? 16u : 0u) | ((bits.z >> 16) >= th ? 32u : 0u) |
                              ((bits.w & 0xffffu) >= th ? 64u : 0u) | ((bits.w >> 16) >= th ? 128u : 0u);
        keep |= (uint64_t)byte << (8 * t2);
      }
  }
  return keep;
}

__device__ __forceinline__ float score_of(float s, bool in_range, bool visible, float masked = -1e9f) {
  return in_range ? (visible ? s : masked) : -INFINITY;
}

// ------------------------------------------------------------------------------ forward
// One query tile (16 rows) of one pair: S^T, softmax, P.V.  bq = this lane's RAW Q fragment (the 1/sqrt(dk) is
// applied to the scores, after the product, like the reference's `matmul(q, k^T) / sqrt(d_k)`); mw = the packed
// mask words of this lane's query row.
template <int NDT, int NT, bool PROBS, typename AfterS>
__device__ __forceinline__ void fwd_unit(const AttnArgs& a, int b, int h, int u, float4 (&bq)[NDT],
                                         const uint32_t (&mw)[(NT + 1) / 2], const float* Ks, const float* Vs, int nkt,
                                         int lane, AfterS after_s) {
  constexpr int DK = 16 * NDT, SD = DK + 4, MW = (NT + 1) / 2;
  const int g = lane >> 4, c16 = lane & 15;
  const int q = 16 * u + c16;
  const uint32_t use = tiles_for_q<MW>(mw, q, a.Lq, a.Lk, nkt);
  uint32_t rowvis = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) rowvis |= mw[w] & range_word(w, a.Lk);
  f32x4 sacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // S^T[k][q] = sum_d K[k][d] * Q[q][d]
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if ((use >> t) & 1u) {
      float4 ak[NDT];
      row_frag<NDT>(ak, Ks, 16 * t + c16, g);
      sacc[t] = dot_frag<NDT>(ak, bq, sacc[t]);
    }
  after_s();          // bq is dead from here on: the caller may reload it (next pair's rows)
  // scale + mask + softmax over keys (regs x tiles in-lane, then lanes l^16, l^32).  Tiles outside `use` hold
  // only masked keys of rows that see a key elsewhere: probability exactly 0, nothing to compute.
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if ((use >> t) & 1u) {
      const uint32_t nib = mw[t >> 1] >> ((t & 1) * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = score_of(sacc[t][r] * a.scale, 16 * t + 4 * g + r < a.Lk, (nib >> r) & 1u);
        sacc[t][r] = sv;
        m = fmaxf(m, sv);
      }
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (m == -INFINITY) m = 0.f;
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if ((use >> t) & 1u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(sacc[t][r] - m);
        sacc[t][r] = e;
        l += e;
      }
    }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const int64_t grow = ((int64_t)b * a.H + h) * a.Lq + q;
  // A row without a visible key is uniform over its Lk keys (every score -1e9).  -1e9 + log(Lk) is not
  // representable in fp32, so such a row stores log(Lk) and the backward scores its masked keys as 0, not -1e9.
  if (g == 0 && q < a.Lq) (a.lse + ((int64_t)b * a.H + h) * a.Lq)[q] = (rowvis ? m : 0.f) + __logf(l);
  const uint64_t keep = keep_bits_row<NT>(a, (uint32_t)grow, use, g);
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = sacc[t][r] * inv;
        const int k = 16 * t + 4 * g + r;
        // (this runtime branch also bounds the scheduling region: with it compiled out, hipcc 7.2 interleaves the loop
        //  with the P.V product below and spills ~200 dwords per lane)
        if (PROBS && a.probs && q < a.Lq && k < a.Lk) a.probs[grow * a.Lk + k] = p;
        sacc[t][r] = ((keep >> (4 * t + r)) & 1ull) ? p * a.keep_scale : 0.f;
      }
    }
  // O^T[d][q] = sum_k V[k][d] * Pdrop^T[k][q]
  f32x4 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if ((use >> t) & 1u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bp = sacc[t][r];
        const float* vrow = Vs + (16 * t + 4 * g + r) * SD + c16;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) oacc[dt] = mfma16(vrow[16 * dt], bp, oacc[dt]);
      }
    }
  if (q < a.Lq) {
    char* obase = reinterpret_cast<char*>(a.o + (int64_t)b * a.Lq * a.ldo + h * DK);             // wave-uniform
    const uint32_t ooff = (uint32_t)((q * a.ldo + 4 * g) * 4);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      *reinterpret_cast<float4*>(obase + ooff + 64 * dt) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
  }
}

template <int NDT, int NT, bool PIPE, int OCC>
__global__ __launch_bounds__(ATT_THREADS, OCC) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, SD = DK + 4, NW = ATT_THREADS / 64, MW = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int LQP = (a.Lq + 15) & ~15, LKP = (a.Lk + 15) & ~15, nkt = LKP / 16, nqt = LQP / 16;
  float* Ks = smem;
  float* Vs = Ks + LKP * SD;
  int pair = blockIdx.x;
  if (pair >= a.npairs) return;
  Stage<DK, NT> sk, sv;
  StageOff<DK, NT> kvo;                  // ldk == ldv (checked on the host): one offset set serves K and V
  kvo.init(a.ldk, a.Lk, tid);
  float4 bq[NDT];
  uint4 mraw[(MW + 3) / 4] = {};
  {
    const int b = pair / a.H, h = pair - b * a.H;
    const int64_t kr0 = a.kstart ? (int64_t)a.kstart[b] : (int64_t)b * a.Lk;
    sk.load_rows(a.k, a.ldk, kr0, h, kvo);
    sv.load_rows(a.v, a.ldv, kr0, h, kvo);
    row_frag_global<NDT>(bq, a.q, a.ldq, b, h, a.Lq, 16 * wave + c16, g);
    if (a.mbits) mask_row_raw<MW>(mraw, a, b, 16 * wave + c16);
  }
  ASTAMP_DECL;
  for (;;) {
    const int b = pair / a.H, h = pair - b * a.H;
    ASTAMP(0);                           // loop seam
    const int Lk_in = a.klen ? a.klen[b] : a.Lk;          // keys that exist as rows (the others are masked: zero rows)
    sk.store(Ks, Lk_in, LKP, tid);
    sv.store(Vs, Lk_in, LKP, tid);
    uint32_t mw[MW];
    mask_row_use<MW>(mw, mraw, a, 16 * wave + c16);
    ASTAMP(1);                           // wait for the staged K / V + LDS stores
    __syncthreads();
    ASTAMP(2);                           // barrier 1
    const int next = pair + (int)gridDim.x;
    const bool more = next < a.npairs;
    const int nb = more ? next / a.H : b, nh = more ? next - nb * a.H : h;
    // Next pair's operands: requested now, consumed at the top of the next iteration.  NOTHING loaded below this
    // point may be consumed before then (a wait on a younger load waits for these too).
    const int64_t nkr0 = a.kstart ? (int64_t)a.kstart[nb] : (int64_t)nb * a.Lk;
    if (PIPE && more) {
      sk.load_rows(a.k, a.ldk, nkr0, nh, kvo);
      sv.load_rows(a.v, a.ldv, nkr0, nh, kvo);
      if (a.mbits) mask_row_raw<MW>(mraw, a, nb, 16 * wave + c16);
    }
    ASTAMP(3);                           // prefetch issue
    for (int u = wave; u < nqt; u += NW) {
      if (u != wave) {                   // only when there are more query tiles than waves (L > 96)
        row_frag_global<NDT>(bq, a.q, a.ldq, b, h, a.Lq, 16 * u + c16, g);
        uint4 r2[(MW + 3) / 4] = {};
        if (a.mbits) mask_row_raw<MW>(r2, a, b, 16 * u + c16);
        mask_row_use<MW>(mw, r2, a, 16 * u + c16);
      }
      const bool last = u + NW >= nqt;
      fwd_unit<NDT, NT, true>(a, b, h, u, bq, mw, Ks, Vs, nkt, lane, [&]() {
        // S^T was bq's last use: this wave's Q rows of the next pair arrive during softmax and P.V
        if (PIPE && more && last) row_frag_global<NDT>(bq, a.q, a.ldq, nb, nh, a.Lq, 16 * wave + c16, g);
      });
    }
    if (PIPE && more && wave >= nqt) row_frag_global<NDT>(bq, a.q, a.ldq, nb, nh, a.Lq, 16 * wave + c16, g);
    ASTAMP(4);                           // compute
    if (!more) break;
    __syncthreads();                     // every wave is done reading Ks / Vs
    ASTAMP(5);                           // barrier 2
    if (!PIPE) {
      sk.load_rows(a.k, a.ldk, nkr0, nh, kvo);
      sv.load_rows(a.v, a.ldv, nkr0, nh, kvo);
      row_frag_global<NDT>(bq, a.q, a.ldq, nb, nh, a.Lq, 16 * wave + c16, g);
      if (a.mbits) mask_row_raw<MW>(mraw, a, nb, 16 * wave + c16);
    }
    pair = next;
  }
  ASTAMP_OUT;
}

// ------------------------------------------------------------------------------ forward, direct
// One WAVE per (pair, query tile), no LDS, no barrier: the K and V fragments of the key tiles the query tile can see
// come straight from L2 (a pair's K, V are 2 x L x 256 B and are read by its 5-6 query tiles within microseconds of each
// other), so a wave never waits for another wave and the hardware dispatcher balances the SIMDs -- the LDS kernel
// above puts 5-6 query tiles on the 4 SIMDs of a CU the same way for every pair and ends every pair on a barrier.
// Same arithmetic, same order of operations, same dropout bits as fwd_unit; used for L_k <= 96.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// NDT consecutive floats as a NATIVE vector (HIP's float4 is a struct: its copies are memcpy calls that keep whole
// operand arrays in scratch memory when they sit under a condition)
template <int NDT> struct VecN;
template <> struct VecN<4> { typedef f32x4 T; };
template <> struct VecN<2> { typedef f32x2 T; };
template <> struct VecN<1> { typedef float T; };
template <int NDT> __device__ __forceinline__ float vec_at(const typename VecN<NDT>::T& v, int i);
template <> __device__ __forceinline__ float vec_at<4>(const f32x4& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vec_at<2>(const f32x2& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vec_at<1>(const float& v, int) { return v; }

// A wave-private LDS tile [16][DK + 4].  Rows arrive from global memory COALESCED -- lane (c16, g), r = 0..3 holds row
// 4g + r, head columns NDT c16 .. (16 lanes cover one row's 64 NDT bytes) -- and leave as the row-per-lane fragments
// the products over the head dimension need.  A row-per-lane float4 load straight from global memory costs the texture
// addresser about one lane per cycle (PMC: 57 cycles of TA_BUSY per load instruction against 16 for a coalesced one,
// TA 57-65 % busy in the first version of these kernels).  One wave's LDS instructions execute in order, so put -> get
// needs no barrier; the compiler is told to keep the order.
template <int NDT>
struct WaveTile {
  static constexpr int SD = 16 * NDT + 4;
  typedef typename VecN<NDT>::T VT;
  float* t;
  __device__ __forceinline__ void put(const VT (&v)[4], int g, int c16) const {
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<VT*>(t + (4 * g + r) * SD + NDT * c16) = v[r];
  }
  __device__ __forceinline__ void get(float4 (&f)[NDT], int g, int c16) const {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NDT; ++j) f[j] = *reinterpret_cast<const float4*>(t + c16 * SD + g * 4 * NDT + 4 * j);
    __builtin_amdgcn_wave_barrier();
  }
};
// the coalesced load: rows row0 + min(4g + r, last) of a [rows][ld] head slice whose (row 0, column 0) is `base`
template <int NDT>
__device__ __forceinline__ void tile_load(typename VecN<NDT>::T (&v)[4], const char* base, int ld, int row0, int last,
                                          int g, int c16) {
  typedef typename VecN<NDT>::T VT;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = row0 + 4 * g + r;
    v[r] = *reinterpret_cast<const VT*>(base + (uint32_t)(((rr < last ? rr : last) * ld + NDT * c16) * 4));
  }
}

#ifndef GCT_FWD_NH          // A/B knobs: -DGCT_FWD_NH=2 -DGCT_FWD_OCC=5, -DGCT_FWD_NH=3 -DGCT_FWD_OCC=4
#define GCT_FWD_NH 1
#define GCT_FWD_OCC 6
#endif
template <int NDT, int NT>
__global__ __launch_bounds__(256, GCT_FWD_OCC) void attn_fwd_direct_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, MW = (NT + 1) / 2;
  typedef typename VecN<NDT>::T VT;
  const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
  const int nqt = (a.Lq + 15) >> 4, nkt = (a.Lk + 15) >> 4;
  const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (item >= a.npairs * nqt) return;
  ASTAMP_DECL;
  const int pair = item / nqt, u = item - pair * nqt;
  const int b = pair / a.H, h = pair - b * a.H;
  const int q = 16 * u + c16;
  // compact query rows (the decoder forward over its live rows): q / o rows of sample b start at cstart[b], only the
  // first nlive[b] exist; a tile without a row has no work
  const int Lq_e = a.qo_compact ? a.nlive[b] : a.Lq;
  const int64_t qr0 = a.qo_compact ? (int64_t)a.cstart[b] : (int64_t)b * a.Lq;
  if (16 * u >= Lq_e) return;
  const int Lk_in = a.klen ? a.klen[b] : a.Lk;            // keys that exist as rows; the others are masked keys
  const int64_t kr0 = a.kstart ? (int64_t)a.kstart[b] : (int64_t)b * a.Lk;
  const int klast = Lk_in > 0 ? Lk_in - 1 : 0;
  // one LDS tile per wave (17 KB per workgroup): K tile t+1 is written after tile t's fragment has been read -- the
  // wave's LDS instructions execute in order
  __shared__ __attribute__((aligned(16))) float tiles[4][16 * WaveTile<NDT>::SD];
  const int wv = threadIdx.x >> 6;
  const WaveTile<NDT> T0{tiles[wv]}, T1{tiles[wv]};
  // Q rows of this tile and the K rows of the visible key tiles arrive coalesced and become row-per-lane fragments in
  // the wave's LDS tiles (WaveTile); rows beyond the existing keys re-read the last one: their scores are masked
  VT tq[4];
  tile_load<NDT>(tq, reinterpret_cast<const char*>(a.q + qr0 * a.ldq + h * DK), a.ldq, 16 * u, Lq_e - 1, g, c16);
  uint4 mraw[(MW + 3) / 4] = {};
  if (a.mbits) mask_row_raw<MW>(mraw, a, b, q);
  const char* kbase = reinterpret_cast<const char*>(a.k + kr0 * a.ldk + h * DK);                  // wave-uniform
  VT tk[NT][4];
  // with the precomputed tile word (a scalar load) the K requests go out before the mask rows are back
  uint32_t use = 0;
  // K / V tiles are requested in batches of NH that reuse one set of registers.  NH = 1: 71 VGPRs, 6-7 waves per SIMD;
  // NH = 2: 87 VGPRs / 5 waves, +1-3 %; NH = 3: 103 / 4, +5 %; all six tiles at once: 150 VGPRs / 3 waves, +12 % --
  // what hides the L2 latency here is the other waves, not the depth of one wave's request queue
  constexpr int NH = GCT_FWD_NH;
  if (a.tbits) {
    use = __builtin_amdgcn_readfirstlane(a.tbits[(int64_t)b * a.tb_sb + u * a.tb_su]);
#pragma unroll
    for (int t = 0; t < NH; ++t)
      if ((use >> t) & 1u) tile_load<NDT>(tk[t], kbase, a.ldk, 16 * t, klast, g, c16);
  }
  uint32_t mw[MW];
  mask_row_use<MW>(mw, mraw, a, q);
  if (!a.tbits) {
    use = tiles_for_q<MW>(mw, q, a.Lq, a.Lk, nkt);
#pragma unroll
    for (int t = 0; t < NH; ++t)
      if ((use >> t) & 1u) tile_load<NDT>(tk[t], kbase, a.ldk, 16 * t, klast, g, c16);
  }
  ASTAMP(0);                           // Q / mask / K requests, mask arrival (visible tiles when not precomputed)
  uint32_t rowvis = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) rowvis |= mw[w] & range_word(w, a.Lk);
  float4 bq[NDT];
  T1.put(tq, g, c16);
  T1.get(bq, g, c16);
  ASTAMP(1);                           // K requests, Q arrival -> LDS -> fragment
  f32x4 sacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= NH && t % NH == 0) {      // next batch of K tiles, into the registers the previous one has left
#pragma unroll
      for (int t2 = t; t2 < t + NH && t2 < NT; ++t2)
        if ((use >> t2) & 1u) tile_load<NDT>(tk[t2], kbase, a.ldk, 16 * t2, klast, g, c16);
    }
    if ((use >> t) & 1u) {
      float4 ak[NDT];
      if (t & 1) {                     // alternate: tile t+1 is written while tile t is read
        T1.put(tk[t], g, c16);
        T1.get(ak, g, c16);
      } else {
        T0.put(tk[t], g, c16);
        T0.get(ak, g, c16);
      }
      sacc[t] = dot_frag<NDT>(ak, bq, sacc[t]);
    }
  }
  ASTAMP(2);                           // K arrival -> LDS -> fragments, S^T issued
  // V fragments, requested now and consumed after the softmax: lane (c16, g) holds V[16t + 4g + r][NDT c16 .. + NDT)
  // -- output tile dt of the P.V product covers the head columns {NDT m + dt}
  const char* vbase = reinterpret_cast<const char*>(a.v + kr0 * a.ldv + h * DK);                  // wave-uniform
  VT av[NT][4];
#pragma unroll
  for (int t = 0; t < NH; ++t)
    if ((use >> t) & 1u) tile_load<NDT>(av[t], vbase, a.ldv, 16 * t, klast, g, c16);
  ASTAMP(3);                           // V requests
  // scale + mask + softmax (as fwd_unit)
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if ((use >> t) & 1u) {
      const uint32_t nib = mw[t >> 1] >> ((t & 1) * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = score_of(sacc[t][r] * a.scale, 16 * t + 4 * g + r < a.Lk, (nib >> r) & 1u);
        sacc[t][r] = sv;
        m = fmaxf(m, sv);
      }
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (m == -INFINITY) m = 0.f;
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if ((use >> t) & 1u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(sacc[t][r] - m);
        sacc[t][r] = e;
        l += e;
      }
    }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const int64_t grow = ((int64_t)b * a.H + h) * a.Lq + q;
  if (g == 0 && q < Lq_e) (a.lse + ((int64_t)b * a.H + h) * a.Lq)[q] = (rowvis ? m : 0.f) + __logf(l);
  uint64_t keep = keep_bits_row<NT>(a, (uint32_t)grow, use, g);
  if (a.klen) {                        // keys without a row: V is zero there (only a row that sees no key weighs them)
    uint64_t exist = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = Lk_in - 16 * t - 4 * g;
      exist |= (uint64_t)(n >= 4 ? 0xfu : (n > 0 ? ((1u << n) - 1u) : 0u)) << (4 * t);
    }
    keep &= exist;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = sacc[t][r] * inv;
        const int k = 16 * t + 4 * g + r;
        if (a.probs && q < a.Lq && k < a.Lk) a.probs[grow * a.Lk + k] = p;
        sacc[t][r] = ((keep >> (4 * t + r)) & 1ull) ? p * a.keep_scale : 0.f;
      }
    }
  ASTAMP(4);                           // S^T results, softmax, dropout bits
  // O^T[m][q] (tile dt) = sum_k V[k][NDT m + dt] * Pdrop^T[k][q]
  f32x4 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= NH && t % NH == 0) {      // next batch of V tiles
#pragma unroll
      for (int t2 = t; t2 < t + NH && t2 < NT; ++t2)
        if ((use >> t2) & 1u) tile_load<NDT>(av[t2], vbase, a.ldv, 16 * t2, klast, g, c16);
    }
    if ((use >> t) & 1u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) oacc[dt] = mfma16(vec_at<NDT>(av[t][r], dt), sacc[t][r], oacc[dt]);
      }
    }
  }
  ASTAMP(5);                           // V arrival, P.V issued
  if (q < Lq_e) {
    char* obase = reinterpret_cast<char*>(a.o + qr0 * a.ldo + h * DK);                           // wave-uniform
    const uint32_t ooff = (uint32_t)((q * a.ldo + 4 * NDT * g) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // accumulator row i of group g is head column NDT (4g + i) + dt
      if constexpr (NDT == 4)
        *reinterpret_cast<float4*>(obase + ooff + 16 * i) = make_float4(oacc[0][i], oacc[1][i], oacc[2][i], oacc[3][i]);
      else if constexpr (NDT == 2)
        *reinterpret_cast<float2*>(obase + ooff + 8 * i) = make_float2(oacc[0][i], oacc[1][i]);
      else
        *reinterpret_cast<float*>(obase + ooff + 4 * i) = oacc[0][i];
    }
  }
  ASTAMP(6);                           // P.V results, store
  DSTAMP_OUT(item);
}

// ----------------------------------------------------------------------------- backward
template <int NDT, int NT>
__global__ __launch_bounds__(ATT_THREADS, (NT <= 8 ? 3 : 2)) void attn_bwd_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, SD = DK + 4, NW = ATT_THREADS / 64, MW = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int LQP = (a.Lq + 15) & ~15, LKP = (a.Lk + 15) & ~15, nkt = LKP / 16, nqt = LQP / 16;
  const int LMX = LQP > LKP ? LQP : LKP;
  // region 0 holds {K, V} in phase A and {Q, dO} in phase B
  float* R0 = smem;
  float* R1 = R0 + LMX * SD;
  float* lse_s = R1 + LMX * SD;                                   // [LQP]
  float* del_s = lse_s + LQP;                                     // [LQP]
  uint32_t* mb_s = reinterpret_cast<uint32_t*>(del_s + LQP);      // [LQP][MW] packed mask rows
  uint32_t* kp_s = mb_s + LQP * MW;                               // [LQP][MW] dropout keep bits (written by phase A)
  uint16_t* kp_h = reinterpret_cast<uint16_t*>(kp_s);             //   ... as one 16-bit half-word per (q, key tile)
  uint8_t* rowok = reinterpret_cast<uint8_t*>(kp_s + LQP * MW);   // [LQP] row sees a key (or is padding)
  uint8_t* rowlive = rowok + LQP;                                 // [LQP] dO row has a non-zero element
  StageOff<DK, NT> kvo;                  // ldk == ldv (checked on the host)
  kvo.init(a.ldk, a.Lk, tid);
  ASTAMP_DECL;
  for (int pair = blockIdx.x; pair < a.npairs; pair += (int)gridDim.x) {
    ASTAMP(0);                           // loop seam
    const int b = pair / a.H, h = pair - b * a.H;
    const int64_t lrow0 = ((int64_t)b * a.H + h) * a.Lq;
    const int Lq_e = a.nlive ? a.nlive[b] : a.Lq;                               // query rows that exist in dout / dq
    const int64_t drow0 = a.cstart ? (int64_t)a.cstart[b] : (int64_t)b * a.Lq;  // row of (b, 0) in dout / dq
    const int64_t kin0 = a.kstart ? (int64_t)a.kstart[b] : (int64_t)b * a.Lk;   // row of key 0 in k / v
    const int Lk_in = a.klen ? a.klen[b] : a.Lk;                                // keys that exist as rows of k / v
    const int Lk_e = a.kv_compact ? Lq_e : Lk_in;                               // key rows that exist in dk / dv
    const int64_t krow0 = a.kv_compact ? drow0 : kin0;
    // ---------------------------------------------------------------- phase A: K, V in LDS -> dQ
    // every global load of this phase is issued up front: this wave's Q / dO / O rows and lse ride along with the
    // K / V staging loads, so the MFMAs below start with everything on chip
    const int q0 = 16 * wave + c16;
    float4 bq[NDT], bd[NDT], bo[NDT];
    row_frag_global<NDT>(bq, a.q, a.ldq, b, h, a.Lq, q0, g);
    row_frag_rows<NDT>(bd, a.dout, a.ldo, drow0, h, Lq_e, q0, g);
    row_frag_global<NDT>(bo, a.o_in, a.ldo, b, h, a.Lq, q0, g);
    float lse0 = a.lse_in[lrow0 + (q0 < a.Lq ? q0 : a.Lq - 1)];
    {
      Stage<DK, NT> sk, sv;
      sk.load_rows(a.k, a.ldk, kin0, h, kvo);
      sv.load_rows(a.v, a.ldv, kin0, h, kvo);
      // packed mask rows and "row sees a key" (both phases read them from LDS)
      for (int q = tid; q < LQP; q += ATT_THREADS) {
        uint4 raw[(MW + 3) / 4] = {};
        if (a.mbits) mask_row_raw<MW>(raw, a, b, q);
        uint32_t mw[MW];
        mask_row_use<MW>(mw, raw, a, q);
        uint32_t vis = 0;
#pragma unroll
        for (int w = 0; w < MW; ++w) {
          mb_s[q * MW + w] = mw[w];
          vis |= mw[w] & range_word(w, a.Lk);
        }
        rowok[q] = (q >= a.Lq) || vis != 0;
      }
      sk.store(R0, Lk_in, LKP, tid);
      sv.store(R1, Lk_in, LKP, tid);
    }
    ASTAMP(1);                           // loads issued, mask rows, wait for K / V, LDS stores
    __syncthreads();
    ASTAMP(2);                           // barrier 1
    float4 bk[NDT], bv[NDT];
    bool tile_live = false;
    for (int u = wave; u < nqt; u += NW) {
      const int q = 16 * u + c16;
      const bool real = q < Lq_e;
      if (u != wave) {                   // more query tiles than waves (L > 96): load in place
        row_frag_global<NDT>(bq, a.q, a.ldq, b, h, a.Lq, q, g);
        row_frag_rows<NDT>(bd, a.dout, a.ldo, drow0, h, Lq_e, q, g);
        row_frag_global<NDT>(bo, a.o_in, a.ldo, b, h, a.Lq, q, g);
        lse0 = a.lse_in[lrow0 + (real ? q : a.Lq - 1)];
      }
      float del = 0.f;
      int nz = 0;
#pragma unroll
      for (int j = 0; j < NDT; ++j) {
        del += (bo[j].x * bd[j].x + bo[j].y * bd[j].y) + (bo[j].z * bd[j].z + bo[j].w * bd[j].w);
        nz |= (bd[j].x != 0.f) | (bd[j].y != 0.f) | (bd[j].z != 0.f) | (bd[j].w != 0.f);
      }
      del += __shfl_xor(del, 16, 64);
      del += __shfl_xor(del, 32, 64);
      nz |= __shfl_xor(nz, 16, 64);
      nz |= __shfl_xor(nz, 32, 64);
      del = real ? del : 0.f;            // rows beyond Lq hold a copy of row Lq-1: neutralise
      nz = real ? nz : 0;
      const float lse = real ? lse0 : 0.f;
      if (g == 0) {
        del_s[q] = del;
        lse_s[q] = lse;
        rowlive[q] = (uint8_t)nz;
      }
      tile_live = __any(nz);
      if (!tile_live) {                       // all 16 gradient rows are zero: dQ rows = 0, nothing else
        if (real) {
          char* dbase = reinterpret_cast<char*>(a.dq + drow0 * a.lddq + h * DK);
          const uint32_t doff = (uint32_t)((q * a.lddq + 4 * g) * 4);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<float4*>(dbase + doff + 64 * dt) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        continue;
      }
      uint32_t mw[MW];
#pragma unroll
      for (int w = 0; w < MW; ++w) mw[w] = mb_s[q * MW + w];
      const uint32_t use = tiles_for_q<MW>(mw, q, a.Lq, a.Lk, nkt);
      uint32_t rowvis = 0;
#pragma unroll
      for (int w = 0; w < MW; ++w) rowvis |= mw[w] & range_word(w, a.Lk);
      const float masked = rowvis ? -1e9f : 0.f;        // see the forward's lse note
      f32x4 sacc[NT];                       // ends up holding dS^T
      const int64_t grow = lrow0 + q;
      const uint64_t keepw = keep_bits_row<NT>(a, (uint32_t)grow, use, g);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((use >> t) & 1u) {
          f32x4 pacc = (f32x4){0.f, 0.f, 0.f, 0.f};
          {
            float4 ak[NDT];
            row_frag<NDT>(ak, R0, 16 * t + c16, g);
            sacc[t] = dot_frag<NDT>(ak, bq, sacc[t]);  // S^T (unscaled)
          }
          {
            float4 av[NDT];
            row_frag<NDT>(av, R1, 16 * t + c16, g);
            pacc = dot_frag<NDT>(av, bd, pacc);        // dP^T
          }
          const uint32_t nib = mw[t >> 1] >> ((t & 1) * 16 + 4 * g);
          const uint32_t keep = (uint32_t)(keepw >> (4 * t)) & 0xfu;
          if (a.thr) {
            // the keep bits of this (query row, 16 keys) go to LDS for phase B, which visits exactly the tiles
            // visited here (same skip predicates): gather the four lane groups' nibbles, one 16-bit store per row
            uint32_t hw = keep << (4 * g);
            hw |= __shfl_xor(hw, 16, 64);
            hw |= __shfl_xor(hw, 32, 64);
            if (g == 0) kp_h[q * (2 * MW) + t] = (uint16_t)hw;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool inr = 16 * t + 4 * g + r < a.Lk, vis = (nib >> r) & 1u;
            const float p = inr ? __expf(score_of(sacc[t][r] * a.scale, true, vis, masked) - lse) : 0.f;
            const float dpd = ((keep >> r) & 1u) ? pacc[r] * a.keep_scale : 0.f;
            sacc[t][r] = (inr && vis) ? p * (dpd - del) : 0.f;  // dS^T (masked_fill passes no grad)
          }
        }
      }
      f32x4 qacc[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) qacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if ((use >> t) & 1u) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float bs = sacc[t][r];
            const float* krow = R0 + (16 * t + 4 * g + r) * SD + c16;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) qacc[dt] = mfma16(krow[16 * dt], bs, qacc[dt]);
          }
        }
      if (real) {
        char* dbase = reinterpret_cast<char*>(a.dq + drow0 * a.lddq + h * DK);
        const uint32_t doff = (uint32_t)((q * a.lddq + 4 * g) * 4);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          *reinterpret_cast<float4*>(dbase + doff + 64 * dt) =
              make_float4(qacc[dt][0] * a.scale, qacc[dt][1] * a.scale, qacc[dt][2] * a.scale, qacc[dt][3] * a.scale);
      }
    }
    ASTAMP(3);                           // phase A
    constexpr bool FRAG_STAGE = NT <= NW;   // one query tile and one key tile per wave: stage through registers
    if (FRAG_STAGE) {
      // this wave's key tile of phase B: K / V rows are still in LDS
      row_frag<NDT>(bk, R0, 16 * wave + c16, g);
      row_frag<NDT>(bv, R1, 16 * wave + c16, g);
    }
    __syncthreads();      // K, V no longer needed; del_s / lse_s / rowlive / keep bits complete
    ASTAMP(4);                           // own K / V rows from LDS + barrier 2
    // ---------------------------------------------------------------- phase B: Q, dO in LDS -> dK, dV
    if (FRAG_STAGE) {
      // Q and dO cross HBM once: every wave still holds the rows of its query tile as fragments, in exactly the
      // layout row_frag reads back (dead tiles are never read by phase B; rows beyond Lq are zeroed)
      if (wave < nqt && tile_live) {
        const int q = 16 * wave + c16;
        const bool real = q < Lq_e;
#pragma unroll
        for (int j = 0; j < NDT; ++j) {
          *reinterpret_cast<float4*>(R0 + q * SD + g * 4 * NDT + 4 * j) = real ? bq[j] : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(R1 + q * SD + g * 4 * NDT + 4 * j) = real ? bd[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
      Stage<DK, NT> sq, sd;
      StageOff<DK, NT> qo, dof;
      qo.init(a.ldq, a.Lq, tid);
      dof.init(a.ldo, a.Lq, tid);
      sq.load(a.q, a.ldq, b, h, a.Lq, qo);           // second read of this pair's Q / dO: L2
      sd.load_rows(a.dout, a.ldo, drow0, h, dof);    // (rows beyond Lq_e belong to other samples: zeroed by store)
      row_frag_rows<NDT>(bk, a.k, a.ldk, kin0, h, Lk_in, 16 * wave + c16, g);
      row_frag_rows<NDT>(bv, a.v, a.ldv, kin0, h, Lk_in, 16 * wave + c16, g);
      sq.store(R0, Lq_e, LQP, tid);
      sd.store(R1, Lq_e, LQP, tid);
    }
    ASTAMP(5);                           // Q / dO fragments -> LDS
    __syncthreads();
    ASTAMP(6);                           // barrier 3
    for (int t = wave; t < nkt; t += NW) {
      const int k = 16 * t + c16;
      if (16 * t >= Lk_e) continue;        // compact self-attention: dead keys have no row (and no gradient)
      if (t != wave) {
        row_frag_rows<NDT>(bk, a.k, a.ldk, kin0, h, Lk_in, k, g);
        row_frag_rows<NDT>(bv, a.v, a.ldv, kin0, h, Lk_in, k, g);
      }
      f32x4 vacc[NDT], kacc[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        vacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        kacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const int kw = k >> 5, kb = k & 31;
      const bool inr = k < a.Lk;
#pragma unroll 1
      for (int u = 0; u < nqt; ++u) {
        if (!__any(rowlive[16 * u + c16] != 0)) continue;  // zero gradient rows: Pd^T dO = 0 and dS = 0
        uint32_t vis4 = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) vis4 |= ((mb_s[(16 * u + 4 * g + r) * MW + kw] >> kb) & 1u) << r;
        const bool ok = __all(rowok[16 * u + c16] != 0);
        if (ok && !__any(inr && vis4 != 0)) continue;       // fully masked tile: P = dS = 0 (phase A skipped it too)
        uint32_t keep4 = 0xfu;
        if (a.thr) {
          keep4 = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) keep4 |= ((kp_s[(16 * u + 4 * g + r) * MW + kw] >> kb) & 1u) << r;
        }
        f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, pa = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
          float4 aq[NDT], ad[NDT];
          row_frag<NDT>(aq, R0, 16 * u + c16, g);
          row_frag<NDT>(ad, R1, 16 * u + c16, g);
          sa = dot_frag<NDT>(aq, bk, sa);   // S[q][k] (unscaled)
          pa = dot_frag<NDT>(ad, bv, pa);   // dP[q][k]
        }
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = 16 * u + 4 * g + r;
          const bool vis = (vis4 >> r) & 1u, keep = (keep4 >> r) & 1u;
          const float p = inr ? __expf(score_of(sa[r] * a.scale, true, vis, rowok[qq] ? -1e9f : 0.f) - lse_s[qq]) : 0.f;
          const float dpd = keep ? pa[r] * a.keep_scale : 0.f;
          pd[r] = keep ? p * a.keep_scale : 0.f;
          ds[r] = (inr && vis) ? p * (dpd - del_s[qq]) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* dorow = R1 + (16 * u + 4 * g + r) * SD + c16;
          const float* qrow = R0 + (16 * u + 4 * g + r) * SD + c16;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            vacc[dt] = mfma16(dorow[16 * dt], pd[r], vacc[dt]);  // dV^T[d][k] += dO[q][d] Pd[q][k]
            kacc[dt] = mfma16(qrow[16 * dt], ds[r], kacc[dt]);   // dK^T[d][k] += Q[q][d] dS[q][k]
          }
        }
      }
      if (k < Lk_e) {
        char* vbase = reinterpret_cast<char*>(a.dv + krow0 * a.lddv + h * DK);
        char* kbase = reinterpret_cast<char*>(a.dk + krow0 * a.lddk + h * DK);
        const uint32_t voff = (uint32_t)((k * a.lddv + 4 * g) * 4), koff = (uint32_t)((k * a.lddk + 4 * g) * 4);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          *reinterpret_cast<float4*>(vbase + voff + 64 * dt) = make_float4(vacc[dt][0], vacc[dt][1], vacc[dt][2], vacc[dt][3]);
          *reinterpret_cast<float4*>(kbase + koff + 64 * dt) =
              make_float4(kacc[dt][0] * a.scale, kacc[dt][1] * a.scale, kacc[dt][2] * a.scale, kacc[dt][3] * a.scale);
        }
      }
    }
    ASTAMP(7);                           // phase B
    __syncthreads();      // before the next pair's staging overwrites the region
  }
  ASTAMP_OUT;
}

// ------------------------------------------------------------------------------ backward, direct
// The same decomposition as the forward: no LDS, no barriers, one wave per item, operands from L2.
//   launch 1, one wave per (pair, query tile):  S^T, dP^T -> dS^T -> dQ;  leaves {lse, delta, masked score, live} per
//             query row, the keep bits and a live byte per tile in the workspace
//   launch 2, one wave per (pair, key tile):    S, dP -> P_drop, dS -> dV, dK over the live query tiles
// Arithmetic, skip predicates and dropout bits are those of attn_bwd_kernel's two phases.
template <int NDT>
struct DqBuf {                         // one key tile's operands of the dQ kernel, as they arrive (coalesced):
  typename VecN<NDT>::T kt[4], vt[4];  // K / V rows 16t + 4g + r, head columns NDT c16 ..; kt is the A operand of dQ^T
};

template <int NDT, int NT>
__global__ __launch_bounds__(256, 4) void attn_bwd_dq_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, MW = (NT + 1) / 2;
  const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
  const int nqt = (a.Lq + 15) >> 4, nkt = (a.Lk + 15) >> 4, LQP = 16 * nqt;
  const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (item >= a.npairs * nqt) return;
  ASTAMP_DECL;
  const int pair = item / nqt, u = item - pair * nqt;
  const int b = pair / a.H, h = pair - b * a.H;
  const int64_t lrow0 = (int64_t)pair * a.Lq;
  const int Lq_e = a.nlive ? a.nlive[b] : a.Lq;                               // query rows that exist in dout / dq
  const int64_t drow0 = a.cstart ? (int64_t)a.cstart[b] : (int64_t)b * a.Lq;  // row of (b, 0) in dout / dq
  const int64_t kin0 = a.kstart ? (int64_t)a.kstart[b] : (int64_t)b * a.Lk;   // row of key 0 in k / v
  const int Lk_in = a.klen ? a.klen[b] : a.Lk;                                // keys that exist as rows of k / v
  const int klast = Lk_in > 0 ? Lk_in - 1 : 0;
  uint32_t* visit = a.ws_use + (int64_t)pair * nqt + u;    // key tiles this query tile contributes to (launch 2)
  if (16 * u >= Lq_e) {                       // compacted away: no dout / dq rows
    if (lane == 0) *visit = 0;
    return;
  }
  const int q = 16 * u + c16;
  const bool real = q < Lq_e;
  // (five waves per SIMD -- one LDS tile, <= 102 VGPRs -- spills 32 registers and runs 1.7x slower: measured)
  __shared__ __attribute__((aligned(16))) float tiles[4][2][16 * WaveTile<NDT>::SD];
  const int wv = threadIdx.x >> 6;
  const WaveTile<NDT> T0{tiles[wv][0]}, T1{tiles[wv][1]};
  DqBuf<NDT> b0;
  float4 bq[NDT], bd[NDT];
  float del = 0.f;
  int nz = 0;
  float lse0;
  uint4 mraw[(MW + 3) / 4] = {};
  {
    // this tile's Q, dO and O rows: coalesced -> the wave's LDS tiles -> row-per-lane fragments
    float4 bo[NDT];
    const int64_t qrow0 = a.qo_compact ? drow0 : (int64_t)b * a.Lq;          // row of (b, 0) in q / o_in
    const int qlast = (a.qo_compact ? Lq_e : a.Lq) - 1;
    const char* qb = reinterpret_cast<const char*>(a.q + qrow0 * a.ldq + h * DK);
    const char* db = reinterpret_cast<const char*>(a.dout + drow0 * a.ldo + h * DK);
    const char* ob = reinterpret_cast<const char*>(a.o_in + qrow0 * a.ldo + h * DK);
    tile_load<NDT>(b0.kt, qb, a.ldq, 16 * u, qlast, g, c16);
    tile_load<NDT>(b0.vt, db, a.ldo, 16 * u, Lq_e - 1, g, c16);
    lse0 = a.lse_in[lrow0 + (q < a.Lq ? q : a.Lq - 1)];
    if (a.mbits) mask_row_raw<MW>(mraw, a, b, q);
    T0.put(b0.kt, g, c16);
    T0.get(bq, g, c16);
    T1.put(b0.vt, g, c16);
    tile_load<NDT>(b0.kt, ob, a.ldo, 16 * u, qlast, g, c16);
    T1.get(bd, g, c16);
    T0.put(b0.kt, g, c16);
    T0.get(bo, g, c16);
#pragma unroll
    for (int j = 0; j < NDT; ++j) {
      del += (bo[j].x * bd[j].x + bo[j].y * bd[j].y) + (bo[j].z * bd[j].z + bo[j].w * bd[j].w);
      nz |= (bd[j].x != 0.f) | (bd[j].y != 0.f) | (bd[j].z != 0.f) | (bd[j].w != 0.f);
    }
  }
  ASTAMP(0);                           // Q / dO / O requests -> arrival -> LDS -> fragments, delta partials
  del += __shfl_xor(del, 16, 64);
  del += __shfl_xor(del, 32, 64);
  nz |= __shfl_xor(nz, 16, 64);
  nz |= __shfl_xor(nz, 32, 64);
  del = real ? del : 0.f;              // rows beyond Lq_e hold a copy of the last row: neutralise
  nz = real ? nz : 0;
  const float lse = real ? lse0 : 0.f;
  uint32_t mw[MW];
  mask_row_use<MW>(mw, mraw, a, q);
  uint32_t rowvis = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) rowvis |= mw[w] & range_word(w, a.Lk);
  const float masked = rowvis ? -1e9f : 0.f;          // see the forward's lse note
  if (g == 0) a.ws_meta[(int64_t)pair * LQP + q] = make_float4(lse, del, masked, nz ? 1.f : 0.f);
  const bool tile_live = __any(nz);
  char* dbase = reinterpret_cast<char*>(a.dq + drow0 * a.lddq + h * DK);                          // wave-uniform
  const uint32_t doff = (uint32_t)((q * a.lddq + 4 * NDT * g) * 4);
  if (!tile_live) {                    // all 16 gradient rows are zero: dQ rows = 0, nothing else
    if (lane == 0) *visit = 0;
    if (real) {
#pragma unroll
      for (int i = 0; i < NDT; ++i) *reinterpret_cast<float4*>(dbase + doff + 16 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const uint32_t use = a.tbits ? __builtin_amdgcn_readfirstlane(a.tbits[(int64_t)b * a.tb_sb + u * a.tb_su])
                               : tiles_for_q<MW>(mw, q, a.Lq, a.Lk, nkt);
  if (lane == 0) *visit = use;
  const uint64_t keepw = keep_bits_row<NT>(a, (uint32_t)(lrow0 + q), use, g);
  ASTAMP(1);                           // mask arrival, row scalars, visible tiles, dropout bits
  uint16_t* kp_h = reinterpret_cast<uint16_t*>(a.ws_keep + ((int64_t)pair * LQP + q) * MW);
  const char* kbase = reinterpret_cast<const char*>(a.k + kin0 * a.ldk + h * DK);                 // wave-uniform
  const char* vbase = reinterpret_cast<const char*>(a.v + kin0 * a.ldv + h * DK);
  // the tile loop is ROLLED over the visible tiles and dS^T of a tile goes straight into the dQ product, so nothing is
  // kept per tile: 118 VGPRs, four waves per SIMD.  (A two-deep register prefetch of the next tile's operands measured
  // no faster at three waves per SIMD: what hides the load latency here is the other waves.)
  auto load = [&](DqBuf<NDT>& B_, int t) {
    tile_load<NDT>(B_.kt, kbase, a.ldk, 16 * t, klast, g, c16);
    tile_load<NDT>(B_.vt, vbase, a.ldv, 16 * t, klast, g, c16);
  };
  f32x4 qacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) qacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const DqBuf<NDT>& B_, int t) {
    ASTAMP(2);                         // (K / V requests issued)
    f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, pacc = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      float4 ak[NDT];                            // K / V rows 16t + c16 as row-per-lane fragments, one after the other
      T0.put(B_.kt, g, c16);
      T0.get(ak, g, c16);
      sacc = dot_frag<NDT>(ak, bq, sacc);        // S^T (unscaled)
      T1.put(B_.vt, g, c16);
      T1.get(ak, g, c16);
      pacc = dot_frag<NDT>(ak, bd, pacc);        // dP^T
    }
    ASTAMP(3);                         // K / V arrival -> LDS -> fragments, S^T / dP^T issued
    uint32_t word = mw[0];
#pragma unroll
    for (int w = 1; w < MW; ++w) word = (t >> 1) == w ? mw[w] : word;
    const uint32_t nib = word >> ((t & 1) * 16 + 4 * g);
    const uint32_t keep = (uint32_t)(keepw >> (4 * t)) & 0xfu;
    if (a.thr) {                       // keep bits of (query row, 16 keys) for launch 2: one 16-bit store per row
      uint32_t hw = keep << (4 * g);
      hw |= __shfl_xor(hw, 16, 64);
      hw |= __shfl_xor(hw, 32, 64);
      if (g == 0) kp_h[t] = (uint16_t)hw;
    }
    float ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool inr = 16 * t + 4 * g + r < a.Lk, vis = (nib >> r) & 1u;
      const float p = inr ? __expf(score_of(sacc[r] * a.scale, true, vis, masked) - lse) : 0.f;
      const float dpd = ((keep >> r) & 1u) ? pacc[r] * a.keep_scale : 0.f;
      ds[r] = (inr && vis) ? p * (dpd - del) : 0.f;            // dS^T (masked_fill passes no grad)
    }
    ASTAMP(4);                         // S^T / dP^T results, dS arithmetic
    // dQ^T[m][q] (tile dt) += sum_k K[k][NDT m + dt] dS^T[k][q]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) qacc[dt] = mfma16(vec_at<NDT>(B_.kt[r], dt), ds[r], qacc[dt]);
    }
    ASTAMP(5);                         // dQ issued
  };
  for (uint32_t rem = use; rem; rem &= rem - 1) {       // wave-uniform
    const int t0 = __builtin_ctz(rem);
    load(b0, t0);
    compute(b0, t0);
  }
  if (real) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // accumulator row i of group g is head column NDT (4g + i) + dt
      if constexpr (NDT == 4)
        *reinterpret_cast<float4*>(dbase + doff + 16 * i) =
            make_float4(qacc[0][i] * a.scale, qacc[1][i] * a.scale, qacc[2][i] * a.scale, qacc[3][i] * a.scale);
      else if constexpr (NDT == 2)
        *reinterpret_cast<float2*>(dbase + doff + 8 * i) = make_float2(qacc[0][i] * a.scale, qacc[1][i] * a.scale);
      else
        *reinterpret_cast<float*>(dbase + doff + 4 * i) = qacc[0][i] * a.scale;
    }
  }
  ASTAMP(6);                           // dQ results, store
  DSTAMP_OUT_AT(item, 4096);          // (records 4096.. : launch 2 writes 0..4095)
}

template <int NDT>
struct DkvBuf {                        // one query tile's operands of the dK / dV kernel
  typename VecN<NDT>::T tq[4], td[4];  // Q / dO rows 16u + 4g + r, head columns NDT c16 ..: A operands of dK^T / dV^T
};

template <int NDT, int NT>
__global__ __launch_bounds__(256, 4) void attn_bwd_dkv_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, MW = (NT + 1) / 2;
  const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
  const int nqt = (a.Lq + 15) >> 4, nkt = (a.Lk + 15) >> 4, LQP = 16 * nqt;
  const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (item >= a.npairs * nkt) return;
  ASTAMP_DECL;
  const int pair = item / nkt, t = item - pair * nkt;
  const int b = pair / a.H, h = pair - b * a.H;
  const int Lq_e = a.nlive ? a.nlive[b] : a.Lq;
  const int64_t drow0 = a.cstart ? (int64_t)a.cstart[b] : (int64_t)b * a.Lq;
  const int64_t kin0 = a.kstart ? (int64_t)a.kstart[b] : (int64_t)b * a.Lk;
  const int Lk_in = a.klen ? a.klen[b] : a.Lk;
  const int Lk_e = a.kv_compact ? Lq_e : Lk_in;                               // key rows that exist in dk / dv
  const int64_t krow0 = a.kv_compact ? drow0 : kin0;
  if (16 * t >= Lk_e) return;          // dead keys have no row (and no gradient)
  const int k = 16 * t + c16;
  const bool inr = k < a.Lk;
  // the query tiles that computed this key tile in launch 1 (same skip predicates by construction)
  uint32_t rem = 0;
  {
    const uint32_t* vw = a.ws_use + (int64_t)pair * nqt;
    const int Lq_x = Lq_e < a.Lq ? Lq_e : a.Lq;
    const int nqt_e = (Lq_x + 15) >> 4;
    for (int u = 0; u < nqt_e; ++u) rem |= ((vw[u] >> t) & 1u) << u;
    rem = __builtin_amdgcn_readfirstlane(rem);
  }
  __shared__ __attribute__((aligned(16))) float tiles[4][2][16 * WaveTile<NDT>::SD];
  const int wv = threadIdx.x >> 6;
  const WaveTile<NDT> T0{tiles[wv][0]}, T1{tiles[wv][1]};
  float4 bk[NDT], bv[NDT];
  DkvBuf<NDT> b0;
  {
    DkvBuf<NDT>& kv = b0;
    const char* kb0 = reinterpret_cast<const char*>(a.k + kin0 * a.ldk + h * DK);
    const char* vb0 = reinterpret_cast<const char*>(a.v + kin0 * a.ldv + h * DK);
    const int klast = Lk_in > 0 ? Lk_in - 1 : 0;
    tile_load<NDT>(kv.tq, kb0, a.ldk, 16 * t, klast, g, c16);
    tile_load<NDT>(kv.td, vb0, a.ldv, 16 * t, klast, g, c16);
    T0.put(kv.tq, g, c16);
    T1.put(kv.td, g, c16);
    T0.get(bk, g, c16);
    T1.get(bv, g, c16);
  }
  ASTAMP(0);                           // visit list, K / V requests -> arrival -> LDS -> fragments
  f32x4 vacc[NDT], kacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    vacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    kacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int kw = t >> 1, kb = (t & 1) * 16 + c16;
  const float4* meta = a.ws_meta + (int64_t)pair * LQP;
  const uint32_t* kp = a.ws_keep + (int64_t)pair * LQP * MW + kw;
  const uint32_t* mb = a.mbits ? a.mbits + (int64_t)b * a.mb_sb + kw : nullptr;
  const int64_t qrow0 = a.qo_compact ? drow0 : (int64_t)b * a.Lq;            // row of (b, 0) in q
  const int qlast = (a.qo_compact ? Lq_e : a.Lq) - 1;
  const char* qbase = reinterpret_cast<const char*>(a.q + qrow0 * a.ldq + h * DK);                // wave-uniform
  const char* dbase = reinterpret_cast<const char*>(a.dout + drow0 * a.ldo + h * DK);
  auto load = [&](DkvBuf<NDT>& B_, int u) {
    tile_load<NDT>(B_.tq, qbase, a.ldq, 16 * u, qlast, g, c16);
    tile_load<NDT>(B_.td, dbase, a.ldo, 16 * u, Lq_e - 1, g, c16);
  };
  auto compute = [&](const DkvBuf<NDT>& B_, int u) {
    // per-row scalars of rows 16u + 4g + r ({lse, delta, masked score, live}, mask / keep word of this key tile):
    // small, L2-resident, requested here and consumed after the S / dP products
    // ... one request per kind: lane (g, c16) asks for row 4g + (c16 & 3), so every quad of lanes holds the four rows
    // of its group, and a quad broadcast (DPP, no memory traffic) hands each lane all four
    float4 mt[4];
    uint32_t mword[4], kword[4];
    {
      const int qq = 16 * u + 4 * g + (c16 & 3);
      const int qc = qq < a.Lq ? qq : a.Lq - 1;
      const float4 m1 = meta[qq];
      const uint32_t w1 = mb ? mb[qc * a.mb_sq] : 0xffffffffu;
      const uint32_t k1 = a.thr ? kp[qq * MW] : 0xffffffffu;
#define GCT_QUAD(v, r) __builtin_amdgcn_mov_dpp((int)(v), (r) * 0x55, 0xf, 0xf, true)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mt[r].x = __int_as_float(r == 0 ? GCT_QUAD(__float_as_int(m1.x), 0) : r == 1 ? GCT_QUAD(__float_as_int(m1.x), 1)
                               : r == 2 ? GCT_QUAD(__float_as_int(m1.x), 2) : GCT_QUAD(__float_as_int(m1.x), 3));
        mt[r].y = __int_as_float(r == 0 ? GCT_QUAD(__float_as_int(m1.y), 0) : r == 1 ? GCT_QUAD(__float_as_int(m1.y), 1)
                               : r == 2 ? GCT_QUAD(__float_as_int(m1.y), 2) : GCT_QUAD(__float_as_int(m1.y), 3));
        mt[r].z = __int_as_float(r == 0 ? GCT_QUAD(__float_as_int(m1.z), 0) : r == 1 ? GCT_QUAD(__float_as_int(m1.z), 1)
                               : r == 2 ? GCT_QUAD(__float_as_int(m1.z), 2) : GCT_QUAD(__float_as_int(m1.z), 3));
        mt[r].w = 0.f;
        mword[r] = (uint32_t)(r == 0 ? GCT_QUAD(w1, 0) : r == 1 ? GCT_QUAD(w1, 1) : r == 2 ? GCT_QUAD(w1, 2) : GCT_QUAD(w1, 3));
        kword[r] = (uint32_t)(r == 0 ? GCT_QUAD(k1, 0) : r == 1 ? GCT_QUAD(k1, 1) : r == 2 ? GCT_QUAD(k1, 2) : GCT_QUAD(k1, 3));
      }
#undef GCT_QUAD
    }
    ASTAMP(1);                         // Q / dO tile + row scalar requests issued
    f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, pa = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      float4 aq[NDT];                    // the same rows as row-per-lane fragments, through the wave's LDS tiles
      T0.put(B_.tq, g, c16);
      T1.put(B_.td, g, c16);
      T0.get(aq, g, c16);
      sa = dot_frag<NDT>(aq, bk, sa);   // S[q][k] (unscaled)
      T1.get(aq, g, c16);
      pa = dot_frag<NDT>(aq, bv, pa);   // dP[q][k]
    }
    ASTAMP(2);                         // Q / dO arrival -> LDS -> fragments, S / dP issued
    float pd[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = 16 * u + 4 * g + r;
      // rows beyond Lq: their mask rows are all ones in launch 1, but nothing of theirs may reach dK / dV
      const bool vis = qq < a.Lq && ((mword[r] >> kb) & 1u), keep = (kword[r] >> kb) & 1u, rreal = qq < Lq_e;
      const float msk = (mt[r].z != 0.f || qq >= a.Lq) ? -1e9f : 0.f;
      const float p = (inr && rreal) ? __expf(score_of(sa[r] * a.scale, true, vis, msk) - mt[r].x) : 0.f;
      const float dpd = keep ? pa[r] * a.keep_scale : 0.f;
      pd[r] = keep ? p * a.keep_scale : 0.f;
      ds[r] = (inr && vis) ? p * (dpd - mt[r].y) : 0.f;
    }
    ASTAMP(3);                         // row scalars, S / dP results, P / dS arithmetic
    // the coalesced rows again, as the A operands of dK^T / dV^T: re-read from the LDS tiles they were put into (32
    // registers that would otherwise be held across the whole tile: 146 -> <= 128 VGPRs, four waves per SIMD)
    typename VecN<NDT>::T tq2[4], td2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tq2[r] = *reinterpret_cast<const typename VecN<NDT>::T*>(T0.t + (4 * g + r) * WaveTile<NDT>::SD + NDT * c16);
      td2[r] = *reinterpret_cast<const typename VecN<NDT>::T*>(T1.t + (4 * g + r) * WaveTile<NDT>::SD + NDT * c16);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        vacc[dt] = mfma16(vec_at<NDT>(td2[r], dt), pd[r], vacc[dt]);   // dV^T[d][k] += dO[q][d] Pd[q][k]
        kacc[dt] = mfma16(vec_at<NDT>(tq2[r], dt), ds[r], kacc[dt]);   // dK^T[d][k] += Q[q][d] dS[q][k]
      }
    }
    __builtin_amdgcn_wave_barrier();   // the next tile's put must stay behind these reads
    ASTAMP(4);                         // dV / dK issued
  };
  while (rem) {                        // wave-uniform
    const int u0 = __builtin_ctz(rem);
    rem &= rem - 1;
    load(b0, u0);
    compute(b0, u0);
  }
  if (k < Lk_e) {
    char* vb = reinterpret_cast<char*>(a.dv + krow0 * a.lddv + h * DK);
    char* kb_ = reinterpret_cast<char*>(a.dk + krow0 * a.lddk + h * DK);
    const uint32_t voff = (uint32_t)((k * a.lddv + 4 * NDT * g) * 4), koff = (uint32_t)((k * a.lddk + 4 * NDT * g) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (NDT == 4) {
        *reinterpret_cast<float4*>(vb + voff + 16 * i) = make_float4(vacc[0][i], vacc[1][i], vacc[2][i], vacc[3][i]);
        *reinterpret_cast<float4*>(kb_ + koff + 16 * i) =
            make_float4(kacc[0][i] * a.scale, kacc[1][i] * a.scale, kacc[2][i] * a.scale, kacc[3][i] * a.scale);
      } else if constexpr (NDT == 2) {
        *reinterpret_cast<float2*>(vb + voff + 8 * i) = make_float2(vacc[0][i], vacc[1][i]);
        *reinterpret_cast<float2*>(kb_ + koff + 8 * i) = make_float2(kacc[0][i] * a.scale, kacc[1][i] * a.scale);
      } else {
        *reinterpret_cast<float*>(vb + voff + 4 * i) = vacc[0][i];
        *reinterpret_cast<float*>(kb_ + koff + 4 * i) = kacc[0][i] * a.scale;
      }
    }
  }
  ASTAMP(5);                           // dV / dK results, store
  DSTAMP_OUT(item);
}

// one thread per packed word: bits[b][q][w] = OR_j (mask[b,q,32w+j] != 0) << j
__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t* __restrict__ mask, int64_t sb, int64_t sq, int B,
                                                        int rows, int Lk, uint32_t* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * rows * MASK_W) return;
  const int w = (int)(i % MASK_W);
  const int64_t br = i / MASK_W;
  const int q = (int)(br % rows), b = (int)(br / rows);
  const uint8_t* m = mask + (int64_t)b * sb + (int64_t)q * sq;
  uint32_t word = 0;
  for (int j = 0; j < 32; ++j) {
    const int k = 32 * w + j;
    if (k < Lk && m[k]) word |= 1u << j;
  }
  bits[i] = word;
}

// decoder self-attention mask straight from the token ids (reference Model/modules.py:47-58 without cond2dec):
// out[b][q][k] = (tok[b][k] != pad) & (k <= q) & (pad & 1) -- the reference multiplies its no-peek pattern by pad_idx
// and ANDs it with the bool pad mask, so only bit 0 of pad_idx survives.  One thread per four keys.
__global__ __launch_bounds__(256) void trg_mask_tokens_kernel(const int64_t* __restrict__ tok, int64_t ld, int64_t pad,
                                                              int B, int T, uint8_t* __restrict__ out) {
  const int64_t n = (int64_t)B * T * T;
  const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= n) return;
  const uint8_t on = (uint8_t)(pad & 1);
  uint8_t v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t e = e0 + j;
    v[j] = 0;
    if (e < n) {
      const int k = (int)(e % T);
      const int64_t bq = e / T;
      const int q = (int)(bq % T), b = (int)(bq / T);
      v[j] = (tok[(int64_t)b * ld + k] != pad && k <= q) ? on : (uint8_t)0;
    }
  }
  if (e0 + 3 < n) {
    *reinterpret_cast<uchar4*>(out + e0) = make_uchar4(v[0], v[1], v[2], v[3]);
  } else {
    for (int j = 0; j < 4 && e0 + j < n; ++j) out[e0 + j] = v[j];
  }
}

#ifdef GCT_STAMPS
unsigned long long* g_attn_stamps = nullptr;   // tools/attn_stamps.hip
#endif
// one wave per (batch, query tile): the word tiles_for_q would compute (bit t: key tile t must be visited)
__global__ __launch_bounds__(256) void mask_tiles_kernel(const uint32_t* __restrict__ bits, int mb_sb, int mb_sq, int B,
                                                         int ntr, int Lq, int Lk, uint32_t* __restrict__ tiles) {
  const int lane = threadIdx.x & 63, c16 = lane & 15;
  const int item = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (item >= B * ntr) return;
  const int b = item / ntr, u = item - b * ntr;
  // key-padding mask (one row per batch): every query row is that row, whatever Lq is
  const int q = mb_sq ? 16 * u + c16 : 0, lq = mb_sq ? Lq : 1;
  const int qq = q < lq ? q : lq - 1;
  uint32_t mw[MASK_W];
#pragma unroll
  for (int w = 0; w < MASK_W; ++w) mw[w] = q < lq ? bits[(int64_t)b * mb_sb + (int64_t)qq * mb_sq + w] : 0xffffffffu;
  const uint32_t use = tiles_for_q<MASK_W>(mw, q, lq, Lk, (Lk + 15) >> 4);
  if (lane == 0) tiles[item] = use;
}

int g_num_cus = 0;
int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cus = n;
  }
  return g_num_cus;
}

template <typename K>
int ensure_lds(K kernel, size_t lds) {
  if (lds <= 64 * 1024) return GCT_OK;
  // idempotent and cheap; done per launch so every instantiation / device is covered
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    gct_set_error("attention: cannot opt in to 160 KB LDS: %s", hipGetErrorString(e));
    return GCT_ERR_HIP;
  }
  return GCT_OK;
}

int check_common(const char* who, const float* q, int64_t ldq, const float* k, int64_t ldk,
                 const float* v, int64_t ldv, const uint32_t* mbits, int64_t mb_sb, int64_t mb_sq, int B, int H,
                 int Lq, int Lk, int dk, float p) {
  GCT_CHECK_ARG(q && k && v && B >= 0 && H > 0 && Lq > 0 && Lk > 0, "%s: bad args", who);
  GCT_CHECK_ARG(dk == 16 || dk == 32 || dk == 64, "%s: head dim %d unsupported (16/32/64)", who, dk);
  GCT_CHECK_ARG(Lq <= L_MAX && Lk <= L_MAX, "%s: sequence length > %d unsupported", who, L_MAX);
  GCT_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && gct_aligned16(q) &&
                    gct_aligned16(k) && gct_aligned16(v),
                "%s: q/k/v must be 16-B aligned with ld %% 4 == 0", who);
  GCT_CHECK_ARG(ldk == ldv, "%s: k and v must share one leading dimension (ldk %lld, ldv %lld)", who, (long long)ldk,
                (long long)ldv);
  GCT_CHECK_ARG((int64_t)(Lq > Lk ? Lq : Lk) * (ldq > ldk ? ldq : ldk) * 4 < (1ll << 31) && mb_sb < (1ll << 30),
                "%s: one (batch) slice exceeds 2 GiB", who);
  GCT_CHECK_ARG(!mbits || (gct_aligned16(mbits) && mb_sb % 4 == 0 && mb_sq % 4 == 0),
                "%s: packed mask rows must be 16-B aligned (use gct_attn_mask_pack)", who);
  GCT_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout p out of range", who);
  GCT_CHECK_ARG((int64_t)B * H <= INT32_MAX, "%s: too many (batch, head) pairs", who);
  return GCT_OK;
}

template <int NDT, int NT>
int launch_fwd(const AttnArgs& a, size_t lds, hipStream_t st) {
  static const bool lds_kernel = getenv("GCT_ATTN_FWD_LDS") != nullptr;     // A/B switch for benchmarks
  if constexpr (NT <= 6) {
    if (!lds_kernel) {
      const int64_t items = (int64_t)a.npairs * ((a.Lq + 15) / 16);
      hipLaunchKernelGGL((attn_fwd_direct_kernel<NDT, NT>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
      return GCT_OK;
    }
  }
  // persistent workgroups: as many as are resident together (2 per CU: 6 waves at <= 168 VGPRs)
  const int per_cu = NT > 8 ? 1 : 2;
  const int64_t want = (int64_t)num_cus() * per_cu;
  const unsigned grid = (unsigned)(a.npairs < want ? a.npairs : want);
  if constexpr (NT > 8) {
    int rc = ensure_lds(attn_fwd_kernel<NDT, NT, false, 2>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_fwd_kernel<NDT, NT, false, 2>), dim3(grid), dim3(ATT_THREADS), lds, st, a);
  } else {
    int rc = ensure_lds(attn_fwd_kernel<NDT, NT, true, 3>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_fwd_kernel<NDT, NT, true, 3>), dim3(grid), dim3(ATT_THREADS), lds, st, a);
  }
  return GCT_OK;
}
template <int NDT, int NT>
int launch_bwd(const AttnArgs& a, size_t lds, unsigned grid, hipStream_t st) {
  int rc = ensure_lds(attn_bwd_kernel<NDT, NT>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_kernel<NDT, NT>), dim3(grid), dim3(ATT_THREADS), lds, st, a);
  return GCT_OK;
}
template <int NT>
int launch_fwd_dk(int dk, const AttnArgs& a, size_t lds, hipStream_t st) {
  return dk == 64 ? launch_fwd<4, NT>(a, lds, st) : dk == 32 ? launch_fwd<2, NT>(a, lds, st) : launch_fwd<1, NT>(a, lds, st);
}
template <int NT>
int launch_bwd_dk(int dk, const AttnArgs& a, size_t lds, unsigned grid, hipStream_t st) {
  return dk == 64 ? launch_bwd<4, NT>(a, lds, grid, st) : dk == 32 ? launch_bwd<2, NT>(a, lds, grid, st)
                                                                   : launch_bwd<1, NT>(a, lds, grid, st);
}

}  // namespace

extern "C" int gct_attn_mask_pack(const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, int B, int Lq, int Lk,
                                  uint32_t* bits, uint32_t* tiles, void* stream) {
  GCT_CHECK_ARG(mask && bits && B >= 0 && Lq > 0 && Lk > 0 && Lk <= 32 * MASK_W && gct_aligned16(bits),
                "attn_mask_pack: bad args (Lk <= %d)", 32 * MASK_W);
  const int rows = mask_sq == 0 ? 1 : Lq;           // key-padding mask: one row per batch
  const int64_t n = (int64_t)B * rows * MASK_W;
  if (n == 0) return GCT_OK;
  hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask,
                     mask_sb, mask_sq, B, rows, Lk, bits);
  GCT_LAUNCH_CHECK("attn_mask_pack");
  if (tiles) {          // [B][1] for a key-padding mask, [B][ceil(Lq / 16)] otherwise
    const int ntr = mask_sq == 0 ? 1 : (Lq + 15) / 16;
    const int64_t items = (int64_t)B * ntr;
    hipLaunchKernelGGL(mask_tiles_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, bits,
                       rows * MASK_W, mask_sq == 0 ? 0 : MASK_W, B, ntr, Lq, Lk, tiles);
    GCT_LAUNCH_CHECK("attn_mask_pack (tiles)");
  }
  return GCT_OK;
}

extern "C" int gct_trg_mask_tokens(const int64_t* tokens, int64_t ld_tok, int64_t pad, int B, int T, uint8_t* out,
                                   void* stream) {
  GCT_CHECK_ARG(tokens && out && B >= 0 && T > 0 && ld_tok >= T && ((uintptr_t)out & 3u) == 0,
                "trg_mask_tokens: bad args");
  const int64_t n4 = ((int64_t)B * T * T + 3) / 4;
  if (n4 == 0) return GCT_OK;
  hipLaunchKernelGGL(trg_mask_tokens_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     tokens, ld_tok, pad, B, T, out);
  GCT_LAUNCH_CHECK("trg_mask_tokens");
  return GCT_OK;
}

extern "C" int gct_attn_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk,
                            const float* v, int64_t ldv, const uint32_t* mbits, int64_t mb_sb,
                            int64_t mb_sq, float* o, int64_t ldo, float* lse, float* probs, int B,
                            int H, int Lq, int Lk, int dk, float scale, float p, uint64_t seed,
                            uint32_t site, const int32_t* kstart, const int32_t* klen, const uint32_t* tbits,
                            int64_t tb_sb, int64_t tb_su, const int32_t* qstart, const int32_t* qlen, void* stream) {
  int rc = check_common("attn_fwd", q, ldq, k, ldk, v, ldv, mbits, mb_sb, mb_sq, B, H, Lq, Lk, dk, p);
  if (rc) return rc;
  GCT_CHECK_ARG(o && lse && ldo % 4 == 0 && gct_aligned16(o) && (int64_t)Lq * ldo * 4 < (1ll << 31), "attn_fwd: bad output");
  if (B == 0) return GCT_OK;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv;
  a.mbits = mbits; a.mb_sb = (int)mb_sb; a.mb_sq = (int)mb_sq;
  a.o = o; a.ldo = (int)ldo; a.lse = lse; a.probs = probs;
  a.tbits = tbits; a.tb_sb = (int)tb_sb; a.tb_su = (int)tb_su;
#ifdef GCT_STAMPS
  a.stamps = g_attn_stamps;
#endif
  GCT_CHECK_ARG((kstart == nullptr) == (klen == nullptr), "attn_fwd: kstart / klen go together");
  a.kstart = kstart; a.klen = klen;
  GCT_CHECK_ARG((qstart == nullptr) == (qlen == nullptr), "attn_fwd: qstart / qlen go together");
  GCT_CHECK_ARG(!qstart || (Lk <= 96 && !probs && getenv("GCT_ATTN_FWD_LDS") == nullptr),
                "attn_fwd: compact query rows need the direct kernel (Lk <= 96) and no probs output");
  a.cstart = qstart; a.nlive = qlen; a.qo_compact = qstart ? 1 : 0;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.npairs = B * H; a.scale = scale;
  a.thr = gct_drop_threshold(p); a.keep_scale = 1.0f / (1.0f - p); a.rng = gct_rng_make(seed, site);
  const int LKP = (Lk + 15) & ~15, SD = dk + 4;
  const int nt = LKP / 16;                      // query tiles beyond the wave count are looped
  const size_t lds = (size_t)(2 * LKP) * SD * 4;
  GCT_CHECK_ARG(lds <= 160 * 1024, "attn_fwd: needs %zu B of LDS", lds);
  hipStream_t st = (hipStream_t)stream;
  rc = nt <= 6 ? launch_fwd_dk<6>(dk, a, lds, st) : nt <= 8 ? launch_fwd_dk<8>(dk, a, lds, st)
                                                            : launch_fwd_dk<13>(dk, a, lds, st);
  if (rc) return rc;
  GCT_LAUNCH_CHECK("attn_fwd");
  return GCT_OK;
}

// workspace of the direct backward kernels (L_k <= 96): per (pair, padded query row) a float4 record and 3 keep words,
// per (pair, query tile) one word; without it (or beyond 96 keys) gct_attn_bwd runs the LDS kernel
extern "C" int64_t gct_attn_bwd_ws_bytes(int B, int H, int Lq, int Lk) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || Lk > 96) return 0;
  const int64_t LQP = (Lq + 15) & ~15, rows = (int64_t)B * H * LQP;
  return rows * 16 + rows * 3 * 4 + (((int64_t)B * H * (LQP / 16) * 4 + 15) & ~(int64_t)15);
}

extern "C" int gct_attn_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk,
                            const float* v, int64_t ldv, const uint32_t* mbits, int64_t mb_sb,
                            int64_t mb_sq, const float* o, const float* dout, int64_t ldo,
                            const float* lse, float* dq, int64_t lddq, float* dk_,
                            int64_t lddk, float* dv, int64_t lddv, int B, int H, int Lq, int Lk,
                            int dk, float scale, float p, uint64_t seed, uint32_t site,
                            const int32_t* cstart, const int32_t* nlive, int kv_compact,
                            const int32_t* kstart, const int32_t* klen, const uint32_t* tbits, int64_t tb_sb,
                            int64_t tb_su, void* ws, int64_t ws_bytes, void* stream) {
  int rc = check_common("attn_bwd", q, ldq, k, ldk, v, ldv, mbits, mb_sb, mb_sq, B, H, Lq, Lk, dk, p);
  if (rc) return rc;
  GCT_CHECK_ARG(o && dout && lse && dq && dk_ && dv, "attn_bwd: null pointer");
  GCT_CHECK_ARG(ldo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 &&
                    gct_aligned16(o) && gct_aligned16(dout) && gct_aligned16(dq) &&
                    gct_aligned16(dk_) && gct_aligned16(dv),
                "attn_bwd: buffers must be 16-B aligned with ld %% 4 == 0");
  GCT_CHECK_ARG((int64_t)Lq * (ldo > lddq ? ldo : lddq) * 4 < (1ll << 31) && (int64_t)Lk * (lddk > lddv ? lddk : lddv) * 4 < (1ll << 31),
                "attn_bwd: one (batch) slice exceeds 2 GiB");
  if (B == 0) return GCT_OK;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv;
  a.mbits = mbits; a.mb_sb = (int)mb_sb; a.mb_sq = (int)mb_sq;
  a.o_in = o; a.dout = dout; a.ldo = (int)ldo; a.lse_in = lse;
  a.tbits = tbits; a.tb_sb = (int)tb_sb; a.tb_su = (int)tb_su;
  a.dq = dq; a.dk = dk_; a.dv = dv; a.lddq = (int)lddq; a.lddk = (int)lddk; a.lddv = (int)lddv;
#ifdef GCT_STAMPS
  a.stamps = g_attn_stamps;
#endif
  // kv_compact: bit 0 = dk / dv in the compact query rows (self-attention, k / v in the forward's layout);
  //             bit 1 = q and o are compact like dout / dq (the forward itself ran on the compact rows)
  const int qo_compact = (kv_compact >> 1) & 1;
  kv_compact &= 1;
  GCT_CHECK_ARG((cstart == nullptr) == (nlive == nullptr) && (!kv_compact || (cstart && Lq == Lk)),
                "attn_bwd: cstart / nlive go together; kv_compact needs them and Lq == Lk");
  GCT_CHECK_ARG(!qo_compact || cstart, "attn_bwd: compact q / o rows need cstart / nlive");
  a.cstart = cstart; a.nlive = nlive; a.kv_compact = kv_compact; a.qo_compact = qo_compact;
  GCT_CHECK_ARG((kstart == nullptr) == (klen == nullptr) && !(kstart && kv_compact),
                "attn_bwd: kstart / klen go together and exclude kv_compact");
  a.kstart = kstart; a.klen = klen;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.npairs = B * H; a.scale = scale;
  a.thr = gct_drop_threshold(p); a.keep_scale = 1.0f / (1.0f - p); a.rng = gct_rng_make(seed, site);
  const int LQP = (Lq + 15) & ~15, LKP = (Lk + 15) & ~15, SD = dk + 4;
  const int LMX = LQP > LKP ? LQP : LKP, nt = LMX / 16;
  const int MW = nt <= 6 ? 3 : nt <= 8 ? 4 : 7;
  hipStream_t st = (hipStream_t)stream;
  static const bool lds_kernel = getenv("GCT_ATTN_BWD_LDS") != nullptr;       // A/B switch for benchmarks
  if (LKP <= 96 && ws && !lds_kernel && gct_aligned16(ws) && ws_bytes >= gct_attn_bwd_ws_bytes(B, H, Lq, Lk)) {
    // direct kernels: one wave per (pair, query tile), then one wave per (pair, key tile)
    const int64_t rows = (int64_t)a.npairs * LQP;
    a.ws_meta = reinterpret_cast<float4*>(ws);
    a.ws_keep = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + rows * 16);
    a.ws_use = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + rows * 16 + rows * 3 * 4);
    const int64_t items_q = (int64_t)a.npairs * (LQP / 16), items_k = (int64_t)a.npairs * (LKP / 16);
    GCT_CHECK_ARG((items_q + 3) / 4 <= INT32_MAX && (items_k + 3) / 4 <= INT32_MAX, "attn_bwd: grid too large");
    const dim3 gq((unsigned)((items_q + 3) / 4)), gk((unsigned)((items_k + 3) / 4));
    if (dk == 64) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<4, 6>), gq, dim3(256), 0, st, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<4, 6>), gk, dim3(256), 0, st, a);
    } else if (dk == 32) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<2, 6>), gq, dim3(256), 0, st, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, 6>), gk, dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<1, 6>), gq, dim3(256), 0, st, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<1, 6>), gk, dim3(256), 0, st, a);
    }
    GCT_LAUNCH_CHECK("attn_bwd (direct)");
    return GCT_OK;
  }
  if (qo_compact) {
    gct_set_error("attn_bwd: compact q / o rows need the direct kernels (Lk <= 96 and a workspace)");
    return GCT_ERR_ARG;
  }
  const size_t lds = (size_t)(2 * LMX) * SD * 4 + (size_t)LQP * 8 + (size_t)LQP * MW * 8 + (size_t)LQP * 2;
  GCT_CHECK_ARG(lds <= 160 * 1024, "attn_bwd: needs %zu B of LDS", lds);
  const int per_cu = lds <= 50 * 1024 ? 3 : lds <= 76 * 1024 ? 2 : 1;
  const int64_t want = (int64_t)num_cus() * per_cu * 2;            // a few pairs per workgroup keep the tail short
  const unsigned grid = (unsigned)(a.npairs < want ? a.npairs : want);
  rc = nt <= 6 ? launch_bwd_dk<6>(dk, a, lds, grid, st) : nt <= 8 ? launch_bwd_dk<8>(dk, a, lds, grid, st)
                                                                  : launch_bwd_dk<13>(dk, a, lds, grid, st);
  if (rc) return rc;
  GCT_LAUNCH_CHECK("attn_bwd");
  return GCT_OK;
}
