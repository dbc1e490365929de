// K4: multi-head attention core, forward and backward.
// Reference: Model/sublayers.py:29-41 attention():  softmax(q k^T / sqrt(dk) masked_fill(mask==0,
// -1e9)) -> dropout on the probabilities -> . v ; head split/merge of sublayers.py:64-69 is
// folded into the addressing (q/k/v are read in place from the fused projection buffer, o is
// written heads-merged).
//
// Machine mapping (gfx950): L <= 128 and dk <= 64, so a whole (batch, head) problem lives in one
// workgroup's LDS: Q (pre-scaled), K, V [L][dk+4] fp32 plus one flag byte per (q,k) holding
// {in-range, not-masked, dropout-keep}.  All products run on v_mfma_f32_16x16x4_f32 (exact fp32).
//  * The score tile is computed TRANSPOSED (S^T = K Q^T): its accumulator layout has the key
//    index on the registers and the query on the lane -- exactly the B-operand layout of the
//    following P.V product (which sums over keys).  Probabilities never leave registers and
//    never touch HBM; a softmax row is reduced over 4 regs x tiles in-lane plus two shuffles.
//  * 16x16 score tiles whose mask bytes are all zero (above the causal diagonal, beyond a
//    sample's length) are skipped -- decided from the staged flags, wave-uniform, and only
//    when every query row of the tile has at least one visible key, so the masked_fill(-1e9)
//    semantics (uniform row when everything is masked) stay exact.
//  * staging loads are issued in batches (fully unrolled, loads before LDS stores) and the
//    workgroup is 8 (fwd) / 12 (bwd) waves so HBM/L2 latency overlaps the MFMA phases.
//  * Backward recomputes probabilities from the saved log-sum-exp (no [B,H,L,L] tensor):
//      pass A (one wave per query tile):  S^T, dP^T -> dS^T -> dQ
//      pass B (one wave per key tile)  :  S, dP -> P_drop, dS -> dV, dK
//    both passes only read LDS, so they run CONCURRENTLY on different waves of the workgroup.
// Roofline: HBM-bound on q,k,v,o (+ gradients); 4.L.d flop/token ~ 3 % of the step's flops.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT_MAX = 8;  // L <= 128
constexpr int FWD_THREADS = 512, BWD_THREADS = 768;

struct AttnArgs {
  const float *q, *k, *v;
  int64_t ldq, ldk, ldv;
  const uint8_t* mask;
  int64_t mask_sb, mask_sq;
  float* o;
  int64_t ldo;
  float* lse;
  float* probs;
  // backward
  const float *o_in, *dout, *lse_in;
  float *dq, *dk, *dv;
  int64_t lddq, lddk, lddv;
  int B, H, Lq, Lk;
  float scale, keep_scale;
  uint32_t thr;
  GctRng rng;
#ifdef GCT_STAMPS
  unsigned long long* stamps;
#endif
};

#ifdef GCT_STAMPS
#define ASTAMP(i)                                                                      \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    unsigned long long t__;                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    seg[i] += t__ - tprev;                                                             \
    tprev = t__;                                                                       \
  } while (0)
#else
#define ASTAMP(i)
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Staging of a [B][L][ld] head slice into LDS [LP][SD] (zero padded, optional scale) is split
// in two halves so a kernel can put ALL its global loads (q, k, v, dO, mask bytes) in flight
// before the first LDS store: one exposed HBM/L2 latency per workgroup instead of one per
// tensor (s_memtime stamps: 14.7k -> ~5k cycles for the forward prologue).
template <int DK, int NTHR>
struct Stage {
  static constexpr int SD = DK + 4, C = DK / 4;
  static constexpr int ITERS = (16 * NT_MAX * C + NTHR - 1) / NTHR;
  float4 v[ITERS];
  __device__ __forceinline__ void load(const float* src, int64_t ld, int b, int h, int L, int tid) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * NTHR;
      const int r = idx / C, c = idx - r * C;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < L) v[it] = *reinterpret_cast<const float4*>(src + ((int64_t)b * L + r) * ld + h * DK + c * 4);
    }
  }
  __device__ __forceinline__ void store(float* dst, int LP, float scale, int tid) const {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * NTHR;
      const int r = idx / C, c = idx - r * C;
      if (r < LP) {
        float4 w = v[it];
        w.x *= scale; w.y *= scale; w.z *= scale; w.w *= scale;
        *reinterpret_cast<float4*>(dst + r * SD + c * 4) = w;
      }
    }
  }
};

// flags[q][k]: bit2 in range, bit0 not masked, bit1 dropout keep. One Philox call per (q, 4 keys).
template <int NTHR>
struct Flags {
  static constexpr int ITERS = (16 * NT_MAX * 4 * NT_MAX + NTHR - 1) / NTHR;
  uint32_t mv[ITERS];
  __device__ __forceinline__ void load(const AttnArgs& a, int b, int LQP, int LKP, int tid);
  __device__ __forceinline__ void build(uint32_t* flags32, const AttnArgs& a, int b, int h, int LQP,
                                        int LKP, int tid) const;
};

template <int NTHR>
__device__ __forceinline__ void Flags<NTHR>::load(const AttnArgs& a, int b, int LQP, int LKP, int tid) {
  const int KG = LKP / 4;
  // all mask bytes in flight (4 consecutive bytes per item, packed into one word)
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * NTHR;
    const int q = idx / KG, kg = idx - q * KG;
    uint32_t w = 0x01010101u;
    if (a.mask && q < a.Lq && idx < LQP * KG) {
      const uint8_t* mp = a.mask + (int64_t)b * a.mask_sb + (int64_t)q * a.mask_sq + kg * 4;
      w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (kg * 4 + e < a.Lk) w |= (mp[e] ? 1u : 0u) << (8 * e);
    }
    mv[it] = w;
  }
}

template <int NTHR>
__device__ __forceinline__ void Flags<NTHR>::build(uint32_t* flags32, const AttnArgs& a, int b, int h,
                                                   int LQP, int LKP, int tid) const {
  const int KG = LKP / 4;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * NTHR;
    if (idx >= LQP * KG) continue;
    const int q = idx / KG, kg = idx - q * KG;
    uint32_t w = 0;
    if (q < a.Lq) {
      uint4 bits = make_uint4(~0u, ~0u, ~0u, ~0u);
      if (a.thr)
        bits = gct_philox(a.rng, (uint32_t)(((int64_t)b * a.H + h) * a.Lq + q), (uint32_t)kg,
                          0xA4093822u, 0x299F31D0u);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kg * 4 + e < a.Lk) {
          uint32_t f = 4u | ((mv[it] >> (8 * e)) & 1u);
          if (gct_pick(bits, e) >= a.thr) f |= 2u;
          w |= f << (8 * e);
        }
      }
    }
    flags32[idx] = w;
  }
}

// rowok[q] = row q has at least one visible key (or is a padding row);
// tile_any[u*NT_MAX+t] = tile (u,t) has at least one visible (q,k).  Needs a barrier after.
template <int NTHR>
__device__ __forceinline__ void build_tile_maps(const uint32_t* flags32, uint8_t* rowok,
                                                uint8_t* tile_any, int Lq, int LQP, int LKP, int tid) {
  const int KG = LKP / 4, nkt = LKP / 16, nqt = LQP / 16;
  for (int q = tid; q < LQP; q += NTHR) {
    uint32_t any = 0;
    for (int kg = 0; kg < KG; ++kg) any |= flags32[q * KG + kg] & 0x01010101u;
    rowok[q] = (q >= Lq) || any != 0;
  }
  for (int ti = tid; ti < nqt * nkt; ti += NTHR) {
    const int u = ti / nkt, t = ti - u * nkt;
    uint32_t any = 0;
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) any |= flags32[(16 * u + r) * KG + 4 * t + j] & 0x01010101u;
    tile_any[u * NT_MAX + t] = any != 0;
  }
}

// Operand fragments for products that contract over the head dimension (S = Q K^T, dP = dO V^T):
// lane group g owns dk indices [g*DK/4, (g+1)*DK/4), so a lane's whole fragment of one row is
// NDT consecutive float4 -- NDT ds_read_b128 feed DK/4 MFMAs (any k permutation is legal when
// both operands use it).  Was: one ds_read_b32 round trip per MFMA (10.8k cycles per 80 MFMAs).
template <int NDT>
__device__ __forceinline__ void row_frag(float4 (&f)[NDT], const float* lds, int row, int g) {
  constexpr int SD = 16 * NDT + 4;
#pragma unroll
  for (int j = 0; j < NDT; ++j)
    f[j] = *reinterpret_cast<const float4*>(lds + row * SD + g * 4 * NDT + 4 * j);
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

// bit t set => key tile t must be computed for query tile u
__device__ __forceinline__ uint32_t tiles_for_q(const uint8_t* rowok, const uint8_t* tile_any, int u,
                                                int nkt, int c16) {
  const bool ok = __all(rowok[16 * u + c16] != 0);
  uint32_t use = 0;
  for (int t = 0; t < nkt; ++t)
    if (!ok || tile_any[u * NT_MAX + t]) use |= 1u << t;
  return __builtin_amdgcn_readfirstlane(use);
}

__device__ __forceinline__ float score_of(float s, uint32_t f) {
  return (f & 4u) ? ((f & 1u) ? s : -1e9f) : -INFINITY;
}

// ------------------------------------------------------------------------------ forward
template <int NDT>
__global__ __launch_bounds__(FWD_THREADS) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, SD = DK + 4, NW = FWD_THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int LQP = (a.Lq + 15) & ~15, LKP = (a.Lk + 15) & ~15, nkt = LKP / 16, KG = LKP / 4;
  // Q is only ever a B-operand fragment of its own query row: it is read straight from global
  // memory into registers (no LDS copy => 50 KB per workgroup at L=80, 3 workgroups per CU)
  float* Ks = smem;
  float* Vs = Ks + LKP * SD;
  uint32_t* flags32 = reinterpret_cast<uint32_t*>(Vs + LKP * SD);
  uint8_t* rowok = reinterpret_cast<uint8_t*>(flags32 + LQP * KG);
  uint8_t* tile_any = rowok + 16 * NT_MAX;
#ifdef GCT_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
  {
    Stage<DK, FWD_THREADS> sk, sv;
    Flags<FWD_THREADS> fl;
    sk.load(a.k, a.ldk, b, h, a.Lk, tid);
    sv.load(a.v, a.ldv, b, h, a.Lk, tid);
    fl.load(a, b, LQP, LKP, tid);
    sk.store(Ks, LKP, 1.0f, tid);
    sv.store(Vs, LKP, 1.0f, tid);
    ASTAMP(0);  // staging q,k,v
    fl.build(flags32, a, b, h, LQP, LKP, tid);
  }
  __syncthreads();
  ASTAMP(1);  // flags + barrier
  build_tile_maps<FWD_THREADS>(flags32, rowok, tile_any, a.Lq, LQP, LKP, tid);
  __syncthreads();
  ASTAMP(2);  // tile maps + barrier

  // this wave's query fragment (first tile) is requested before the barriers above would let
  // it be consumed, so its latency overlaps the K/V staging
  for (int u = wave; u < LQP / 16; u += NW) {
    const int q = 16 * u + c16;
    float4 bq[NDT];
#pragma unroll
    for (int j = 0; j < NDT; ++j) {
      bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < a.Lq) {
        bq[j] = *reinterpret_cast<const float4*>(a.q + ((int64_t)b * a.Lq + q) * a.ldq + h * DK +
                                                 g * 4 * NDT + 4 * j);
        bq[j].x *= a.scale; bq[j].y *= a.scale; bq[j].z *= a.scale; bq[j].w *= a.scale;
      }
    }
    const uint32_t use = tiles_for_q(rowok, tile_any, u, nkt, c16);
    f32x4 sacc[NT_MAX];
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t) sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // S^T[k][q] = sum_d K[k][d] * Qs[q][d]
    {
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t)
        if ((use >> t) & 1u) {
          float4 ak[NDT];
          row_frag<NDT>(ak, Ks, 16 * t + c16, g);
          sacc[t] = dot_frag<NDT>(ak, bq, sacc[t]);
        }
    }
    ASTAMP(3);  // S = K Q^T
    // mask + softmax over keys (regs x tiles in-lane, then lanes l^16, l^32)
    uint32_t fw[NT_MAX];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t) {
      fw[t] = 0;
      if (t < nkt) {
        fw[t] = flags32[q * KG + 4 * t + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv = score_of(sacc[t][r], (fw[t] >> (8 * r)) & 0xffu);
          sacc[t][r] = sv;
          m = fmaxf(m, sv);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (m == -INFINITY) m = 0.f;
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t)
      if (t < nkt) {
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
    if (g == 0 && q < a.Lq) a.lse[grow] = m + __logf(l);
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t)
      if (t < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = sacc[t][r] * inv;
          const int k = 16 * t + 4 * g + r;
          if (a.probs && q < a.Lq && k < a.Lk) a.probs[grow * a.Lk + k] = p;
          sacc[t][r] = ((fw[t] >> (8 * r)) & 2u) ? p * a.keep_scale : 0.f;
        }
      }
    ASTAMP(4);  // softmax
    // O^T[d][q] = sum_k V[k][d] * Pdrop^T[k][q]
    f32x4 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t)
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
      float* orow = a.o + ((int64_t)b * a.Lq + q) * a.ldo + h * DK + 4 * g;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        *reinterpret_cast<float4*>(orow + 16 * dt) =
            make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
    }
    ASTAMP(5);  // P.V + store
  }
#ifdef GCT_STAMPS
  if (a.stamps && lane == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * NW + wave) * 8 + i] = seg[i];
#endif
}

// ----------------------------------------------------------------------------- backward
template <int NDT>
__global__ __launch_bounds__(BWD_THREADS) void attn_bwd_kernel(const AttnArgs a) {
  constexpr int DK = 16 * NDT, SD = DK + 4, NW = BWD_THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int LQP = (a.Lq + 15) & ~15, LKP = (a.Lk + 15) & ~15, nkt = LKP / 16, nqt = LQP / 16,
            KG = LKP / 4;
  float* Qs = smem;               // scaled by `scale`
  float* Ks = Qs + LQP * SD;
  float* Vs = Ks + LKP * SD;
  float* dOs = Vs + LKP * SD;
  float* lse_s = dOs + LQP * SD;  // [LQP]
  float* del_s = lse_s + LQP;     // [LQP]
  uint32_t* flags32 = reinterpret_cast<uint32_t*>(del_s + LQP);
  const uint8_t* flags8 = reinterpret_cast<const uint8_t*>(flags32);
  uint8_t* rowok = reinterpret_cast<uint8_t*>(flags32 + LQP * KG);
  uint8_t* tile_any = rowok + 16 * NT_MAX;
  uint8_t* rowlive = tile_any + NT_MAX * NT_MAX;   // dO row has a non-zero element
#ifdef GCT_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
  {
    Stage<DK, BWD_THREADS> sq, sk, sv, sd;
    Flags<BWD_THREADS> fl;
    sq.load(a.q, a.ldq, b, h, a.Lq, tid);
    sk.load(a.k, a.ldk, b, h, a.Lk, tid);
    sv.load(a.v, a.ldv, b, h, a.Lk, tid);
    sd.load(a.dout, a.ldo, b, h, a.Lq, tid);
    fl.load(a, b, LQP, LKP, tid);
    sq.store(Qs, LQP, a.scale, tid);
    sk.store(Ks, LKP, 1.0f, tid);
    sv.store(Vs, LKP, 1.0f, tid);
    sd.store(dOs, LQP, 1.0f, tid);
    fl.build(flags32, a, b, h, LQP, LKP, tid);
  }
  ASTAMP(0);  // staging + flags
  // delta[q] = sum_d dO[q][d] * O[q][d]  (16 lanes per row)
  // rowlive[q]: the incoming gradient row is not identically zero.  A query tile whose 16 rows are all
  // zero (padded target positions under an ignore_index loss) contributes exactly nothing: dP = 0,
  // delta = 0 => dS = 0 => dQ rows = 0 and no dK / dV contribution -- such tiles are skipped below.
  for (int r0 = tid >> 4; r0 < LQP; r0 += BWD_THREADS / 16) {
    float acc = 0.f;
    int nz = 0;
    if (r0 < a.Lq) {
      const float* orow = a.o_in + ((int64_t)b * a.Lq + r0) * a.ldo + h * DK;
      const float* drow = a.dout + ((int64_t)b * a.Lq + r0) * a.ldo + h * DK;
      for (int c = (tid & 15) * 4; c < DK; c += 64) {
        const float4 x = *reinterpret_cast<const float4*>(orow + c);
        const float4 y = *reinterpret_cast<const float4*>(drow + c);
        acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        nz |= (y.x != 0.f) | (y.y != 0.f) | (y.z != 0.f) | (y.w != 0.f);
      }
    }
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    nz |= __shfl_xor(nz, 8, 64);
    nz |= __shfl_xor(nz, 4, 64);
    nz |= __shfl_xor(nz, 2, 64);
    nz |= __shfl_xor(nz, 1, 64);
    if ((tid & 15) == 0) {
      del_s[r0] = acc;
      lse_s[r0] = r0 < a.Lq ? a.lse_in[((int64_t)b * a.H + h) * a.Lq + r0] : 0.f;
      rowlive[r0] = (uint8_t)nz;
    }
  }
  __syncthreads();
  ASTAMP(1);  // delta + barrier
  build_tile_maps<BWD_THREADS>(flags32, rowok, tile_any, a.Lq, LQP, LKP, tid);
  __syncthreads();
  ASTAMP(2);  // maps + barrier

  // work units 0..nqt-1 = pass A (query tiles), nqt..nqt+nkt-1 = pass B (key tiles)
  for (int unit = wave; unit < nqt + nkt; unit += NW) {
    if (unit < nqt) {
      // ---- pass A: dQ for query tile u; key index on registers.
      const int u = unit, q = 16 * u + c16;
      if (!__any(rowlive[q] != 0)) {            // all 16 gradient rows are zero: dQ rows = 0, nothing else
        if (q < a.Lq) {
          float* drow = a.dq + ((int64_t)b * a.Lq + q) * a.lddq + h * DK + 4 * g;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
            *reinterpret_cast<float4*>(drow + 16 * dt) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        ASTAMP(3);
        continue;
      }
      const uint32_t use = tiles_for_q(rowok, tile_any, u, nkt, c16);
      f32x4 sacc[NT_MAX], pacc[NT_MAX];
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      {
        float4 bq[NDT], bd[NDT];
        row_frag<NDT>(bq, Qs, q, g);
        row_frag<NDT>(bd, dOs, q, g);
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t)
          if ((use >> t) & 1u) {
            float4 ak[NDT], av[NDT];
            row_frag<NDT>(ak, Ks, 16 * t + c16, g);
            row_frag<NDT>(av, Vs, 16 * t + c16, g);
            sacc[t] = dot_frag<NDT>(ak, bq, sacc[t]);  // S^T
            pacc[t] = dot_frag<NDT>(av, bd, pacc[t]);  // dP^T
          }
      }
      const float lse = lse_s[q], del = del_s[q];
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t)
        if ((use >> t) & 1u) {
          const uint32_t w = flags32[q * KG + 4 * t + g];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t f = (w >> (8 * r)) & 0xffu;
            const float p = (f & 4u) ? __expf(score_of(sacc[t][r], f) - lse) : 0.f;
            const float dpd = (f & 2u) ? pacc[t][r] * a.keep_scale : 0.f;
            sacc[t][r] = (f & 1u) ? p * (dpd - del) : 0.f;  // dS^T (masked_fill passes no grad)
          }
        }
      f32x4 qacc[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) qacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t)
        if ((use >> t) & 1u) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float bs = sacc[t][r];
            const float* krow = Ks + (16 * t + 4 * g + r) * SD + c16;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) qacc[dt] = mfma16(krow[16 * dt], bs, qacc[dt]);
          }
        }
      if (q < a.Lq) {
        float* drow = a.dq + ((int64_t)b * a.Lq + q) * a.lddq + h * DK + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          *reinterpret_cast<float4*>(drow + 16 * dt) =
              make_float4(qacc[dt][0] * a.scale, qacc[dt][1] * a.scale, qacc[dt][2] * a.scale,
                          qacc[dt][3] * a.scale);
      }
    } else {
      // ---- pass B: dK, dV for key tile t; query index on registers.
      const int t = unit - nqt, k = 16 * t + c16;
      f32x4 vacc[NDT], kacc[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        vacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        kacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      float4 bk[NDT], bv[NDT];
      row_frag<NDT>(bk, Ks, k, g);
      row_frag<NDT>(bv, Vs, k, g);
#pragma unroll 1
      for (int u = 0; u < nqt; ++u) {
        if (!__any(rowlive[16 * u + c16] != 0)) continue;  // zero gradient rows: Pd^T dO = 0 and dS = 0
        const bool ok = __all(rowok[16 * u + c16] != 0);
        if (ok && !tile_any[u * NT_MAX + t]) continue;  // fully masked tile: P = dS = 0
        f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, pa = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
          float4 aq[NDT], ad[NDT];
          row_frag<NDT>(aq, Qs, 16 * u + c16, g);
          row_frag<NDT>(ad, dOs, 16 * u + c16, g);
          sa = dot_frag<NDT>(aq, bk, sa);   // S[q][k]
          pa = dot_frag<NDT>(ad, bv, pa);   // dP[q][k]
        }
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = 16 * u + 4 * g + r;
          const uint32_t f = flags8[qq * LKP + k];
          const float p = (f & 4u) ? __expf(score_of(sa[r], f) - lse_s[qq]) : 0.f;
          const float dpd = (f & 2u) ? pa[r] * a.keep_scale : 0.f;
          pd[r] = (f & 2u) ? p * a.keep_scale : 0.f;
          ds[r] = (f & 1u) ? p * (dpd - del_s[qq]) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* dorow = dOs + (16 * u + 4 * g + r) * SD + c16;
          const float* qrow = Qs + (16 * u + 4 * g + r) * SD + c16;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            vacc[dt] = mfma16(dorow[16 * dt], pd[r], vacc[dt]);  // dV^T[d][k] += dO[q][d] Pd[q][k]
            kacc[dt] = mfma16(qrow[16 * dt], ds[r], kacc[dt]);   // dK^T[d][k] += Qs[q][d] dS[q][k]
          }
        }
      }
      if (k < a.Lk) {
        float* vrow = a.dv + ((int64_t)b * a.Lk + k) * a.lddv + h * DK + 4 * g;
        float* krow = a.dk + ((int64_t)b * a.Lk + k) * a.lddk + h * DK + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          *reinterpret_cast<float4*>(vrow + 16 * dt) =
              make_float4(vacc[dt][0], vacc[dt][1], vacc[dt][2], vacc[dt][3]);
          *reinterpret_cast<float4*>(krow + 16 * dt) =
              make_float4(kacc[dt][0], kacc[dt][1], kacc[dt][2], kacc[dt][3]);
        }
      }
    }
    ASTAMP(3);  // unit (pass A or pass B)
  }
#ifdef GCT_STAMPS
  if (a.stamps && lane == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * NW + wave) * 8 + i] = seg[i];
#endif
}

template <typename K>
int ensure_lds(K kernel, size_t lds, bool* done) {
  if (lds <= 64 * 1024 || *done) return GCT_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024);
  if (e != hipSuccess) {
    gct_set_error("attention: cannot opt in to 160 KB LDS: %s", hipGetErrorString(e));
    return GCT_ERR_HIP;
  }
  *done = true;
  return GCT_OK;
}

int check_common(const char* who, const float* q, int64_t ldq, const float* k, int64_t ldk,
                 const float* v, int64_t ldv, int B, int H, int Lq, int Lk, int dk, float p) {
  GCT_CHECK_ARG(q && k && v && B >= 0 && H > 0 && Lq > 0 && Lk > 0, "%s: bad args", who);
  GCT_CHECK_ARG(dk == 16 || dk == 32 || dk == 64, "%s: head dim %d unsupported (16/32/64)", who, dk);
  GCT_CHECK_ARG(Lq <= 16 * NT_MAX && Lk <= 16 * NT_MAX, "%s: sequence length > %d unsupported", who,
                16 * NT_MAX);
  GCT_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && gct_aligned16(q) &&
                    gct_aligned16(k) && gct_aligned16(v),
                "%s: q/k/v must be 16-B aligned with ld %% 4 == 0", who);
  GCT_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout p out of range", who);
  return GCT_OK;
}

constexpr size_t MAPS_BYTES = 16 * NT_MAX + NT_MAX * NT_MAX + 16 * NT_MAX;  // rowok + tile_any + (bwd) rowlive

}  // namespace

extern "C" int gct_attn_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk,
                            const float* v, int64_t ldv, const uint8_t* mask, int64_t mask_sb,
                            int64_t mask_sq, float* o, int64_t ldo, float* lse, float* probs, int B,
                            int H, int Lq, int Lk, int dk, float scale, float p, uint64_t seed,
                            uint32_t site, void* stream) {
  int rc = check_common("attn_fwd", q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dk, p);
  if (rc) return rc;
  GCT_CHECK_ARG(o && lse && ldo % 4 == 0 && gct_aligned16(o), "attn_fwd: bad output");
  if (B == 0) return GCT_OK;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = mask_sq;
  a.o = o; a.ldo = ldo; a.lse = lse; a.probs = probs;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  a.thr = gct_drop_threshold(p); a.keep_scale = 1.0f / (1.0f - p); a.rng = gct_rng_make(seed, site);
  const int LQP = (Lq + 15) & ~15, LKP = (Lk + 15) & ~15, SD = dk + 4;
  const size_t lds = (size_t)(2 * LKP) * SD * 4 + (size_t)LQP * LKP + MAPS_BYTES;
  dim3 grid((unsigned)(B * H)), block(FWD_THREADS);
  hipStream_t st = (hipStream_t)stream;
  static bool f4 = false, f2 = false, f1 = false;
  GCT_CHECK_ARG(lds <= 160 * 1024, "attn_fwd: needs %zu B of LDS", lds);
  if (dk == 64) { if ((rc = ensure_lds(attn_fwd_kernel<4>, lds, &f4))) return rc; hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, block, lds, st, a); }
  else if (dk == 32) { if ((rc = ensure_lds(attn_fwd_kernel<2>, lds, &f2))) return rc; hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, block, lds, st, a); }
  else { if ((rc = ensure_lds(attn_fwd_kernel<1>, lds, &f1))) return rc; hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, block, lds, st, a); }
  GCT_LAUNCH_CHECK("attn_fwd");
  return GCT_OK;
}

extern "C" int gct_attn_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk,
                            const float* v, int64_t ldv, const uint8_t* mask, int64_t mask_sb,
                            int64_t mask_sq, const float* o, const float* dout, int64_t ldo,
                            const float* lse, float* delta, float* dq, int64_t lddq, float* dk_,
                            int64_t lddk, float* dv, int64_t lddv, int B, int H, int Lq, int Lk,
                            int dk, float scale, float p, uint64_t seed, uint32_t site,
                            void* stream) {
  (void)delta;
  int rc = check_common("attn_bwd", q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, dk, p);
  if (rc) return rc;
  GCT_CHECK_ARG(o && dout && lse && dq && dk_ && dv, "attn_bwd: null pointer");
  GCT_CHECK_ARG(ldo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 &&
                    gct_aligned16(o) && gct_aligned16(dout) && gct_aligned16(dq) &&
                    gct_aligned16(dk_) && gct_aligned16(dv),
                "attn_bwd: buffers must be 16-B aligned with ld %% 4 == 0");
  if (B == 0) return GCT_OK;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = mask_sq;
  a.o_in = o; a.dout = dout; a.ldo = ldo; a.lse_in = lse;
  a.dq = dq; a.dk = dk_; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  a.thr = gct_drop_threshold(p); a.keep_scale = 1.0f / (1.0f - p); a.rng = gct_rng_make(seed, site);
  const int LQP = (Lq + 15) & ~15, LKP = (Lk + 15) & ~15, SD = dk + 4;
  const size_t lds = (size_t)(2 * LQP + 2 * LKP) * SD * 4 + (size_t)LQP * 8 + (size_t)LQP * LKP +
                     MAPS_BYTES;
  dim3 grid((unsigned)(B * H)), block(BWD_THREADS);
  hipStream_t st = (hipStream_t)stream;
  static bool f4 = false, f2 = false, f1 = false;
  GCT_CHECK_ARG(lds <= 160 * 1024, "attn_bwd: needs %zu B of LDS", lds);
  if (dk == 64) { if ((rc = ensure_lds(attn_bwd_kernel<4>, lds, &f4))) return rc; hipLaunchKernelGGL(attn_bwd_kernel<4>, grid, block, lds, st, a); }
  else if (dk == 32) { if ((rc = ensure_lds(attn_bwd_kernel<2>, lds, &f2))) return rc; hipLaunchKernelGGL(attn_bwd_kernel<2>, grid, block, lds, st, a); }
  else { if ((rc = ensure_lds(attn_bwd_kernel<1>, lds, &f1))) return rc; hipLaunchKernelGGL(attn_bwd_kernel<1>, grid, block, lds, st, a); }
  GCT_LAUNCH_CHECK("attn_bwd");
  return GCT_OK;
}
