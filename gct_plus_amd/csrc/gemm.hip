// K3: nn.Linear forward / dgrad / wgrad as one LDS-tiled fp32-input MFMA GEMM family.
//
// Reference call sites: Model/sublayers.py:54-59,64-66,70,81-88 ; Model/vaetf.py:81,133.
//
// Machine mapping (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact fp32 fma chains (parity with the fp32 reference),
//     64 FLOP/clk/SIMD => 157 TFLOP/s chip peak (the roofline these kernels are priced on).
//   * block tile 128x128x32, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles, 64 acc VGPRs.
//   * operands staged HBM -> VGPR -> LDS (two LDS buffers, register prefetch of tile t+1
//     issued before the MFMAs of tile t, written after them: one barrier per K-tile).
//   * K-contiguous operands use a 16-B-chunk XOR swizzle so the ds_read_b128 fragment
//     reads are bank-conflict free; row-contiguous operands are read with ds_read_b32.
//   * lane half h owns k = 16h..16h+15 of every 32-wide K-tile for BOTH operands, so each
//     lane's fragment is 16 contiguous floats (4 x ds_read_b128) -- any k permutation is
//     legal as long as A and B agree.
//   * 1-D grid with the bijective XCD remap: the N-tiles that share an A row-panel get
//     consecutive logical ids => same XCD L2.
//   * "segmented" operands: q/k/v (mu/log_var) weights stay separate checkpoint tensors,
//     the kernel selects the base pointer per 4-element chunk.
#include <limits.h>
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_FLOATS = BM * BK;  // 4096 floats = 16 KB per operand tile

// A segmented matrix is addressed as base + {0, d1, d2}[segment] + ...: ONE pointer plus integer
// element offsets (computed on the host).  Selecting among integer offsets compiles to
// v_cndmask; selecting among pointers made hipcc either reload the pointer from kernarg memory
// in front of every tile load or build an LDS lookup table and fall back to flat_load.
struct Seg3 {
  const float* p0;
  int64_t d1, d2;
};

struct GemmArgs {
  int64_t M, N, K;  // logical C[M][N] = A[M][K] B[K][N]
  Seg3 a;
  int64_t lda;
  int64_t a_nper;
  Seg3 b;
  int64_t ldb;
  int64_t b_nper;
  float* c0;
  int64_t c_d1, c_d2;
  int64_t ldc;
  int64_t c_nper;
  int64_t slab_stride;  // split-K: C written to c.p[0] + z*slab_stride
  int64_t ksplit;       // K range per split (multiple of BK), == K when no split
  int nsplit;
  int epi;              // fwd: GCT_EPI_* ; dgrad: 16 + GCT_DEPI_* ; slab: 32
  const float* bias0;  // nullptr: no bias
  int64_t bias_d1, bias_d2;
  const float* resid;
  float* pre;
  const float* pre_in;
  float keep_scale;
  uint32_t thr;
  GctRng rng;
  float* bias_slab;  // wgrad fast path: per-split column sums of dY, [nsplit][M] (nullptr: off)
  int stagger;       // s_sleep(127) repeats for the second resident workgroup of each CU
  int64_t row_base;  // rows in front of this launch's row 0 (a launch on a row range keeps the dropout coordinates)
  const int32_t* quad_map;  // rows are a quad compaction (csrc/liverows.hip): dropout coordinates of compact quad q are
                            // those of original quad quad_map[q] (nullptr: identity)
  int64_t pre_rows;         // > 0 (with quad_map): pre_in keeps the forward's row space (pre_rows rows) and is read
                            // through the quad map -- no gathered copy of the 4d-wide pre-activation is needed
  const int32_t* kt_list;   // bf16x6 wgrad: ascending 32-row K-tile indices to reduce over (nullptr: all)
  const int32_t* kt_count;  // device scalar: entries of kt_list
  const uint16_t* bp0;  // bf16x6 kernels: pre-split planes of the B operand (same element offsets as b)
  int64_t bp_stride;    // elements between the hi / mid / lo planes
#ifdef GCT_STAMPS
  unsigned long long* stamps;  // diagnostic build only (tools/gemm_stamps.hip)
#endif
};

enum { EPI_SLAB = 32, EPI_D0 = 16 };

// row of pre_in that belongs to row `row` (+0..3, row % 4 == 0) of this launch; -1: none (padding)
__device__ __forceinline__ int64_t pre_row0(const GemmArgs& g, int64_t row) {
  if (g.pre_rows <= 0 || !g.quad_map) return row;
  const int v = g.quad_map[(row + g.row_base) >> 2];
  return v < 0 ? -1 : (int64_t)v * 4;
}

// quad (row >> 2) whose Philox stream covers `row` of this launch
__device__ __forceinline__ uint32_t drop_quad(const GemmArgs& g, int64_t row) {
  int64_t q = (row + g.row_base) >> 2;
  if (g.quad_map) {
    const int v = g.quad_map[q];
    q = v < 0 ? 0 : v;                 // padding quads: their rows are zero whatever the mask says
  }
  return (uint32_t)q;
}

#ifdef GCT_STAMPS
#define STAMP(i)                                                                       \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    unsigned long long t__;                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    seg[i] += t__ - tprev;                                                             \
    tprev = t__;                                                                       \
  } while (0)
#else
#define STAMP(i)
#endif

__device__ __forceinline__ const float* seg_ptr(const Seg3& s, int64_t idx, int64_t nper,
                                                int64_t& local) {
  // nseg <= 3: two compares instead of a 64-bit division
  const bool g1 = idx >= nper, g2 = idx >= 2 * nper;
  local = idx - (g2 ? 2 * nper : (g1 ? nper : 0));
  return s.p0 + (g2 ? s.d2 : (g1 ? s.d1 : 0));
}

// ---- global -> register tile loaders ------------------------------------------------
// "KC": tile rows indexed by the free dim (m or n), K contiguous in memory.
//   AK: A[m][k] = a.p[k/nper] + m*lda + k%nper        (segmented along k)
//   BK_: B[k][n] = b.p[n/nper] + (n%nper)*ldb + k      (segmented along rows n)
template <bool VEC, bool SEG_ON_K>
__device__ __forceinline__ void load_kc(float4 (&r)[4], const Seg3& s, int64_t ld, int64_t nper,
                                        int64_t row0, int64_t rows, int64_t k0, int64_t kend,
                                        int tid) {
  const int c8 = tid & 7;
  const int64_t k = k0 + c8 * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = row0 + (tid >> 3) + 32 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
      if (VEC) {
        if (k < kend) {
          int64_t loc;
          const float* base;
          if (SEG_ON_K) {
            base = seg_ptr(s, k, nper, loc);
            v = *reinterpret_cast<const float4*>(base + row * ld + loc);
          } else {
            base = seg_ptr(s, row, nper, loc);
            v = *reinterpret_cast<const float4*>(base + loc * ld + k);
          }
        }
      } else {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k + e < kend) {
            int64_t loc;
            if (SEG_ON_K) {
              const float* base = seg_ptr(s, k + e, nper, loc);
              t[e] = base[row * ld + loc];
            } else {
              const float* base = seg_ptr(s, row, nper, loc);
              t[e] = base[loc * ld + k + e];
            }
          }
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
    r[i] = v;
  }
}

// "RC": tile = 32 k-rows x 128 contiguous free-dim columns.
//   AM: A[m][k] = a.p[m/nper] + k*lda + m%nper        (segmented along columns m)
//   BN_: B[k][n] = b.p[k/nper] + (k%nper)*ldb + n      (segmented along rows k)
template <bool VEC, bool SEG_ON_K>
__device__ __forceinline__ void load_rc(float4 (&r)[4], const Seg3& s, int64_t ld, int64_t nper,
                                        int64_t col0, int64_t cols, int64_t k0, int64_t kend,
                                        int tid) {
  const int c32 = tid & 31;
  const int64_t col = col0 + c32 * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t k = k0 + (tid >> 5) + 8 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend) {
      if (VEC) {
        if (col < cols) {
          int64_t loc;
          if (SEG_ON_K) {
            const float* base = seg_ptr(s, k, nper, loc);
            v = *reinterpret_cast<const float4*>(base + loc * ld + col);
          } else {
            const float* base = seg_ptr(s, col, nper, loc);
            v = *reinterpret_cast<const float4*>(base + k * ld + loc);
          }
        }
      } else {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col + e < cols) {
            int64_t loc;
            if (SEG_ON_K) {
              const float* base = seg_ptr(s, k, nper, loc);
              t[e] = base[loc * ld + col + e];
            } else {
              const float* base = seg_ptr(s, col + e, nper, loc);
              t[e] = base[k * ld + loc];
            }
          }
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
    r[i] = v;
  }
}

// ---- register -> LDS ------------------------------------------------------------------
__device__ __forceinline__ void store_kc(float* lds, const float4 (&r)[4], int tid) {
  const int c8 = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const int chunk = c8 ^ ((row >> 1) & 7);
    *reinterpret_cast<float4*>(lds + row * BK + chunk * 4) = r[i];
  }
}
__device__ __forceinline__ void store_rc(float* lds, const float4 (&r)[4], int tid) {
  const int c32 = tid & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (tid >> 5) + 8 * i;
    *reinterpret_cast<float4*>(lds + k * BM + c32 * 4) = r[i];
  }
}

// ---- LDS -> fragments, one QUARTER (4 k-values) at a time: f[t][e] holds k = 16*h + 4*c + e of
// free-dim index base+32t+(lane&31).  KC: one ds_read_b128 per t; RC: four ds_read_b32 per t.
__device__ __forceinline__ void fragq_kc(float (&f)[2][4], const float* lds, int base, int lane,
                                         int c) {
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = base + t * 32 + (lane & 31);
    const int sw = (row >> 1) & 7;
    const float4 v = *reinterpret_cast<const float4*>(lds + row * BK + (((h * 4 + c) ^ sw) * 4));
    f[t][0] = v.x; f[t][1] = v.y; f[t][2] = v.z; f[t][3] = v.w;
  }
}
__device__ __forceinline__ void fragq_rc(float (&f)[2][4], const float* lds, int base, int lane,
                                         int c) {
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = base + t * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[t][e] = lds[(h * 16 + c * 4 + e) * BM + col];
  }
}

// A_KC: A is [M][K] K-contiguous (fwd, dgrad); else [K][M] (wgrad).
// B_KC: B is weights [N][K] (fwd); else [K][N] (dgrad, wgrad).
template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[4 * TILE_FLOATS];  // A0 B0 A1 B1 : 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  const unsigned tiles_n = (unsigned)((g.N + BN - 1) / BN);
  const unsigned tiles_m = (unsigned)((g.M + BM - 1) / BM);
  const unsigned per_split = tiles_m * tiles_n;
  const unsigned lid = gct_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lid / per_split, rest = lid - z * per_split;
  const int64_t m0 = (int64_t)(rest / tiles_n) * BM, n0 = (int64_t)(rest % tiles_n) * BN;
  const int64_t kbeg = (int64_t)z * g.ksplit;
  const int64_t kend = (kbeg + g.ksplit < g.K) ? kbeg + g.ksplit : g.K;

  const Seg3 sa = g.a, sb = g.b;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  auto gload = [&](int64_t k0) {
    if (A_KC)
      load_kc<VEC, true>(ra, sa, g.lda, g.a_nper, m0, g.M, k0, kend, tid);
    else
      load_rc<VEC, false>(ra, sa, g.lda, g.a_nper, m0, g.M, k0, kend, tid);
    if (B_KC)
      load_kc<VEC, false>(rb, sb, g.ldb, g.b_nper, n0, g.N, k0, kend, tid);
    else
      load_rc<VEC, true>(rb, sb, g.ldb, g.b_nper, n0, g.N, k0, kend, tid);
  };
  auto lstore = [&](int buf) {
    float* la = lds + buf * 2 * TILE_FLOATS;
    float* lb = la + TILE_FLOATS;
    if (A_KC) store_kc(la, ra, tid); else store_rc(la, ra, tid);
    if (B_KC) store_kc(lb, rb, tid); else store_rc(lb, rb, tid);
  };

  const int64_t nkt = (kend - kbeg + BK - 1) / BK;
#ifdef GCT_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
  if (nkt > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  STAMP(0);  // prologue
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nkt) gload(kbeg + (kt + 1) * BK);  // in flight during the MFMAs below
    STAMP(1);  // global load issue
    const float* la = lds + cur * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
    // fragments are prefetched ONE QUARTER (16 MFMAs = 1024 cycles) ahead of their use, so the
    // LDS latency is always covered by a full MFMA group instead of two instructions
    float fa[2][2][4], fb[2][2][4];
    auto fragq = [&](int set, int c) {
      if (A_KC) fragq_kc(fa[set], la, wm, lane, c); else fragq_rc(fa[set], la, wm, lane, c);
      if (B_KC) fragq_kc(fb[set], lb, wn, lane, c); else fragq_rc(fb[set], lb, wn, lane, c);
    };
    fragq(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int set = c & 1;
      if (c < 3) fragq(set ^ 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][e], fb[set][j][e], acc[i][j],
                                                             0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(2);  // frag reads + 64 MFMAs
#ifdef GCT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(3);  // wait for the prefetched tile
#endif
    if (kt + 1 < nkt) lstore(cur ^ 1);
    STAMP(4);  // LDS stores
    __syncthreads();
    STAMP(5);  // barrier
  }

  // ---- epilogue: acc[i][j][r] -> C[row][col], col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn + j * 32 + (lane & 31);
      if (col >= g.N) continue;
      int64_t cloc;
      float* cbase;
      float bias = 0.f;
      if (g.epi == EPI_SLAB) {
        cbase = g.c0 + (int64_t)z * g.slab_stride;
        cloc = col;
      } else {
        const bool g1 = col >= g.c_nper, g2 = col >= 2 * g.c_nper;
        cloc = col - (g2 ? 2 * g.c_nper : (g1 ? g.c_nper : 0));
        cbase = g.c0 + (g2 ? g.c_d2 : (g1 ? g.c_d1 : 0));
        if (g.epi < EPI_D0 && g.bias0)
          bias = g.bias0[(g2 ? g.bias_d2 : (g1 ? g.bias_d1 : 0)) + cloc];
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int64_t row_base = m0 + wm + i * 32 + 8 * q4 + 4 * h;  // multiple of 4
        uint4 bits = make_uint4(~0u, ~0u, ~0u, ~0u);
        const bool need_rng = g.thr != 0u && (g.epi == GCT_EPI_GELU_DROP ||
                                              g.epi == GCT_EPI_DROP_RESID ||
                                              g.epi == EPI_D0 + GCT_DEPI_GELU_BWD);
        if (need_rng) bits = gct_drop_bits(g.rng, drop_quad(g, row_base), (uint32_t)col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t row = row_base + e;
          if (row >= g.M) continue;
          float v = acc[i][j][q4 * 4 + e];
          const int64_t off = row * g.ldc + cloc;
          const bool keep = gct_drop_keep(bits, e, (uint32_t)col, g.thr);
          switch (g.epi) {
            case GCT_EPI_BIAS:
              v += bias;
              break;
            case GCT_EPI_GELU_DROP: {
              v += bias;
              g.pre[off] = v;
              v = gct_gelu(v);
              v = keep ? v * g.keep_scale : 0.f;
            } break;
            case GCT_EPI_DROP_RESID: {
              v += bias;
              v = keep ? v * g.keep_scale : 0.f;
              v += g.resid[off];
            } break;
            case EPI_D0 + GCT_DEPI_ACCUM:
              v += cbase[off];
              break;
            case EPI_D0 + GCT_DEPI_GELU_BWD: {
              float u = 0.f;
              if (g.pre_rows > 0) {
                const int64_t er0 = pre_row0(g, row_base);
                if (er0 >= 0 && er0 + e < g.pre_rows) u = g.pre_in[(er0 + e) * g.ldc + cloc];
              } else {
                u = g.pre_in[off];
              }
              v = keep ? v * gct_gelu_grad(u) * g.keep_scale : 0.f;
            } break;
            default:
              break;  // STORE / SLAB
          }
          cbase[off] = v;
        }
      }
    }
  }
#ifdef GCT_STAMPS
  STAMP(6);  // epilogue
  if (g.stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 64) {
    for (int i = 0; i < 8; ++i) g.stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i] = seg[i];
  }
#endif
}


// =====================================================================================
// FAST PATH kernel: same tiling/pipeline as gemm_f32_kernel, for the shapes that matter
// (K % 32 == 0, 16-B aligned operands, segment boundaries aligned to the tile) with the two
// costs the s_memtime stamps exposed removed:
//   * tile loads: a wave-uniform 64-bit base (SGPRs, advanced per K-tile) + per-thread 32-bit
//     offsets computed ONCE; edge rows/columns are clamped to valid memory instead of being
//     predicated (they only feed accumulator rows/columns that are never stored) -- the loop
//     issues 8 global_load_dwordx4 back to back with no address arithmetic or branches;
//   * epilogue: each wave transposes its 64x64 accumulator block through LDS and every lane
//     finishes a 4-row x 4-column patch: float4 bias/resid/pre accesses, one Philox call per
//     column (4 rows each), float4 stores of 256 contiguous bytes per 16 lanes (was: 64 scalar
//     stores per lane behind a per-element switch, 42k cycles per tile).
// =====================================================================================
// epilogue stores: 16 B per lane, non-temporal -- the outputs (84-336 MB) are far larger than the L2
// and are not re-read by this kernel, so they should not evict the operand panels (A/B: +3-7 % on the
// K = 512 shapes, +1 % on the step; -DGCT_EPI_PLAIN restores ordinary stores)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void epi_store4(float* p, const float (&x)[4]) {
#ifdef GCT_LAB_NO_EPI_STORE   // tools/gemm_lab.hip: arithmetic kept alive, (almost) nothing stored
  if (x[0] != 123456.789f) return;
#endif
#ifdef GCT_EPI_PLAIN
  *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
#else
  __builtin_nontemporal_store(f32x4_t{x[0], x[1], x[2], x[3]}, reinterpret_cast<f32x4_t*>(p));
#endif
}

struct FastEpi {
  const GemmArgs& g;
  // number of extra float4 reads per patch row this epilogue needs (resid / accumulate / pre_in)
  __device__ __forceinline__ bool needs_extra() const {
#ifdef GCT_LAB_NO_EPI_MATH
    return false;
#endif
    return g.epi == GCT_EPI_DROP_RESID || g.epi == EPI_D0 + GCT_DEPI_ACCUM ||
           g.epi == EPI_D0 + GCT_DEPI_GELU_BWD;
  }
  __device__ __forceinline__ const float* extra_base(const float* cbase) const {
    return g.epi == GCT_EPI_DROP_RESID ? g.resid
                                       : (g.epi == EPI_D0 + GCT_DEPI_GELU_BWD ? g.pre_in : cbase);
  }
  // issue the extra reads of one 4x4 patch (latency overlaps the other patches' work)
  __device__ __forceinline__ void prefetch(float4 (&x)[4], int64_t row0, const float* cbase,
                                           int64_t cloc) const {
    const float* eb = extra_base(cbase);
    int64_t er0 = row0, elim = g.M;
    if (g.epi == EPI_D0 + GCT_DEPI_GELU_BWD && g.pre_rows > 0) {
      er0 = pre_row0(g, row0);
      elim = er0 < 0 ? -1 : g.pre_rows;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      x[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + rr < g.M && er0 + rr < elim) x[rr] = *reinterpret_cast<const float4*>(eb + (er0 + rr) * g.ldc + cloc);
    }
  }
  __device__ __forceinline__ void apply(float4 (&v)[4], int64_t row0, int64_t col0,
                                        float* cbase, int64_t cloc, float4 bias) const {
    float4 x[4];
    if (needs_extra()) prefetch(x, row0, cbase, cloc);
    apply(v, x, row0, col0, cbase, cloc, bias);
  }
  __device__ __forceinline__ void apply(float4 (&v)[4], const float4 (&ex)[4], int64_t row0,
                                        int64_t col0, float* cbase, int64_t cloc, float4 bias) const {
    // v[rr] = 4 consecutive columns (col0..col0+3) of row row0+rr; row0 % 4 == 0
#ifdef GCT_LAB_NO_EPI_MATH    // tools/gemm_lab.hip: raw accumulators
    const int epi = EPI_D0 + GCT_DEPI_STORE;
#else
    const int epi = g.epi;
#endif
    uint4 bits[2];   // one Philox call per 4 rows x 2 columns (col0 % 4 == 0)
    const bool rng = g.thr != 0u && (epi == GCT_EPI_GELU_DROP || epi == GCT_EPI_DROP_RESID ||
                                     epi == EPI_D0 + GCT_DEPI_GELU_BWD);
    if (rng) {
#pragma unroll
      for (int cp = 0; cp < 2; ++cp)
        bits[cp] = gct_drop_bits(g.rng, drop_quad(g, row0), (uint32_t)(col0 + 2 * cp));
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t row = row0 + rr;
      if (row >= g.M) break;
      const int64_t off = row * g.ldc + cloc;
      float x[4] = {v[rr].x, v[rr].y, v[rr].z, v[rr].w};
      const float bs[4] = {bias.x, bias.y, bias.z, bias.w};
      bool keep[4] = {true, true, true, true};
      if (rng) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) keep[cc] = gct_drop_keep(bits[cc >> 1], rr, (uint32_t)cc, g.thr);
      }
      if (epi == GCT_EPI_BIAS) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) x[cc] += bs[cc];
      } else if (epi == GCT_EPI_GELU_DROP) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) x[cc] += bs[cc];
        epi_store4(g.pre + off, x);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) x[cc] = keep[cc] ? gct_gelu(x[cc]) * g.keep_scale : 0.f;
      } else if (epi == GCT_EPI_DROP_RESID) {
        const float4 r = ex[rr];
        const float rs[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          x[cc] = (keep[cc] ? (x[cc] + bs[cc]) * g.keep_scale : 0.f) + rs[cc];
      } else if (epi == EPI_D0 + GCT_DEPI_ACCUM) {
        const float4 r = ex[rr];
        x[0] += r.x; x[1] += r.y; x[2] += r.z; x[3] += r.w;
      } else if (epi == EPI_D0 + GCT_DEPI_GELU_BWD) {
        const float4 u = ex[rr];
        const float us[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          x[cc] = keep[cc] ? x[cc] * gct_gelu_grad(us[cc]) * g.keep_scale : 0.f;
      }
      epi_store4(cbase + off, x);
    }
  }
};


// Per-wave epilogue shared by the 128x128 fp32-MFMA kernel and the bf16x6 kernels (both leave a
// 64x64 block per wave in the 32x32 MFMA accumulator layout): transpose through LDS (stg: 64x64
// floats private to the wave), then every lane finishes 4 patches of 4 rows x 4 columns.
__device__ __forceinline__ void wave_epilogue_tail(const GemmArgs& g, float* stg, int lane, int64_t mw,
                                                   int64_t nw, unsigned z);
__device__ __forceinline__ void wave_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][2], float* stg,
                                              int lane, int64_t mw, int64_t nw, unsigned z) {
  {
    const int h = lane >> 5, c32 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stg[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + c32] = acc[i][j][r];
  }
  wave_epilogue_tail(g, stg, lane, mw, nw, z);
}

// second part of the per-wave epilogue: the 64x64 block is in `stg` (row-major, this wave's own LDS writes)
__device__ __forceinline__ void wave_epilogue_tail(const GemmArgs& g, float* stg, int lane, int64_t mw,
                                                   int64_t nw, unsigned z) {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes are visible to it
  __builtin_amdgcn_wave_barrier();
  const FastEpi ep{g};
  // the lane's column chunk is the same for its 4 patches: destination, bias and the
  // segment select are resolved once; the residual / accumulate / pre-activation rows of
  // all 4 patches are requested up front so their latency overlaps
  const int c4 = lane & 15;
  const int64_t col0 = nw + c4 * 4;
  if (col0 < g.N) {
    float* cbase;
    int64_t cloc;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.epi == EPI_SLAB) {
      cbase = g.c0 + (int64_t)z * g.slab_stride;
      cloc = col0;
    } else {
      const bool g1 = col0 >= g.c_nper, g2 = col0 >= 2 * g.c_nper;
      cloc = col0 - (g2 ? 2 * g.c_nper : (g1 ? g.c_nper : 0));
      cbase = g.c0 + (g2 ? g.c_d2 : (g1 ? g.c_d1 : 0));
      if (g.epi < EPI_D0 && g.bias0)
        bias = *reinterpret_cast<const float4*>(g.bias0 + (g2 ? g.bias_d2 : (g1 ? g.bias_d1 : 0)) + cloc);
    }
    float4 ex[4][4];
    const bool extra = ep.needs_extra();
    if (extra) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        ep.prefetch(ex[it], mw + (it * 4 + (lane >> 4)) * 4, cbase, cloc);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rg = it * 4 + (lane >> 4);
      const int64_t row0 = mw + rg * 4;
      if (row0 >= g.M) continue;
      float4 v[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        v[rr] = *reinterpret_cast<const float4*>(stg + (rg * 4 + rr) * 64 + c4 * 4);
      ep.apply(v, ex[it], row0, col0, cbase, cloc, bias);
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void
gemm_f32_fast_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[4 * TILE_FLOATS];  // A0 B0 A1 B1 : 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // Two workgroups share a CU and, started together, stay in lockstep: their prologues and
  // epilogues coincide and the MFMA pipe idles.  Delaying the second resident workgroup of the
  // first dispatch round by about half a tile de-phases them for the rest of the launch (tiles
  // have equal duration, so the offset persists).  Placement is a speed guess only.
  if (g.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
    for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  const unsigned tiles_n = (unsigned)((g.N + BN - 1) / BN);
  const unsigned tiles_m = (unsigned)((g.M + BM - 1) / BM);
  const unsigned per_split = tiles_m * tiles_n;
  const unsigned lid = gct_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lid / per_split, rest = lid - z * per_split;
  const int64_t m0 = (int64_t)(rest / tiles_n) * BM, n0 = (int64_t)(rest % tiles_n) * BN;
  const int64_t kbeg = (int64_t)z * g.ksplit;
  const int64_t kend = (kbeg + g.ksplit < g.K) ? kbeg + g.ksplit : g.K;

  // ---- per-thread invariant 32-bit offsets (floats), edges clamped into valid memory
  uint32_t offa[4], offb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (A_KC) {
      int64_t r = (tid >> 3) + 32 * i;
      if (m0 + r > g.M - 1) r = g.M - 1 - m0;
      offa[i] = (uint32_t)(r * g.lda + (tid & 7) * 4);
    } else {
      int64_t c = (tid & 31) * 4;
      if (m0 + c > g.M - 4) c = g.M - 4 - m0;
      offa[i] = (uint32_t)(((tid >> 5) + 8 * i) * g.lda + c);
    }
    if (B_KC) {
      int64_t r = (tid >> 3) + 32 * i;
      if (n0 + r > g.N - 1) r = g.N - 1 - n0;
      offb[i] = (uint32_t)(r * g.ldb + (tid & 7) * 4);
    } else {
      int64_t c = (tid & 31) * 4;
      if (n0 + c > g.N - 4) c = g.N - 4 - n0;
      offb[i] = (uint32_t)(((tid >> 5) + 8 * i) * g.ldb + c);
    }
  }
  // ---- wave-uniform tile bases; a K-tile (resp. the row/column tile) lies in ONE segment
  auto useg = [](const Seg3& s, int64_t idx, int64_t nper, int64_t& local) -> const float* {
    const bool g1 = idx >= nper, g2 = idx >= 2 * nper;
    local = idx - (g2 ? 2 * nper : (g1 ? nper : 0));
    return s.p0 + (g2 ? s.d2 : (g1 ? s.d1 : 0));
  };
  int64_t aloc_m, bloc_n;
  const float* a_mbase = nullptr;
  const float* b_nbase = nullptr;
  if (!A_KC) a_mbase = useg(g.a, m0, g.a_nper, aloc_m) + aloc_m;                     // + k*lda
  if (B_KC) b_nbase = useg(g.b, n0, g.b_nper, bloc_n) + bloc_n * g.ldb;             // + k
  auto abase = [&](int64_t k0) -> const float* {
    if (A_KC) {
      int64_t loc;
      const float* p = useg(g.a, k0, g.a_nper, loc);
      return p + m0 * g.lda + loc;
    }
    return a_mbase + k0 * g.lda;
  };
  auto bbase = [&](int64_t k0) -> const float* {
    if (B_KC) return b_nbase + k0;
    int64_t loc;
    const float* p = useg(g.b, k0, g.b_nper, loc);
    return p + loc * g.ldb + n0;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
#define GCT_GLOAD(k0_)                                                                     \
  do {                                                                                     \
    const float* ab__ = abase(k0_);                                                        \
    const float* bb__ = bbase(k0_);                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
        ra[i] = *reinterpret_cast<const float4*>(ab__ + offa[i]);                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
        rb[i] = *reinterpret_cast<const float4*>(bb__ + offb[i]);                          \
  } while (0)
#define GCT_LSTORE(buf_)                                                                   \
  do {                                                                                     \
    float* la__ = lds + (buf_) * 2 * TILE_FLOATS;                                          \
    float* lb__ = la__ + TILE_FLOATS;                                                      \
    if (A_KC) store_kc(la__, ra, tid); else store_rc(la__, ra, tid);                       \
    if (B_KC) store_kc(lb__, rb, tid); else store_rc(lb__, rb, tid);                       \
  } while (0)
// the prefetched tile must stay in VGPRs until the MFMA block is done: an opaque use right
// before the LDS stores stops hipcc from sinking the stores (and their vmcnt waits, and
// scratch spills) to just behind the loads
#define GCT_PIN(v_) asm volatile("" : "+v"(v_.x), "+v"(v_.y), "+v"(v_.z), "+v"(v_.w))

  const int64_t nkt = (kend - kbeg) / BK;
#ifdef GCT_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
  // wgrad only: db[n] = sum_m dY[m][n] falls out of the A tiles already in registers.  Weights
  // (0/1) instead of branches keep the loop a single basic block; only the tile_n == 0 column of
  // workgroups contributes, and the duplicated final prefetch is masked out.
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const float bw0 = (!A_KC && !B_KC && g.bias_slab && n0 == 0) ? 1.f : 0.f;
#define GCT_BSUM(w_)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
    bsum.x = fmaf(w_, ra[i].x, bsum.x); bsum.y = fmaf(w_, ra[i].y, bsum.y);                 \
    bsum.z = fmaf(w_, ra[i].z, bsum.z); bsum.w = fmaf(w_, ra[i].w, bsum.w);                 \
  }
  if (nkt > 0) {
    GCT_GLOAD(kbeg);
    if (!A_KC && !B_KC) { GCT_BSUM(bw0) }
    GCT_LSTORE(0);
  }
  __syncthreads();
  STAMP(0);
  // The loop body is ONE basic block (no branches: the last iteration re-loads the final tile
  // and stores it to the idle LDS buffer, which nobody reads) so that the scheduler can
  // interleave memory instructions with the MFMA stream:
  //   quarter 0: the 8 global loads of tile t+1 trickle out, one per two MFMAs (issued as a
  //              burst right after the barrier the 4 waves block ~750 cycles on the CU's
  //              vector-memory issue path -- s_memtime stamps, tools/gemm_stamps.hip);
  //   quarter 3: the 8 ds_write_b128 of tile t+1 (other LDS buffer), one per two MFMAs.
  // Exposed per K-tile: the barrier and the first fragment read.
  float fa[2][2][4], fb[2][2][4];
#define GCT_FRAGQ(set_, c_, la_, lb_)                                                        \
  do {                                                                                       \
    if (A_KC) fragq_kc(fa[set_], la_, wm, lane, c_); else fragq_rc(fa[set_], la_, wm, lane, c_); \
    if (B_KC) fragq_kc(fb[set_], lb_, wn, lane, c_); else fragq_rc(fb[set_], lb_, wn, lane, c_); \
  } while (0)
#define GCT_MFMA16(set_)                                                                     \
  _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_][i][e], fb[set_][j][e], acc[i][j], 0, 0, 0)
  if (nkt > 0) GCT_FRAGQ(0, 0, lds, lds + TILE_FLOATS);
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = (int)(kt & 1);
    const float* la = lds + cur * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
    const int64_t knext = (kt + 1 < nkt) ? kbeg + (kt + 1) * BK : kbeg + kt * BK;
    __builtin_amdgcn_sched_barrier(0);
    // ---- quarter 0
    GCT_FRAGQ(1, 1, la, lb);
    GCT_GLOAD(knext);
    GCT_MFMA16(0);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);  // fragment reads first
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 global load
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    // ---- quarter 1, 2
    GCT_FRAGQ(0, 2, la, lb);
    __builtin_amdgcn_sched_barrier(0);
    GCT_MFMA16(1);
    __builtin_amdgcn_sched_barrier(0);
    GCT_FRAGQ(1, 3, la, lb);
    __builtin_amdgcn_sched_barrier(0);
    GCT_MFMA16(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(2);
    // ---- quarter 3 + LDS stores of the next tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      GCT_PIN(ra[i]);
      GCT_PIN(rb[i]);
    }
    if (!A_KC && !B_KC) {
      const float bw = (kt + 1 < nkt) ? bw0 : 0.f;
      GCT_BSUM(bw)
    }
    GCT_LSTORE(cur ^ 1);
    GCT_MFMA16(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 LDS store
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    __syncthreads();
    STAMP(4);
    GCT_FRAGQ(0, 0, lds + (cur ^ 1) * 2 * TILE_FLOATS, lds + (cur ^ 1) * 2 * TILE_FLOATS + TILE_FLOATS);
  }
  __syncthreads();  // the speculative fragment read above must finish before LDS is reused
#undef GCT_FRAGQ
#undef GCT_MFMA16
#undef GCT_BSUM
  if (!A_KC && !B_KC) {
    if (g.bias_slab && n0 == 0) {            // block-uniform: combine the 8 row groups, fixed order
      float4* red = reinterpret_cast<float4*>(lds);
      red[tid] = bsum;
      __syncthreads();
      if (tid < 32) {
        float4 t4 = red[tid];
        for (int r = 1; r < 8; ++r) {
          const float4 u = red[r * 32 + tid];
          t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w;
        }
        const int64_t col = m0 + tid * 4;
        if (col + 3 < g.M) *reinterpret_cast<float4*>(g.bias_slab + (int64_t)z * g.M + col) = t4;
      }
      __syncthreads();
    }
  }
#undef GCT_GLOAD
#undef GCT_LSTORE
#undef GCT_PIN

  // ---- epilogue: per-wave 64x64 transpose through LDS (the tile buffers are free now)
  wave_epilogue(g, acc, lds + wave * 4096, lane, m0 + wm, n0 + wn, z);
#ifdef GCT_STAMPS
  STAMP(6);  // epilogue
  if (g.stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 64) {
    for (int i = 0; i < 8; ++i) g.stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i] = seg[i];
  }
#endif
}


int64_t g_x6_kernel_launches = 0;   // gemm_x6_kernel launches (a tail-balanced call makes two)
#include "gemm_x6.inc"
#ifdef GCT_LAB_X6P
// tools/gemm_lab.hip only: the persistent stream-K form of the forward / dgrad kernels (round-4 experiment, measured and
// not adopted: profiles/r04_gemm_experiment_persistent_*.log).  Nothing of it is compiled into the library.
#include "../../tools/gemm_x6p.inc"
#include "../../tools/gemm_x6w.inc"      // one wave per SIMD (round-4 experiment, forward only): g_x6p_on == 2
#include "../../tools/gemm_x6h.inc"      // two 4-wave workgroups per CU, 128 x 128 tiles (round-4 experiment, forward only): g_x6p_on == 4
constexpr int X6P_SYNC_SLOTS = 32;
int g_x6p_on = 0;
int* g_x6p_counters = nullptr;
unsigned g_x6p_seq = 0;
inline int* x6p_counter_slot() {
  if (!g_x6p_counters) return nullptr;
  return g_x6p_counters + (size_t)((g_x6p_seq++) % X6P_SYNC_SLOTS) * 256;
}
#endif

// =====================================================================================
// SKINNY-M forward kernel (KV-cached decode: M = batch rows per step = 512): 64x64 tiles, 4 waves
// of one 32x32 MFMA tile each, 32 KB LDS => 4-5 workgroups per CU and 4x as many tiles as the
// 128x128 kernel, which fills only 16-64 of the 256 CUs at M = 512.  Same operand layout
// (x [M][K], w [N][K]), same epilogue (FastEpi), simple two-buffer pipeline.
// =====================================================================================
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const GemmArgs g) {
  constexpr int SB = 64, STILE = SB * BK;  // 2048 floats per operand tile
  __shared__ __attribute__((aligned(16))) float lds[4 * STILE];  // A0 B0 A1 B1 : 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const unsigned tiles_n = (unsigned)((g.N + SB - 1) / SB);
  const unsigned tiles_m = (unsigned)((g.M + SB - 1) / SB);
  const unsigned lid0 = gct_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lid0 / (tiles_m * tiles_n), lid = lid0 - z * (tiles_m * tiles_n);
  const int64_t m0 = (int64_t)(lid / tiles_n) * SB, n0 = (int64_t)(lid % tiles_n) * SB;
  const int64_t kbeg = (int64_t)z * g.ksplit;
  const int64_t kend = (kbeg + g.ksplit < g.K) ? kbeg + g.ksplit : g.K;

  uint32_t offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t r = (tid >> 3) + 32 * i;
    int64_t ra_ = r, rb_ = r;
    if (m0 + ra_ > g.M - 1) ra_ = g.M - 1 - m0;
    if (n0 + rb_ > g.N - 1) rb_ = g.N - 1 - n0;
    offa[i] = (uint32_t)(ra_ * g.lda + (tid & 7) * 4);
    offb[i] = (uint32_t)(rb_ * g.ldb + (tid & 7) * 4);
  }
  const bool g1 = n0 >= g.b_nper, g2 = n0 >= 2 * g.b_nper;   // the 64-row weight tile lies in one segment
  const float* bbase0 = g.b.p0 + (g2 ? g.b.d2 : (g1 ? g.b.d1 : 0)) +
                        (n0 - (g2 ? 2 * g.b_nper : (g1 ? g.b_nper : 0))) * g.ldb;
  const float* abase0 = g.a.p0 + m0 * g.lda;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[2], rb[2];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(abase0 + k0 + offa[i]);
      rb[i] = *reinterpret_cast<const float4*>(bbase0 + k0 + offb[i]);
    }
  };
  auto lstore = [&](int buf) {
    float* la = lds + buf * 2 * STILE;
    float* lb = la + STILE;
    const int c8 = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int chunk = c8 ^ ((row >> 1) & 7);
      *reinterpret_cast<float4*>(la + row * BK + chunk * 4) = ra[i];
      *reinterpret_cast<float4*>(lb + row * BK + chunk * 4) = rb[i];
    }
  };
  const int64_t nkt = (kend - kbeg) / BK;
  gload(kbeg);
  lstore(0);
  __syncthreads();
  const int h = lane >> 5;
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nkt) gload(kbeg + (kt + 1) * BK);
    const float* la = lds + cur * 2 * STILE;
    const float* lb = la + STILE;
    const int rowa = wm + (lane & 31), rowb = wn + (lane & 31);
    const int swa = (rowa >> 1) & 7, swb = (rowb >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a4 = *reinterpret_cast<const float4*>(la + rowa * BK + (((h * 4 + c) ^ swa) * 4));
      const float4 b4 = *reinterpret_cast<const float4*>(lb + rowb * BK + (((h * 4 + c) ^ swb) * 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
  }
  // epilogue: per-wave 32x32 transpose through LDS, 4x4 patch per lane
  float* stg = lds + wave * 1024;
  {
    const int c32 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c32] = acc[r];
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  const FastEpi ep{g};
  const int rg = lane >> 3, c4 = lane & 7;           // 8 row groups x 8 column chunks
  const int64_t row0 = m0 + wm + rg * 4, col0 = n0 + wn + c4 * 4;
  if (row0 < g.M && col0 < g.N) {
    float4 v[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) v[rr] = *reinterpret_cast<const float4*>(stg + (rg * 4 + rr) * 32 + c4 * 4);
    if (g.epi == EPI_SLAB) {                 // split-K partial: raw sums, dense [M][N] slab z
      float* sl = g.c0 + (int64_t)z * g.slab_stride;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (row0 + rr < g.M) *reinterpret_cast<float4*>(sl + (row0 + rr) * g.N + col0) = v[rr];
      return;
    }
    const bool q1 = col0 >= g.c_nper, q2 = col0 >= 2 * g.c_nper;
    const int64_t cloc = col0 - (q2 ? 2 * g.c_nper : (q1 ? g.c_nper : 0));
    float* cbase = g.c0 + (q2 ? g.c_d2 : (q1 ? g.c_d1 : 0));
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias0) bias = *reinterpret_cast<const float4*>(g.bias0 + (q2 ? g.bias_d2 : (q1 ? g.bias_d1 : 0)) + cloc);
    ep.apply(v, row0, col0, cbase, cloc, bias);
  }
}


// =====================================================================================
// PANEL kernel for the smallest skinny problems (a 512-row decode step: M x N / (32 x TN) <= ~512 tiles): one
// workgroup per 32 x TN output tile runs the WHOLE reduction -- no split-K slabs, no fix-up launch -- on K chunks of
// 256: the chunk's operand panels (A 32 x 256, B TN x 256 fp32) are requested in one burst (8 + TN/4 float4 per
// thread in flight), written to LDS once and multiplied from there (v_mfma_f32_16x16x4_f32, four waves = 2 x 2 or
// 2 x 4 sub-tiles); the next chunk's burst is in flight during the MFMAs.  A step of the KV-cached decode is ~37
// dependent GEMMs, each far too small to fill the chip: what counts is the latency of one launch (was: 64 x 64
// tiles with K split 4-12 ways, 17 us, + a 9 us fix-up launch).
// =====================================================================================
// RAGGED (TN = 32 only): N <= 32 output columns of any count and leading dimension (the vocabulary head: 28-31
// columns) -- weight rows beyond N are clamped to the last one (they feed columns that are never stored), the
// epilogue is the plain bias one with scalar guarded stores.  One column of workgroups, M / 32 of them.
template <int TN, bool RAGGED = false>
__global__ __launch_bounds__(256) void gemm_f32_panel_kernel(const GemmArgs g) {
  constexpr int TM = 32, KC = 256, SD = KC + 4, NBL = TN / 4;   // NBL float4 of the B panel per thread and chunk
  extern __shared__ __attribute__((aligned(16))) float plds[];
  float* As = plds;                    // [TM][SD]
  float* Bs = plds + TM * SD;          // [TN][SD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, c16 = lane & 15;
  const unsigned tiles_m = (unsigned)((g.M + TM - 1) / TM);
  const unsigned lid = gct_xcd_remap(blockIdx.x, gridDim.x);
  // consecutive logical ids (= one XCD) share a weight panel
  const int64_t m0 = (int64_t)(lid % tiles_m) * TM, n0 = (int64_t)(lid / tiles_m) * TN;
  const bool s1 = n0 >= g.b_nper, s2 = n0 >= 2 * g.b_nper;     // the TN weight rows lie in one segment
  const float* bbase = g.b.p0 + (s2 ? g.b.d2 : (s1 ? g.b.d1 : 0)) + (n0 - (s2 ? 2 * g.b_nper : (s1 ? g.b_nper : 0))) * g.ldb;
  const float* abase = g.a.p0;
  uint32_t offa[8], offb[NBL];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c4 = idx & 63;
    int64_t r = m0 + row;
    if (r > g.M - 1) r = g.M - 1;
    offa[i] = (uint32_t)(r * g.lda + c4 * 4);
  }
#pragma unroll
  for (int i = 0; i < NBL; ++i) {
    const int idx = tid + 256 * i, c4 = idx & 63;
    int64_t row = idx >> 6;
    if (RAGGED && n0 + row > g.N - 1) row = g.N - 1 - n0;
    offb[i] = (uint32_t)(row * g.ldb + c4 * 4);
  }
  f32x4_t ra[8], rb[NBL];           // native vectors: HIP's float4 struct copies kept these arrays in scratch
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const f32x4_t*>(abase + k0 + offa[i]);
#pragma unroll
    for (int i = 0; i < NBL; ++i) rb[i] = *reinterpret_cast<const f32x4_t*>(bbase + k0 + offb[i]);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<f32x4_t*>(As + (idx >> 6) * SD + (idx & 63) * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<f32x4_t*>(Bs + (idx >> 6) * SD + (idx & 63) * 4) = rb[i];
    }
  };
  constexpr int NS = TN / 32;          // 16 x 16 sub-tiles per wave (they share the A fragment)
  const int wm = (wave >> 1) * 16, wn = (wave & 1) * (TN / 2);
  f32x4_t acc[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int64_t nch = g.K / KC;
  gload(0);
  lstore();
  __syncthreads();
  for (int64_t c = 0; c < nch; ++c) {
    if (c + 1 < nch) gload((c + 1) * KC);
    // lane group g4 owns k = 64 g4 .. 64 g4 + 63 of the chunk for BOTH operands (any k permutation is legal)
    const float* ap = As + (wm + c16) * SD + g4 * 64;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const float4 a4 = *reinterpret_cast<const float4*>(ap + 4 * j);
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + (wn + 16 * u + c16) * SD + g4 * 64 + 4 * j);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc[u], 0, 0, 0);
      }
    }
    __syncthreads();                   // every wave is done with this chunk's panels
    if (c + 1 < nch) {
      lstore();
      __syncthreads();
    }
  }
  // epilogue: per-wave 16 x 16 transpose through LDS (the panels are free), one 4 x 4 patch per lane 0..15
  float* stg = plds + wave * (16 * 20);
  const FastEpi ep{g};
#pragma unroll
  for (int u = 0; u < NS; ++u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stg[(4 * g4 + i) * 20 + c16] = acc[u][i];   // C[m = 4 g4 + i][n = c16]
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) {
      const int pr = lane >> 2, pc = lane & 3;
      const int64_t row0 = m0 + wm + 4 * pr, col0 = n0 + wn + 16 * u + 4 * pc;
      if (RAGGED) {
        for (int rr = 0; rr < 4; ++rr)
          for (int cc = 0; cc < 4; ++cc)
            if (row0 + rr < g.M && col0 + cc < g.N)
              g.c0[(row0 + rr) * g.ldc + col0 + cc] = stg[(4 * pr + rr) * 20 + 4 * pc + cc] + (g.bias0 ? g.bias0[col0 + cc] : 0.f);
      } else if (row0 < g.M && col0 < g.N) {
        float4 v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v[rr] = *reinterpret_cast<const float4*>(stg + (4 * pr + rr) * 20 + 4 * pc);
        const bool q1 = col0 >= g.c_nper, q2 = col0 >= 2 * g.c_nper;
        const int64_t cloc = col0 - (q2 ? 2 * g.c_nper : (q1 ? g.c_nper : 0));
        float* cbase = g.c0 + (q2 ? g.c_d2 : (q1 ? g.c_d1 : 0));
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias0) bias = *reinterpret_cast<const float4*>(g.bias0 + (q2 ? g.bias_d2 : (q1 ? g.bias_d1 : 0)) + cloc);
        ep.apply(v, row0, col0, cbase, cloc, bias);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int TN, bool RAGGED = false>
int launch_panel(const GemmArgs& g, hipStream_t st) {
  constexpr size_t LDS = (size_t)(32 + TN) * 260 * sizeof(float);
  static bool attr_set = false;        // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f32_panel_kernel<TN, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) {
      gct_set_error("gemm_f32_panel: cannot reserve %zu bytes of LDS: %s", LDS, hipGetErrorString(e));
      return GCT_ERR_HIP;
    }
    attr_set = true;
  }
  const int64_t tiles = ((g.M + 31) / 32) * (RAGGED ? 1 : g.N / TN);
  hipLaunchKernelGGL((gemm_f32_panel_kernel<TN, RAGGED>), dim3((unsigned)tiles), dim3(256), LDS, st, g);
  GCT_LAUNCH_CHECK("gemm_f32_panel");
  return GCT_OK;
}

// split-K tail of the skinny-M path: out = epilogue(sum_s slab[s] + bias ...), one 4x4 patch per
// thread, float4 everywhere, same FastEpi as the GEMM kernels.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs g, const float* slabs,
                                                              int nslab, int64_t slab_stride) {
  const int64_t pc = g.N / 4, pr = (g.M + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pc * pr) return;
  const int64_t row0 = (idx / pc) * 4, col0 = (idx % pc) * 4;
  float4 v[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    v[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + rr < g.M) {
      const float* p = slabs + (row0 + rr) * g.N + col0;
      for (int s = 0; s < nslab; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)s * slab_stride);
        v[rr].x += t.x; v[rr].y += t.y; v[rr].z += t.z; v[rr].w += t.w;
      }
    }
  }
  const bool q1 = col0 >= g.c_nper, q2 = col0 >= 2 * g.c_nper;
  const int64_t cloc = col0 - (q2 ? 2 * g.c_nper : (q1 ? g.c_nper : 0));
  float* cbase = g.c0 + (q2 ? g.c_d2 : (q1 ? g.c_d1 : 0));
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias0) bias = *reinterpret_cast<const float4*>(g.bias0 + (q2 ? g.bias_d2 : (q1 ? g.bias_d1 : 0)) + cloc);
  const FastEpi ep{g};
  ep.apply(v, row0, col0, cbase, cloc, bias);
}

// GEMM arithmetic: GCT_GEMM_F32 = v_mfma_f32_32x32x2_f32 everywhere; GCT_GEMM_BF16X6 = exact 3-way bf16
// split with six partial products (gemm_x6.inc) wherever a launch qualifies.  Process-wide; the
// default comes from GCT_GEMM_MODE=f32|x6 (x6 when unset).
int g_gemm_mode = -1;
int64_t g_gemm_launches[2] = {0, 0};   // GEMM calls served by [fp32-MFMA kernels, bf16x6 kernels] (tests / diagnostics)
inline int gemm_mode() {
  if (g_gemm_mode < 0) {
    const char* e = getenv("GCT_GEMM_MODE");
    g_gemm_mode = (e && (e[0] == 'f' || e[0] == '0')) ? GCT_GEMM_F32 : GCT_GEMM_BF16X6;
  }
  return g_gemm_mode;
}

// single source of truth for "this launch takes gemm_f32_fast_kernel"
template <bool A_KC, bool B_KC>
bool fast_ok(const GemmArgs& g, bool vec) {
  static const bool no_fast = getenv("GCT_GEMM_NO_FAST") != nullptr;   // A/B switch for benchmarks
  auto seg_ok = [](int64_t nper, int64_t extent, int64_t gran) { return nper >= extent || nper % gran == 0; };
  return vec && !no_fast && g.K % BK == 0 && g.ksplit % BK == 0 && g.M >= 4 && g.N >= 4 &&
         (A_KC ? seg_ok(g.a_nper, g.K, BK) : (seg_ok(g.a_nper, g.M, BM) && g.M % 4 == 0)) &&
         (B_KC ? seg_ok(g.b_nper, g.N, BN) : (seg_ok(g.b_nper, g.K, BK) && g.N % 4 == 0)) &&
         g.N % 4 == 0 && g.ldc % 4 == 0 && gct_aligned16(g.c0) && g.c_d1 % 4 == 0 && g.c_d2 % 4 == 0 &&
         (g.c_nper >= g.N || g.c_nper % 4 == 0) && g.slab_stride % 4 == 0 &&
         (!g.bias0 || (gct_aligned16(g.bias0) && g.bias_d1 % 4 == 0 && g.bias_d2 % 4 == 0)) &&
         (!g.resid || gct_aligned16(g.resid)) && (!g.pre || gct_aligned16(g.pre)) &&
         (!g.pre_in || gct_aligned16(g.pre_in)) && 130 * g.lda < (1ll << 31) && 130 * g.ldb < (1ll << 31);
}

// Tail balancing for the bf16x6 forward / dgrad launches (one 128x256 workgroup per CU, 256 CUs): when the
// tile count leaves a partial last round, the rows of that round are computed as a second launch with K
// split s ways (s * rem workgroups of 1/s duration) into fp32 slabs, and a small fix-up kernel sums the slabs
// and applies the epilogue.  E.g. 640 tiles (N = 512 at 40 960 rows): 3 rounds -> 2.5; 2 592 tiles (N = 2 048
// at 41 472 rows): 11 rounds -> 10.25.  Deterministic (fixed summation order, no atomics).
// decides whether (and how) a [M][N] x K bf16x6 forward / dgrad launch is tail-split: returns the number of
// K-splits (1 = no) and the first tail row
int x6_tail_plan(int64_t M, int64_t N, int64_t K, int64_t* m1_out) {
  constexpr int64_t CUS = 256;
  static const bool off = getenv("GCT_X6_NO_TAIL_SPLIT") != nullptr;
  const int64_t tm = (M + XBM - 1) / XBM, tn = (N + XBN - 1) / XBN, tiles = tm * tn;
  const int64_t rem = tiles % CUS;
  if (off || tiles <= CUS || rem == 0 || rem % tn != 0 || K % XBK != 0) return 1;
  const int64_t nkt = K / XBK;
  int best = 1;
  double cost = 1.0;                                   // duration of the last round, in rounds
  // a workgroup costs (o + K-tiles) with o ~ 3.3 K-tile times of prologue + epilogue (s_memtime stamps),
  // so a 1/s share of K costs (o + nkt / s) / (o + nkt) of a full tile, not 1/s
  const double o = 3.3;
  for (int s2 = 2; s2 <= 8 && nkt / s2 >= 2; ++s2) {
    const double c = (double)((rem * s2 + CUS - 1) / CUS) * (o + (double)nkt / s2) / (o + nkt) + 0.05;  // + fix-up pass
    if (c < cost - 1e-9) { cost = c; best = s2; }
  }
  if (best == 1 || cost > 0.75) return 1;
  *m1_out = (tiles - rem) / tn * XBM;
  return best;
}

template <int MODE>
int launch_x6_tail_split(const GemmArgs& g, hipStream_t st, float* ws, int64_t ws_bytes) {
  if (MODE == X6_WGRAD || g.nsplit != 1 || !ws || !gct_aligned16(ws)) return launch_x6<MODE>(g, st);
  int64_t m1 = 0;
  const int best = x6_tail_plan(g.M, g.N, g.K, &m1);
  const int64_t m2 = g.M - m1, nkt = g.K / XBK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // keep captured graphs to one kernel per GEMM
  if (best == 1 || (int64_t)best * m2 * g.N * (int64_t)sizeof(float) > ws_bytes ||
      hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return launch_x6<MODE>(g, st);
  GemmArgs head = g;
  head.M = m1;
  int rc = launch_x6<MODE>(head, st);
  if (rc) return rc;
  GemmArgs e = g;                                      // the tail rows as their own problem
  e.M = m2; e.row_base = g.row_base + m1;
  e.a.p0 += m1 * g.lda;                                // FWD / DGRAD: A is [M][K]
  e.c0 += m1 * g.ldc;
  if (e.resid) e.resid += m1 * g.ldc;
  if (e.pre) e.pre += m1 * g.ldc;
  // pre_in read through the quad map keeps the forward's row space: the tail finds its rows through row_base
  if (e.pre_in && !(g.pre_rows > 0 && g.quad_map)) e.pre_in += m1 * g.ldc;
  {
    // K <= 1 024 and a short tail: the tail rows on 64 x 128 tiles (gemm_x6s_kernel, up to two per CU) finish with
    // their own epilogue -- no slabs, no fix-up launch (48.6 -> 48.2 ms per step with the forward launches alone)
    static const int small_cap = getenv("GCT_X6_TAIL_SMALL") ? atoi(getenv("GCT_X6_TAIL_SMALL")) : 512;  // A/B switch: 0 = off
    static const int small_dg = getenv("GCT_X6_TAIL_SMALL_DGRAD") ? atoi(getenv("GCT_X6_TAIL_SMALL_DGRAD")) : 1;
    const int64_t small = ((m2 + SBM - 1) / SBM) * ((g.N + SBN - 1) / SBN);
    if (nkt <= 32 && small <= small_cap) {
      if (MODE == X6_FWD && x6s_ok(e, true)) return launch_x6s<X6_FWD>(e, st);
      if (MODE == X6_DGRAD && small_dg && x6s_dgrad_ok(e, true)) return launch_x6s<X6_DGRAD>(e, st);
    }
  }
  GemmArgs p = e;
  p.nsplit = best;
  p.ksplit = ((nkt + best - 1) / best) * XBK;
  p.nsplit = (int)((g.K + p.ksplit - 1) / p.ksplit);
  p.epi = EPI_SLAB; p.c0 = ws; p.ldc = g.N; p.slab_stride = m2 * g.N;
  p.c_d1 = p.c_d2 = 0; p.c_nper = INT64_MAX / 4; p.bias0 = nullptr; p.resid = nullptr; p.pre = nullptr; p.pre_in = nullptr;
  rc = launch_x6<MODE>(p, st);
  if (rc) return rc;
  const int64_t patches = ((m2 + 3) / 4) * (g.N / 4);
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((patches + 255) / 256)), dim3(256), 0, st, e,
                     (const float*)ws, p.nsplit, p.slab_stride);
  GCT_LAUNCH_CHECK("x6 tail fix-up");
  return GCT_OK;
}

// Few 128x256 tiles but a long reduction (FFN-2 of a 1 024 .. 8 192-row decode step: 32-64 tiles, K = 2 048): the bf16x6
// kernel with K split s ways into fp32 slabs + the fix-up kernel fills the chip where the plain launch would use a
// quarter of it and the skinny fp32 kernel would run at the fp32 pipe's rate.  Returns 1 if it took the launch.
// number of K-splits of that route (1 = not taken); shared with gct_linear_fwd_ws_bytes
int x6_splitk_all_plan(int64_t M, int64_t N, int64_t K) {
  constexpr int64_t CUS = 256;
  const int64_t tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN), nkt = K / XBK;
  // K = 1024 (the folded cross-attention output projection of a decode step) pays only from 2 048 rows on: with fewer
  // the panel kernel is faster (+7 % per step at n = 1 024, measured both ways at 1 024 ... 8 192)
  if (tiles > CUS / 2 || K % XBK != 0 || nkt < 32 || M < 1024 || (nkt < 64 && M < 2048)) return 1;
  int64_t sp = CUS / tiles;
  if (sp > nkt / 8) sp = nkt / 8;                      // >= 8 K-tiles per split
  if (sp > 8) sp = 8;
  return sp < 2 ? 1 : (int)sp;
}
template <int MODE>
int launch_x6_splitk_all(const GemmArgs& g, hipStream_t st, float* ws, int64_t ws_bytes, int* taken) {
  *taken = 0;
  if ((MODE != X6_FWD && MODE != X6_DGRAD) || g.nsplit != 1 || !ws || !gct_aligned16(ws)) return GCT_OK;
  const int64_t sp = x6_splitk_all_plan(g.M, g.N, g.K), nkt = g.K / XBK;
  if (sp < 2 || sp * g.M * g.N * (int64_t)sizeof(float) > ws_bytes) return GCT_OK;
  GemmArgs p = g;
  p.ksplit = ((nkt + sp - 1) / sp) * XBK;
  p.nsplit = (int)((g.K + p.ksplit - 1) / p.ksplit);
  p.epi = EPI_SLAB; p.c0 = ws; p.ldc = g.N; p.slab_stride = g.M * g.N;
  p.c_d1 = p.c_d2 = 0; p.c_nper = INT64_MAX / 4; p.bias0 = nullptr; p.resid = nullptr; p.pre = nullptr; p.pre_in = nullptr;
  int rc = launch_x6<MODE>(p, st);
  if (rc) return rc;
  const int64_t patches = ((g.M + 3) / 4) * (g.N / 4);
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((patches + 255) / 256)), dim3(256), 0, st, g,
                     (const float*)ws, p.nsplit, p.slab_stride);
  GCT_LAUNCH_CHECK("x6 split-K fix-up");
  *taken = 1;
  return GCT_OK;
}

template <bool A_KC, bool B_KC>
int launch(const GemmArgs& g, bool vec, hipStream_t st, float* skinny_ws = nullptr, int64_t ws_bytes = 0) {
  const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.nsplit;
  if (tiles <= 0) return GCT_OK;
  if (tiles > INT_MAX) {
    gct_set_error("gemm: grid too large");
    return GCT_ERR_ARG;
  }
  dim3 grid((unsigned)tiles), block(256);
  const bool fast = fast_ok<A_KC, B_KC>(g, vec);
  static const int stagger_env = getenv("GCT_GEMM_STAGGER") ? atoi(getenv("GCT_GEMM_STAGGER")) : -1;
  if (A_KC && B_KC && gemm_mode() == GCT_GEMM_BF16X6 && g.epi < EPI_D0) {
    // few 128 x 256 tiles but enough 64 x 128 ones: the small-tile bf16x6 kernel (decode steps of 2 000-4 000 rows,
    // training at small batches)
    static const int x6s_on = getenv("GCT_X6S") ? atoi(getenv("GCT_X6S")) : 1;          // A/B switch
    const int64_t big = ((g.M + XBM - 1) / XBM) * ((g.N + XBN - 1) / XBN);
    const int64_t small = ((g.M + SBM - 1) / SBM) * ((g.N + SBN - 1) / SBN);
    // measured (tools/kernel_bench.py --suite decgemm, rows 1 024 ... 5 184): a 64 x 128 tile takes 0.75 us per K-tile
    // alone on its CU and 1.45 us when two share it, a 128 x 256 tile 2.8 us; from K = 2 048 on the large kernel's
    // K-split routes fill the chip and win, at K = 1 024 only while every small tile has a CU to itself
    const int64_t nkt_ = g.K / XBK;
    if (x6s_on && big <= 160 && small >= 96 && small <= (nkt_ <= 16 ? 512 : 256) && nkt_ <= 32 && x6s_ok(g, vec)) {
      ++g_gemm_launches[1];
      return launch_x6s(g, st);
    }
    // weight segments of 128 rows (the sampler's mu | log_var, N = 2 x 128): not a whole number of 256-row tiles per
    // segment, so the large kernel cannot take them -- the small-tile kernel for the whole problem instead of fp32 MFMA
    if (x6s_on && !x6_ok<X6_FWD>(g, vec) && small >= 96 && x6s_ok(g, vec)) {
      ++g_gemm_launches[1];
      return launch_x6s(g, st);
    }
    if (x6_ok<X6_FWD>(g, vec)) {
      int taken = 0;
      const int rc = launch_x6_splitk_all<X6_FWD>(g, st, skinny_ws, ws_bytes, &taken);
      if (rc || taken) {
        if (taken) ++g_gemm_launches[1];
        return rc;
      }
    }
  }
  if (A_KC && !B_KC && gemm_mode() == GCT_GEMM_BF16X6 && g.epi >= EPI_D0 && g.epi < EPI_SLAB) {
    // the same for dgrad launches (training at small batches): few 128 x 256 tiles, enough 64 x 128 ones
    static const int x6s_dg = getenv("GCT_X6S_DGRAD") ? atoi(getenv("GCT_X6S_DGRAD")) : 1;              // A/B switch
    const int64_t big = ((g.M + XBM - 1) / XBM) * ((g.N + XBN - 1) / XBN);
    const int64_t small = ((g.M + SBM - 1) / SBM) * ((g.N + SBN - 1) / SBN);
    const int64_t nkt_ = g.K / XBK;
    if (x6s_dg && big <= 160 && small >= 96 && small <= (nkt_ <= 16 ? 512 : 256) && nkt_ <= 32 && x6s_dgrad_ok(g, vec)) {
      ++g_gemm_launches[1];
      return launch_x6s<X6_DGRAD>(g, st);
    }
    // ... and a long reduction over few tiles (K' = 1 536 / 2 048 at batch 64): K split over the whole problem + fix-up
    static const int dg_splitk = getenv("GCT_X6_DGRAD_SPLITK") ? atoi(getenv("GCT_X6_DGRAD_SPLITK")) : 1;  // A/B switch
    if (dg_splitk && x6_ok<X6_DGRAD>(g, vec)) {
      int taken = 0;
      const int rc = launch_x6_splitk_all<X6_DGRAD>(g, st, skinny_ws, ws_bytes, &taken);
      if (rc || taken) {
        if (taken) ++g_gemm_launches[1];
        return rc;
      }
    }
  }
  // narrow outputs (the vocabulary head, N = 28-31): the ragged panel kernel, exact fp32 MFMA, M / 32 workgroups
  if (A_KC && B_KC && vec && g.N <= 32 && g.K % 256 == 0 && g.epi == GCT_EPI_BIAS && g.nsplit == 1 && g.b_nper >= g.N &&
      g.c_nper >= g.N && g.M >= 1 && g.lda * 4 * 33 < (1ll << 31) && g.ldb * 4 * 33 < (1ll << 31) && g.M < (1ll << 36)) {
    static const bool no_thin = getenv("GCT_GEMM_NO_THIN") != nullptr;     // A/B switch
    if (!no_thin) {
      ++g_gemm_launches[0];
      return launch_panel<32, true>(g, st);
    }
  }
  // skinny M (decode steps): 64x64 tiles when the 128x128 grid would leave most CUs idle
  // (192: with more tiles the bf16x6 kernel wins even on half the CUs -- decode at n >= 6144 lost 5-8 % with 384)
  if (fast && A_KC && B_KC && g.nsplit == 1 && tiles < 192 && g.epi < EPI_D0 &&
      (g.b_nper >= g.N || g.b_nper % 64 == 0)) {
    // ... and the panel kernel (whole reduction in one workgroup, one launch) when even those are few
    static const bool no_panel = getenv("GCT_GEMM_NO_PANEL") != nullptr;   // A/B switch for benchmarks
    if (!no_panel && g.K % 256 == 0 && g.lda * 4 * 33 < (1ll << 31) && g.ldb * 4 * 65 < (1ll << 31) &&
        (g.c_nper >= g.N || g.c_nper % 64 == 0)) {
      const int64_t tm32 = (g.M + 31) / 32;
      static const int64_t cap32 = getenv("GCT_PANEL_MAX32") ? atoll(getenv("GCT_PANEL_MAX32")) : 4096;  // knobs; 4096: +9 % / +3 % per decode step at n = 1024 / 4096 over 384
      static const int64_t cap64 = getenv("GCT_PANEL_MAX64") ? atoll(getenv("GCT_PANEL_MAX64")) : 768;
      if (g.N % 32 == 0 && (g.b_nper >= g.N || g.b_nper % 32 == 0) && tm32 * (g.N / 32) <= cap32)
        return launch_panel<32>(g, st);
      if (g.N % 64 == 0 && tm32 * (g.N / 64) <= cap64) return launch_panel<64>(g, st);
    }
    const int64_t st_ = ((g.M + 63) / 64) * ((g.N + 63) / 64);
    const int64_t nkt = g.K / BK;
    int64_t ns = 1;
    if (skinny_ws && gct_aligned16(skinny_ws)) {      // split-K: >= 4 K-tiles per split, ~768 blocks
      ns = 768 / st_;
      if (ns > nkt / 4) ns = nkt / 4;
      static const int ns_cap = getenv("GCT_SKINNY_SPLIT_MAX") ? atoi(getenv("GCT_SKINNY_SPLIT_MAX")) : 1 << 30;
      if (ns > ns_cap) ns = ns_cap;
      // never more slabs than the caller's workspace holds (fewer splits, same result up to summation order)
      const int64_t slab_b = g.M * g.N * (int64_t)sizeof(float);
      if (slab_b > 0 && ns * slab_b > ws_bytes) ns = ws_bytes / slab_b;
      if (ns < 1) ns = 1;
    }
    if (ns <= 1) {
      hipLaunchKernelGGL(gemm_f32_small_kernel, dim3((unsigned)st_), dim3(256), 0, st, g);
      GCT_LAUNCH_CHECK("gemm_f32_small");
      return GCT_OK;
    }
    GemmArgs p = g;                                   // pass 1: raw partial sums into slabs
    p.ksplit = ((nkt + ns - 1) / ns) * BK;
    p.nsplit = (int)((g.K + p.ksplit - 1) / p.ksplit);
    p.epi = EPI_SLAB;
    p.c0 = skinny_ws;
    p.slab_stride = g.M * g.N;
    hipLaunchKernelGGL(gemm_f32_small_kernel, dim3((unsigned)(st_ * p.nsplit)), dim3(256), 0, st, p);
    GCT_LAUNCH_CHECK("gemm_f32_small(split-K)");
    const int64_t patches = ((g.M + 3) / 4) * (g.N / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((patches + 255) / 256)), dim3(256), 0, st,
                       g, (const float*)skinny_ws, p.nsplit, p.slab_stride);
    GCT_LAUNCH_CHECK("splitk_epilogue");
    return GCT_OK;
  }
  if (gemm_mode() == GCT_GEMM_BF16X6) {
    constexpr int MODE = A_KC ? (B_KC ? X6_FWD : X6_DGRAD) : X6_WGRAD;
    if ((A_KC || !B_KC) && x6_ok<MODE>(g, vec)) {
      ++g_gemm_launches[1];
#ifdef GCT_LAB_X6P
      if constexpr (MODE == X6_FWD) {
        if (g_x6p_on == 2 && g.nsplit == 1 && g.epi < EPI_D0 && g.a_nper >= g.K) return launch_x6w<X6_FWD>(g, st);
      }
      if constexpr (MODE == X6_FWD) {          // lab variant 3: the 64 x 128-tile kernel (two workgroups per CU) for the whole problem
        if (g_x6p_on == 3 && x6s_ok(g, vec)) return launch_x6s<X6_FWD>(g, st);
        if (g_x6p_on == 4 && x6s_ok(g, vec)) return launch_x6h<X6_FWD>(g, st);
      }
      if constexpr (MODE == X6_DGRAD) {
        if (g_x6p_on == 3 && x6s_dgrad_ok(g, vec)) return launch_x6s<X6_DGRAD>(g, st);
      }
      if constexpr (MODE != X6_WGRAD) {
        if (g_x6p_on == 1 && x6p_ok<MODE>(g, vec)) {
          int taken = 0;
          const int rc = launch_x6p<MODE>(g, st, skinny_ws, ws_bytes, x6p_counter_slot(), &taken);
          if (rc || taken) return rc;
        }
      }
#endif
      return launch_x6_tail_split<MODE>(g, st, skinny_ws, ws_bytes);
    }
  }
  ++g_gemm_launches[0];
  if (fast) {
    GemmArgs gs = g;
    // half of one tile's main loop: nkt K-tiles x ~4.5k cycles / 2, in units of s_sleep(127) ~ 8.1k cycles
    const int64_t nkt = (g.ksplit < g.K ? g.ksplit : g.K) / BK;
    // measured (tools/kernel_bench.py, GCT_GEMM_STAGGER=0/2/4/8): within +-2 % on every shape, so the
    // stagger is OFF unless requested -- kept as an experiment knob
    (void)nkt;
    gs.stagger = stagger_env > 0 ? stagger_env : 0;
    if (tiles <= 512) gs.stagger = 0;      // a single round: nothing to de-phase
    hipLaunchKernelGGL((gemm_f32_fast_kernel<A_KC, B_KC>), grid, block, 0, st, gs);
  }
  else if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true>), grid, block, 0, st, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false>), grid, block, 0, st, g);
  GCT_LAUNCH_CHECK("gemm_f32");
  return GCT_OK;
}

inline bool al16(const void* p) { return p == nullptr || gct_aligned16(p); }

inline Seg3 mkseg(const float* p0, const float* p1, const float* p2) {
  Seg3 r;
  r.p0 = p0;
  r.d1 = p1 ? (int64_t)(p1 - p0) : 0;
  r.d2 = p2 ? (int64_t)(p2 - p0) : 0;
  return r;
}

// deterministic slab reduction + bias-gradient helpers live in reduce.hip
}  // namespace

int gct_reduce_slabs_seg(const float* slabs, int nslab, int64_t stride, float* d0, float* d1,
                         float* d2, int64_t nper_elems, int64_t n, hipStream_t st);
int gct_reduce_slabs_seg2(const float* sa, int nslab, int64_t stride_a, float* a0, float* a1, float* a2,
                          int64_t nper_a, int64_t na, const float* sb, int64_t stride_b, float* b0, float* b1,
                          float* b2, int64_t nper_b, int64_t nb, hipStream_t st);
int gct_colsum(const float* y0, const float* y1, const float* y2, int64_t ld, int64_t M, int nseg,
               int nper, float* d0, float* d1, float* d2, float* ws, hipStream_t st);
int64_t gct_colsum_ws_floats(int64_t M, int64_t N);

static int wgrad_splits(int64_t M, int64_t Ntot, int64_t K, bool x6 = false) {
  const int64_t tiles = x6 ? ((Ntot + XBM - 1) / XBM) * ((K + XBN - 1) / XBN)
                           : ((Ntot + BM - 1) / BM) * ((K + BN - 1) / BN);
  // exactly one resident round: 256 CUs x 2 blocks (64 KB LDS each) = 512 blocks, never 513
  // (bf16x6: one 144 KB workgroup per CU = 256 blocks)
  int64_t want = (x6 ? 256 : 512) / tiles;
  const int64_t maxs = (M + 4 * BK - 1) / (4 * BK);
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int)want;
}

extern "C" int64_t gct_wgrad_ws_bytes(int64_t M, int64_t Ntot, int64_t K) {
  const int s1 = wgrad_splits(M, Ntot, K), s2 = wgrad_splits(M, Ntot, K, true);
  const int s = s1 > s2 ? s1 : s2;
  const int64_t slab = (int64_t)s * Ntot * K;
  const int64_t cs = gct_colsum_ws_floats(M, Ntot);
  const int64_t need = slab + 4 + (int64_t)s * Ntot;   // weight slabs + fused bias slabs
  return (need > cs ? need : cs) * (int64_t)sizeof(float) + 256;
}

static int linear_fwd_impl(const float* x, int64_t ldx, int64_t M, int K, const float* w0,
                           const float* w1, const float* w2, int64_t ldw, const float* b0,
                           const float* b1, const float* b2, int nseg, int nper, float* y0,
                           float* y1, float* y2, int64_t ldy, int epi, const float* resid,
                           float* pre, float p, uint64_t seed, uint32_t site, float* ws, int64_t ws_bytes,
                           void* stream, const uint16_t* wp0 = nullptr, int64_t pstride = 0,
                           const int32_t* quad_map = nullptr) {
  GCT_CHECK_ARG(x && w0 && y0 && M >= 0 && K > 0 && nseg >= 1 && nseg <= 3 && nper > 0,
                "linear_fwd: bad args");
  GCT_CHECK_ARG(!quad_map || M % 4 == 0, "linear_fwd: compacted rows come in quads");
  GCT_CHECK_ARG(ws_bytes >= 0, "linear_fwd: negative workspace size");
  GCT_CHECK_ARG(nseg < 2 || (w1 && y1), "linear_fwd: missing segment 1");
  GCT_CHECK_ARG(nseg < 3 || (w2 && y2), "linear_fwd: missing segment 2");
  GCT_CHECK_ARG(epi == GCT_EPI_BIAS || nseg == 1, "linear_fwd: fused epilogues need nseg == 1");
  GCT_CHECK_ARG(epi != GCT_EPI_GELU_DROP || pre, "linear_fwd: GELU epilogue needs pre");
  GCT_CHECK_ARG(epi != GCT_EPI_DROP_RESID || resid, "linear_fwd: residual epilogue needs resid");
  GCT_CHECK_ARG(p >= 0.f && p < 1.f, "linear_fwd: dropout p out of range");
  GemmArgs g = {};
  g.M = M; g.N = (int64_t)nseg * nper; g.K = K;
  GCT_CHECK_ARG(!b0 || nseg < 2 || b1, "linear_fwd: bias of segment 1 missing");
  GCT_CHECK_ARG(!b0 || nseg < 3 || b2, "linear_fwd: bias of segment 2 missing");
  g.a = mkseg(x, nullptr, nullptr); g.lda = ldx; g.a_nper = INT64_MAX / 4;
  g.b = mkseg(w0, w1, w2); g.ldb = ldw; g.b_nper = nper;
  g.c0 = y0; g.c_d1 = y1 ? y1 - y0 : 0; g.c_d2 = y2 ? y2 - y0 : 0; g.ldc = ldy; g.c_nper = nper;
  g.ksplit = K; g.nsplit = 1; g.epi = epi;
  g.bias0 = b0; g.bias_d1 = (b0 && b1) ? b1 - b0 : 0; g.bias_d2 = (b0 && b2) ? b2 - b0 : 0;
  g.resid = resid; g.pre = pre;
  g.thr = gct_drop_threshold(p); g.keep_scale = 1.0f / (1.0f - p); g.rng = gct_rng_make(seed, site);
  g.bp0 = wp0; g.bp_stride = pstride;
  g.quad_map = quad_map;          // rows are a quad compaction: the dropout coordinates of the epilogues follow the map
  const bool vec = al16(x) && al16(w0) && al16(w1) && al16(w2) && (ldx % 4 == 0) &&
                   (ldw % 4 == 0) && (K % 4 == 0);
  return launch<true, true>(g, vec, (hipStream_t)stream, ws, ws ? ws_bytes : 0);   // every slab route checks its need against ws_bytes
}

extern "C" int gct_linear_fwd(const float* x, int64_t ldx, int64_t M, int K, const float* w0,
                              const float* w1, const float* w2, int64_t ldw, const float* b0,
                              const float* b1, const float* b2, int nseg, int nper, float* y0,
                              float* y1, float* y2, int64_t ldy, int epi, const float* resid,
                              float* pre, float p, uint64_t seed, uint32_t site, void* stream) {
  return linear_fwd_impl(x, ldx, M, K, w0, w1, w2, ldw, b0, b1, b2, nseg, nper, y0, y1, y2, ldy, epi,
                         resid, pre, p, seed, site, nullptr, 0, stream);
}

extern "C" int gct_linear_fwd_ws(const float* x, int64_t ldx, int64_t M, int K, const float* w0,
                                 const float* w1, const float* w2, int64_t ldw, const float* b0,
                                 const float* b1, const float* b2, int nseg, int nper, float* y0,
                                 float* y1, float* y2, int64_t ldy, int epi, const float* resid,
                                 float* pre, float p, uint64_t seed, uint32_t site, float* ws,
                                 int64_t ws_bytes, void* stream) {
  return linear_fwd_impl(x, ldx, M, K, w0, w1, w2, ldw, b0, b1, b2, nseg, nper, y0, y1, y2, ldy, epi,
                         resid, pre, p, seed, site, ws, ws_bytes, stream);
}

extern "C" int gct_linear_fwd_p(const float* x, int64_t ldx, int64_t M, int K, const float* w0,
                                const float* w1, const float* w2, int64_t ldw, const uint16_t* wp0,
                                int64_t plane_stride, const float* b0, const float* b1,
                                const float* b2, int nseg, int nper, float* y0, float* y1, float* y2,
                                int64_t ldy, int epi, const float* resid, float* pre, float p,
                                uint64_t seed, uint32_t site, float* ws, int64_t ws_bytes,
                                const int32_t* quad_map, void* stream) {
  return linear_fwd_impl(x, ldx, M, K, w0, w1, w2, ldw, b0, b1, b2, nseg, nper, y0, y1, y2, ldy, epi,
                         resid, pre, p, seed, site, ws, ws_bytes, stream, wp0, plane_stride, quad_map);
}

extern "C" int gct_gemm_set_mode(int mode) {
  GCT_CHECK_ARG(mode == GCT_GEMM_F32 || mode == GCT_GEMM_BF16X6, "gemm_set_mode: unknown mode %d", mode);
  g_gemm_mode = mode;
  return GCT_OK;
}
extern "C" int gct_gemm_get_mode(void) { return gemm_mode(); }
#ifdef GCT_LAB_X6P
extern "C" int gct_gemm_set_persistent(int on) {
  g_x6p_on = on;
  return GCT_OK;
}
extern "C" int gct_gemm_set_sync_buffer(int32_t* buf, int64_t bytes) {
  GCT_CHECK_ARG(!buf || bytes >= (int64_t)X6P_SYNC_SLOTS * 256 * 4, "gemm_set_sync_buffer: need %d bytes, zero-initialised",
                X6P_SYNC_SLOTS * 256 * 4);
  g_x6p_counters = buf;
  return GCT_OK;
}
#endif
extern "C" int64_t gct_gemm_x6_kernel_launches(void) { return g_x6_kernel_launches; }
extern "C" int gct_gemm_launch_counts(int64_t* out2) {
  GCT_CHECK_ARG(out2, "gemm_launch_counts: null");
  out2[0] = g_gemm_launches[0]; out2[1] = g_gemm_launches[1];
  return GCT_OK;
}

extern "C" int gct_split_planes(const float* src, int64_t numel, uint16_t* planes, int64_t plane_stride,
                                void* stream) {
  GCT_CHECK_ARG(src && planes && numel >= 0 && numel % 4 == 0 && plane_stride >= numel && plane_stride % 4 == 0 &&
                    gct_aligned16(src) && ((((uintptr_t)planes) & 7u) == 0),
                "split_planes: bad args (numel and plane_stride must be multiples of 4, src 16-B aligned)");
  if (numel == 0) return GCT_OK;
  const int64_t n4 = numel / 4;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, planes, n4, plane_stride);
  GCT_LAUNCH_CHECK("split_planes");
  return GCT_OK;
}

extern "C" int64_t gct_linear_fwd_ws_bytes(int64_t M, int K, int Ntot) {
  const int64_t tiles = ((M + BM - 1) / BM) * (((int64_t)Ntot + BN - 1) / BN);
  int64_t need = 0;
  if (tiles < 192) {
    // skinny M: worst case K/128 slabs of [M][Ntot]
    const int64_t ns = K / (4 * BK) > 0 ? K / (4 * BK) : 1;
    need = ns * M * Ntot * (int64_t)sizeof(float);
  } else {
    int64_t m1 = 0;                                  // bf16x6 tail balancing: s slabs of the tail rows
    const int s = x6_tail_plan(M, Ntot, K, &m1);
    if (s > 1) need = (int64_t)s * (M - m1) * Ntot * (int64_t)sizeof(float);
  }
  const int64_t sp = x6_splitk_all_plan(M, Ntot, K);   // bf16x6 split-K over the whole problem (few tiles, long K)
  if (sp > 1 && sp * M * Ntot * (int64_t)sizeof(float) > need) need = sp * M * Ntot * (int64_t)sizeof(float);
  return need + 256;
}

extern "C" int64_t gct_linear_dgrad_ws_bytes(int64_t M, int Ntot, int K) {
  int64_t m1 = 0;
  const int s = x6_tail_plan(M, K, Ntot, &m1);       // dx is [M][K], reduced over Ntot
  const int64_t tail = s > 1 ? (int64_t)s * (M - m1) * K * (int64_t)sizeof(float) : 0;
  const int64_t sp = x6_splitk_all_plan(M, K, Ntot);  // few tiles, long reduction: K split over the whole problem
  const int64_t all = sp > 1 ? sp * M * K * (int64_t)sizeof(float) : 0;
  return (tail > all ? tail : all) + 256;
}

static int linear_dgrad_impl(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                             int64_t M, int nseg, int nper, const float* w0, const float* w1,
                             const float* w2, int64_t ldw, int K, float* dx, int64_t lddx,
                             int depi, const float* pre, float p, uint64_t seed, uint32_t site,
                             void* stream, const uint16_t* wp0, int64_t pstride, float* ws, int64_t ws_bytes,
                             const int32_t* quad_map = nullptr, int64_t pre_rows = 0) {
  GCT_CHECK_ARG(dy0 && w0 && dx && M >= 0 && K > 0 && nseg >= 1 && nseg <= 3 && nper > 0,
                "linear_dgrad: bad args");
  GCT_CHECK_ARG(!quad_map || M % 4 == 0, "linear_dgrad: compacted rows come in quads");
  GCT_CHECK_ARG(nseg < 2 || (w1 && dy1), "linear_dgrad: missing segment 1");
  GCT_CHECK_ARG(nseg < 3 || (w2 && dy2), "linear_dgrad: missing segment 2");
  GCT_CHECK_ARG(depi != GCT_DEPI_GELU_BWD || pre, "linear_dgrad: GELU bwd needs pre");
  GCT_CHECK_ARG(p >= 0.f && p < 1.f, "linear_dgrad: dropout p out of range");
  GemmArgs g = {};
  g.M = M; g.N = K; g.K = (int64_t)nseg * nper;  // reduce over the layer's output features
  g.a = mkseg(dy0, dy1, dy2); g.lda = lddy; g.a_nper = nper;
  g.b = mkseg(w0, w1, w2); g.ldb = ldw; g.b_nper = nper;
  g.c0 = dx; g.ldc = lddx; g.c_nper = INT64_MAX / 4;
  g.ksplit = g.K; g.nsplit = 1; g.epi = EPI_D0 + depi;
  g.pre_in = pre;
  g.thr = gct_drop_threshold(p); g.keep_scale = 1.0f / (1.0f - p); g.rng = gct_rng_make(seed, site);
  g.bp0 = wp0; g.bp_stride = pstride;
  g.quad_map = quad_map;
  g.pre_rows = quad_map ? pre_rows : 0;
  const bool vec = al16(dy0) && al16(dy1) && al16(dy2) && al16(w0) && al16(w1) && al16(w2) &&
                   (lddy % 4 == 0) && (ldw % 4 == 0) && (nper % 4 == 0) && (K % 4 == 0);
  return launch<true, false>(g, vec, (hipStream_t)stream, ws, ws ? ws_bytes : 0);
}

extern "C" int gct_linear_dgrad(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                                int64_t M, int nseg, int nper, const float* w0, const float* w1,
                                const float* w2, int64_t ldw, int K, float* dx, int64_t lddx,
                                int depi, const float* pre, float p, uint64_t seed, uint32_t site,
                                void* stream) {
  return linear_dgrad_impl(dy0, dy1, dy2, lddy, M, nseg, nper, w0, w1, w2, ldw, K, dx, lddx, depi, pre,
                           p, seed, site, stream, nullptr, 0, nullptr, 0);
}

extern "C" int gct_linear_dgrad_p(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                                  int64_t M, int nseg, int nper, const float* w0, const float* w1,
                                  const float* w2, int64_t ldw, const uint16_t* wp0,
                                  int64_t plane_stride, int K, float* dx, int64_t lddx, int depi,
                                  const float* pre, float p, uint64_t seed, uint32_t site,
                                  float* ws, int64_t ws_bytes, const int32_t* quad_map, int64_t pre_rows,
                                  void* stream) {
  GCT_CHECK_ARG(ws_bytes >= 0, "linear_dgrad: negative workspace size");
  return linear_dgrad_impl(dy0, dy1, dy2, lddy, M, nseg, nper, w0, w1, w2, ldw, K, dx, lddx, depi, pre,
                           p, seed, site, stream, wp0, plane_stride, ws, ws_bytes, quad_map, pre_rows);
}

static int linear_wgrad_impl(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                             int64_t M, int nseg, int nper, const float* x, int64_t ldx, int K,
                             float* dw0, float* dw1, float* dw2, int64_t lddw, float* db0,
                             float* db1, float* db2, float* ws, void* stream, const int32_t* kt_list,
                             const int32_t* kt_count) {
  GCT_CHECK_ARG(dy0 && x && dw0 && ws && M >= 0 && K > 0 && nseg >= 1 && nseg <= 3 && nper > 0,
                "linear_wgrad: bad args");
  GCT_CHECK_ARG(nseg < 2 || (dy1 && dw1), "linear_wgrad: missing segment 1");
  GCT_CHECK_ARG(nseg < 3 || (dy2 && dw2), "linear_wgrad: missing segment 2");
  GCT_CHECK_ARG(lddw == K, "linear_wgrad: dW must be dense [nper][K]");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Ntot = (int64_t)nseg * nper;
  const bool vec = al16(dy0) && al16(dy1) && al16(dy2) && al16(x) && (lddy % 4 == 0) &&
                   (ldx % 4 == 0) && (nper % 4 == 0) && (K % 4 == 0);
  GemmArgs g = {};
  g.M = Ntot; g.N = K; g.K = M;  // dW[n][k] = sum_m dY[m][n] X[m][k]
  g.a = mkseg(dy0, dy1, dy2); g.lda = lddy; g.a_nper = nper;
  g.b = mkseg(x, nullptr, nullptr); g.ldb = ldx; g.b_nper = INT64_MAX / 4;
  g.c0 = ws; g.ldc = K; g.c_nper = INT64_MAX / 4;
  g.slab_stride = Ntot * K;
  g.ksplit = BK; g.nsplit = 1; g.epi = EPI_SLAB;
  const bool use_x6 = gemm_mode() == GCT_GEMM_BF16X6 && x6_ok<X6_WGRAD>(g, vec);
  if (use_x6 && kt_list && kt_count) { g.kt_list = kt_list; g.kt_count = kt_count; }   // other kernels reduce over every row
  const int splits = wgrad_splits(M, Ntot, K, use_x6);
  int64_t ks = (M + splits - 1) / splits;
  ks = (ks + BK - 1) / BK * BK;
  g.ksplit = ks > 0 ? ks : BK;
  g.nsplit = (int)((M + g.ksplit - 1) / g.ksplit);
  if (g.nsplit < 1) g.nsplit = 1;
  g.epi = EPI_SLAB;
  // bias gradients: fused into the GEMM on the fast path (column sums of the A tiles), else a
  // separate column-sum pass that borrows ws before the slabs are written (stream-ordered)
  float* bslab = nullptr;
  if (db0) {
    if (use_x6 || fast_ok<false, false>(g, vec)) {
      int64_t off = (int64_t)g.nsplit * g.slab_stride;
      off = (off + 3) / 4 * 4;
      bslab = ws + off;                     // [nsplit][Ntot] right behind the weight slabs
      g.bias_slab = bslab;
    } else {
      int rc = gct_colsum(dy0, dy1, dy2, lddy, M, nseg, nper, db0, db1, db2, ws, st);
      if (rc) return rc;
    }
  }
  int rc = launch<false, false>(g, vec, st);
  if (rc) return rc;
  if (bslab)     // weight and bias slabs in one launch
    return gct_reduce_slabs_seg2(ws, g.nsplit, g.slab_stride, dw0, dw1, dw2, (int64_t)nper * K, Ntot * K, bslab, Ntot,
                                 db0, db1, db2, nper, Ntot, st);
  return gct_reduce_slabs_seg(ws, g.nsplit, g.slab_stride, dw0, dw1, dw2, (int64_t)nper * K, Ntot * K, st);
}

extern "C" int gct_linear_wgrad(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                                int64_t M, int nseg, int nper, const float* x, int64_t ldx, int K,
                                float* dw0, float* dw1, float* dw2, int64_t lddw, float* db0,
                                float* db1, float* db2, float* ws, void* stream) {
  return linear_wgrad_impl(dy0, dy1, dy2, lddy, M, nseg, nper, x, ldx, K, dw0, dw1, dw2, lddw, db0, db1, db2,
                           ws, stream, nullptr, nullptr);
}

extern "C" int gct_linear_wgrad_kt(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                                   int64_t M, int nseg, int nper, const float* x, int64_t ldx, int K,
                                   float* dw0, float* dw1, float* dw2, int64_t lddw, float* db0,
                                   float* db1, float* db2, float* ws, const int32_t* kt_list,
                                   const int32_t* kt_count, void* stream) {
  return linear_wgrad_impl(dy0, dy1, dy2, lddy, M, nseg, nper, x, ldx, K, dw0, dw1, dw2, lddw, db0, db1, db2,
                           ws, stream, kt_list, kt_count);
}
