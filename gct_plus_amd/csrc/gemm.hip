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
};

enum { EPI_SLAB = 32, EPI_D0 = 16 };

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

// ---- LDS -> fragments: frag[t][s] holds k = 16*h + s of free-dim index base+32t+(lane&31)
__device__ __forceinline__ void frag_kc(float (&f)[2][16], const float* lds, int base, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = base + t * 32 + (lane & 31);
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(lds + row * BK + (((h * 4 + c) ^ sw) * 4));
      f[t][c * 4 + 0] = v.x;
      f[t][c * 4 + 1] = v.y;
      f[t][c * 4 + 2] = v.z;
      f[t][c * 4 + 3] = v.w;
    }
  }
}
__device__ __forceinline__ void frag_rc(float (&f)[2][16], const float* lds, int base, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = base + t * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < 16; ++s) f[t][s] = lds[(h * 16 + s) * BM + col];
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
  if (nkt > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nkt) gload(kbeg + (kt + 1) * BK);  // in flight during the MFMAs below
    const float* la = lds + cur * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
    float fa[2][16], fb[2][16];
    if (A_KC) frag_kc(fa, la, wm, lane); else frag_rc(fa, la, wm, lane);
    if (B_KC) frag_kc(fb, lb, wn, lane); else frag_rc(fb, lb, wn, lane);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
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
        if (need_rng) bits = gct_drop_bits(g.rng, (uint32_t)(row_base >> 2), (uint32_t)col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t row = row_base + e;
          if (row >= g.M) continue;
          float v = acc[i][j][q4 * 4 + e];
          const int64_t off = row * g.ldc + cloc;
          const bool keep = gct_pick(bits, e) >= g.thr;
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
              const float u = g.pre_in[off];
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
}

template <bool A_KC, bool B_KC>
int launch(const GemmArgs& g, bool vec, hipStream_t st) {
  const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * g.nsplit;
  if (tiles <= 0) return GCT_OK;
  if (tiles > INT_MAX) {
    gct_set_error("gemm: grid too large");
    return GCT_ERR_ARG;
  }
  dim3 grid((unsigned)tiles), block(256);
  if (vec)
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
int gct_colsum(const float* y0, const float* y1, const float* y2, int64_t ld, int64_t M, int nseg,
               int nper, float* d0, float* d1, float* d2, float* ws, hipStream_t st);
int64_t gct_colsum_ws_floats(int64_t M, int64_t N);

static int wgrad_splits(int64_t M, int64_t Ntot, int64_t K) {
  const int64_t tiles = ((Ntot + BM - 1) / BM) * ((K + BN - 1) / BN);
  // exactly one resident round: 256 CUs x 2 blocks (64 KB LDS each) = 512 blocks, never 513
  int64_t want = 512 / tiles;
  const int64_t maxs = (M + 4 * BK - 1) / (4 * BK);
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int)want;
}

extern "C" int64_t gct_wgrad_ws_bytes(int64_t M, int64_t Ntot, int64_t K) {
  const int s = wgrad_splits(M, Ntot, K);
  const int64_t slab = (int64_t)s * Ntot * K;
  const int64_t cs = gct_colsum_ws_floats(M, Ntot);
  return (slab > cs ? slab : cs) * (int64_t)sizeof(float) + 256;
}

extern "C" int gct_linear_fwd(const float* x, int64_t ldx, int64_t M, int K, const float* w0,
                              const float* w1, const float* w2, int64_t ldw, const float* b0,
                              const float* b1, const float* b2, int nseg, int nper, float* y0,
                              float* y1, float* y2, int64_t ldy, int epi, const float* resid,
                              float* pre, float p, uint64_t seed, uint32_t site, void* stream) {
  GCT_CHECK_ARG(x && w0 && y0 && M >= 0 && K > 0 && nseg >= 1 && nseg <= 3 && nper > 0,
                "linear_fwd: bad args");
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
  const bool vec = al16(x) && al16(w0) && al16(w1) && al16(w2) && (ldx % 4 == 0) &&
                   (ldw % 4 == 0) && (K % 4 == 0);
  return launch<true, true>(g, vec, (hipStream_t)stream);
}

extern "C" int gct_linear_dgrad(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                                int64_t M, int nseg, int nper, const float* w0, const float* w1,
                                const float* w2, int64_t ldw, int K, float* dx, int64_t lddx,
                                int depi, const float* pre, float p, uint64_t seed, uint32_t site,
                                void* stream) {
  GCT_CHECK_ARG(dy0 && w0 && dx && M >= 0 && K > 0 && nseg >= 1 && nseg <= 3 && nper > 0,
                "linear_dgrad: bad args");
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
  const bool vec = al16(dy0) && al16(dy1) && al16(dy2) && al16(w0) && al16(w1) && al16(w2) &&
                   (lddy % 4 == 0) && (ldw % 4 == 0) && (nper % 4 == 0) && (K % 4 == 0);
  return launch<true, false>(g, vec, (hipStream_t)stream);
}

extern "C" int gct_linear_wgrad(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                                int64_t M, int nseg, int nper, const float* x, int64_t ldx, int K,
                                float* dw0, float* dw1, float* dw2, int64_t lddw, float* db0,
                                float* db1, float* db2, float* ws, void* stream) {
  GCT_CHECK_ARG(dy0 && x && dw0 && ws && M >= 0 && K > 0 && nseg >= 1 && nseg <= 3 && nper > 0,
                "linear_wgrad: bad args");
  GCT_CHECK_ARG(nseg < 2 || (dy1 && dw1), "linear_wgrad: missing segment 1");
  GCT_CHECK_ARG(nseg < 3 || (dy2 && dw2), "linear_wgrad: missing segment 2");
  GCT_CHECK_ARG(lddw == K, "linear_wgrad: dW must be dense [nper][K]");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Ntot = (int64_t)nseg * nper;
  if (db0) {  // bias gradients first: they share ws with the slabs (stream-ordered)
    int rc = gct_colsum(dy0, dy1, dy2, lddy, M, nseg, nper, db0, db1, db2, ws, st);
    if (rc) return rc;
  }
  const int splits = wgrad_splits(M, Ntot, K);
  GemmArgs g = {};
  g.M = Ntot; g.N = K; g.K = M;  // dW[n][k] = sum_m dY[m][n] X[m][k]
  g.a = mkseg(dy0, dy1, dy2); g.lda = lddy; g.a_nper = nper;
  g.b = mkseg(x, nullptr, nullptr); g.ldb = ldx; g.b_nper = INT64_MAX / 4;
  g.c0 = ws; g.ldc = K; g.c_nper = INT64_MAX / 4;
  g.slab_stride = Ntot * K;
  int64_t ks = (M + splits - 1) / splits;
  ks = (ks + BK - 1) / BK * BK;
  g.ksplit = ks > 0 ? ks : BK;
  g.nsplit = (int)((M + g.ksplit - 1) / g.ksplit);
  if (g.nsplit < 1) g.nsplit = 1;
  g.epi = EPI_SLAB;
  const bool vec = al16(dy0) && al16(dy1) && al16(dy2) && al16(x) && (lddy % 4 == 0) &&
                   (ldx % 4 == 0) && (nper % 4 == 0) && (K % 4 == 0);
  int rc = launch<false, false>(g, vec, st);
  if (rc) return rc;
  return gct_reduce_slabs_seg(ws, g.nsplit, g.slab_stride, dw0, dw1, dw2, (int64_t)nper * K,
                              Ntot * K, st);
}
