// Prototype (never shipped): fp32-accurate GEMM on the bf16 MFMA pipes.
//   C[M][N] = A[M][K] * W[N][K]^T, A fp32 split on the fly into three bf16 pieces (exact 24-bit
//   decomposition, round-to-nearest at every level), W pre-split into planes [3][N][K]; the six leading
//   partial products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_x6_proto.hip -o tools/_build/gemm_x6_proto
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// two fp32 -> three packed bf16 pairs (h, m, l) with a + b == exactly h + m + l per element
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk_bf16(a, b);
  const float ra = a - lo_f(h), rb = b - hi_f(h);
  m = pk_bf16(ra, rb);
  const float sa = ra - lo_f(m), sb = rb - hi_f(m);
  l = pk_bf16(sa, sb);
}

__global__ void split_planes_kernel(const float* __restrict__ w, unsigned* __restrict__ planes, int64_t n_pairs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const float2 v = reinterpret_cast<const float2*>(w)[i];
  unsigned h, m, l;
  split_pair(v.x, v.y, h, m, l);
  planes[i] = h; planes[n_pairs + i] = m; planes[2 * n_pairs + i] = l;
}

__device__ __forceinline__ int lds_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// 128x128x32 tile, 4 waves (2x2 of 64x64), LDS: A planes 3 x 8 KB, B planes 3 x 8 KB
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                         float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * 8192];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int tiles_n = N / 128;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int64_t plane = (int64_t)N * K;

  // staging coordinates
  const int a_c4 = t & 7, a_r = t >> 3;                 // A: float4 column, rows a_r + 32 j
  const int b_c = t & 3, b_r = t >> 2;                  // B: 16-B chunk, rows b_r + 64 j
  const float* ag = A + (int64_t)(bm * 128 + a_r) * K + a_c4 * 4;
  const unsigned short* bg = Wp + (int64_t)(bn * 128 + b_r) * K + b_c * 8;

  float4 pa[4];
  u32x4 pb[3][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pa[j] = *reinterpret_cast<const float4*>(ag + (int64_t)j * 32 * K + k0);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) pb[p][j] = *reinterpret_cast<const u32x4*>(bg + p * plane + (int64_t)j * 64 * K + k0);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h0, m0, l0, h1, m1, l1;
      split_pair(pa[j].x, pa[j].y, h0, m0, l0);
      split_pair(pa[j].z, pa[j].w, h1, m1, l1);
      const int r = a_r + 32 * j;
      const int off = lds_off(r, a_c4 >> 1) + (a_c4 & 1) * 8;
      *reinterpret_cast<u32x2*>(lds + 0 * 8192 + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(lds + 1 * 8192 + off) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(lds + 2 * 8192 + off) = u32x2{l0, l1};
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<u32x4*>(lds + (3 + p) * 8192 + lds_off(b_r + 64 * j, b_c)) = pb[p][j];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore();
  __syncthreads();
  const int fr = l & 31, fh = l >> 5;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const bool more = k0 + 32 < K;
    if (more) gload(k0 + 32);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = *reinterpret_cast<const bf16x8*>(lds + p * 8192 + lds_off(64 * wm + 32 * i + fr, 2 * s + fh));
          fb[i][p] = *reinterpret_cast<const bf16x8*>(lds + (3 + p) * 8192 + lds_off(64 * wn + 32 * i + fr, 2 * s + fh));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          MFMA(fa[i][2], fb[j][0], acc[i][j]);
          MFMA(fa[i][0], fb[j][2], acc[i][j]);
          MFMA(fa[i][1], fb[j][1], acc[i][j]);
          MFMA(fa[i][1], fb[j][0], acc[i][j]);
          MFMA(fa[i][0], fb[j][1], acc[i][j]);
          MFMA(fa[i][0], fb[j][0], acc[i][j]);
        }
    }
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm * 128 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int col = bn * 128 + 64 * wn + 32 * j + fr;
        C[(int64_t)row * N + col] = acc[i][j][r];
      }
}



// ---- v0 with v_mfma_f32_16x16x32_bf16 (A/B test of the MFMA shape: same tile, same staging, same loop)
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int lds_off16(int r, int c) {
  const int g = (0x78 >> (2 * ((r >> 2) & 3))) & 3;     // {0,2,3,1}: conflict-free for the 16x16x32 row reads
  return r * 64 + ((c ^ g) << 4);
}
__global__ __launch_bounds__(256, 2) void gemm_x6_v0_16_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                               float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * 8192];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int tiles_n = N / 128;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int64_t plane = (int64_t)N * K;
  const int a_c4 = t & 7, a_r = t >> 3;
  const int b_c = t & 3, b_r = t >> 2;
  const float* ag = A + (int64_t)(bm * 128 + a_r) * K + a_c4 * 4;
  const unsigned short* bg = Wp + (int64_t)(bn * 128 + b_r) * K + b_c * 8;
  float4 pa[4];
  u32x4 pb[3][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pa[j] = *reinterpret_cast<const float4*>(ag + (int64_t)j * 32 * K + k0);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) pb[p][j] = *reinterpret_cast<const u32x4*>(bg + p * plane + (int64_t)j * 64 * K + k0);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h0, m0, l0, h1, m1, l1;
      split_pair(pa[j].x, pa[j].y, h0, m0, l0);
      split_pair(pa[j].z, pa[j].w, h1, m1, l1);
      const int off = lds_off16(a_r + 32 * j, a_c4 >> 1) + (a_c4 & 1) * 8;
      *reinterpret_cast<u32x2*>(lds + 0 * 8192 + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(lds + 1 * 8192 + off) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(lds + 2 * 8192 + off) = u32x2{l0, l1};
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<u32x4*>(lds + (3 + p) * 8192 + lds_off16(b_r + 64 * j, b_c)) = pb[p][j];
  };
  f32x4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  gload(0);
  lstore();
  __syncthreads();
  const int fr = l & 15, fc = l >> 4;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const bool more = k0 + 32 < K;
    if (more) gload(k0 + 32);
    bf16x8 fa[4][3], fb[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(lds + p * 8192 + lds_off16(64 * wm + 16 * i + fr, fc));
        fb[i][p] = *reinterpret_cast<const bf16x8*>(lds + (3 + p) * 8192 + lds_off16(64 * wn + 16 * i + fr, fc));
      }
#define MFMA16(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        MFMA16(fa[i][2], fb[j][0], acc[i][j]);
        MFMA16(fa[i][0], fb[j][2], acc[i][j]);
        MFMA16(fa[i][1], fb[j][1], acc[i][j]);
        MFMA16(fa[i][1], fb[j][0], acc[i][j]);
        MFMA16(fa[i][0], fb[j][1], acc[i][j]);
        MFMA16(fa[i][0], fb[j][0], acc[i][j]);
      }
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bm * 128 + 64 * wm + 16 * i + fc * 4 + r;
        const int col = bn * 128 + 64 * wn + 16 * j + fr;
        C[(int64_t)row * N + col] = acc[i][j][r];
      }
}

// ---- v1: 128x256x32 tile, 8 waves (2x4 of 64x64), double-buffered LDS (2 x 72 KB), one barrier per K-tile
constexpr int STAGE = 72 * 1024, A_PLANE = 8192, B_PLANE = 16384;
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void
gemm_x6_v1_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int tiles_n = N / 256;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int64_t plane = (int64_t)N * K;
  const int a_c4 = t & 7, a_r = t >> 3;                 // rows a_r + 64 j
  const int b_c = t & 3, b_r = t >> 2;                  // rows b_r + 128 j
  const float* ag = A + (int64_t)(bm * 128 + a_r) * K + a_c4 * 4;
  const unsigned short* bg = Wp + (int64_t)(bn * 256 + b_r) * K + b_c * 8;
  const int a_off0 = lds_off(a_r, a_c4 >> 1) + (a_c4 & 1) * 8, a_off1 = lds_off(a_r + 64, a_c4 >> 1) + (a_c4 & 1) * 8;
  const int b_off0 = 3 * A_PLANE + lds_off(b_r, b_c), b_off1 = 3 * A_PLANE + lds_off(b_r + 128, b_c);

  float4 pa[2];
  u32x4 pb[3][2];
#define GLOAD(k0_)                                                                               \
  do {                                                                                           \
    pa[0] = *reinterpret_cast<const float4*>(ag + (k0_));                                        \
    pa[1] = *reinterpret_cast<const float4*>(ag + (int64_t)64 * K + (k0_));                      \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                              \
      pb[p][0] = *reinterpret_cast<const u32x4*>(bg + p * plane + (k0_));                        \
      pb[p][1] = *reinterpret_cast<const u32x4*>(bg + p * plane + (int64_t)128 * K + (k0_));     \
    }                                                                                            \
  } while (0)
#define LSTORE(buf_)                                                                             \
  do {                                                                                           \
    unsigned char* st__ = lds + (buf_) * STAGE;                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
      unsigned h0, m0, l0, h1, m1, l1;                                                           \
      split_pair(pa[j].x, pa[j].y, h0, m0, l0);                                                  \
      split_pair(pa[j].z, pa[j].w, h1, m1, l1);                                                  \
      const int off = j ? a_off1 : a_off0;                                                       \
      *reinterpret_cast<u32x2*>(st__ + 0 * A_PLANE + off) = u32x2{h0, h1};                       \
      *reinterpret_cast<u32x2*>(st__ + 1 * A_PLANE + off) = u32x2{m0, m1};                       \
      *reinterpret_cast<u32x2*>(st__ + 2 * A_PLANE + off) = u32x2{l0, l1};                       \
    }                                                                                            \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                              \
      *reinterpret_cast<u32x4*>(st__ + p * B_PLANE + b_off0) = pb[p][0];                         \
      *reinterpret_cast<u32x4*>(st__ + p * B_PLANE + b_off1) = pb[p][1];                         \
    }                                                                                            \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = l & 31, fh = l >> 5;
  int fa_off[2][2], fb_off[2][2];  // [s][i]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa_off[s][i] = lds_off(64 * wm + 32 * i + fr, 2 * s + fh);
      fb_off[s][i] = 3 * A_PLANE + lds_off(64 * wn + 32 * i + fr, 2 * s + fh);
    }
  bf16x8 fa[2][3], fb[2][3];
#define FRAGS(st_, s_)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                 \
    fa[i][p] = *reinterpret_cast<const bf16x8*>((st_) + p * A_PLANE + fa_off[s_][i]);             \
    fb[i][p] = *reinterpret_cast<const bf16x8*>((st_) + p * B_PLANE + fb_off[s_][i]);             \
  }
#define MFMA24()                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                 \
    MFMA(fa[i][2], fb[j][0], acc[i][j]);                                                          \
    MFMA(fa[i][0], fb[j][2], acc[i][j]);                                                          \
    MFMA(fa[i][1], fb[j][1], acc[i][j]);                                                          \
    MFMA(fa[i][1], fb[j][0], acc[i][j]);                                                          \
    MFMA(fa[i][0], fb[j][1], acc[i][j]);                                                          \
    MFMA(fa[i][0], fb[j][0], acc[i][j]);                                                          \
  }

  bf16x8 ga[2][3], gb[2][3];
#define FRAGS2(fa_, fb_, st_, s_)                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                 \
    fa_[i][p] = *reinterpret_cast<const bf16x8*>((st_) + p * A_PLANE + fa_off[s_][i]);            \
    fb_[i][p] = *reinterpret_cast<const bf16x8*>((st_) + p * B_PLANE + fb_off[s_][i]);            \
  }
#define MFMA24X(fa_, fb_)                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                 \
    MFMA(fa_[i][2], fb_[j][0], acc[i][j]);                                                        \
    MFMA(fa_[i][0], fb_[j][2], acc[i][j]);                                                        \
    MFMA(fa_[i][1], fb_[j][1], acc[i][j]);                                                        \
    MFMA(fa_[i][1], fb_[j][0], acc[i][j]);                                                        \
    MFMA(fa_[i][0], fb_[j][1], acc[i][j]);                                                        \
    MFMA(fa_[i][0], fb_[j][0], acc[i][j]);                                                        \
  }
  const int nkt = K / 32;
  GLOAD(0);
  LSTORE(0);
  GLOAD(nkt > 1 ? 32 : 0);
  __syncthreads();
  FRAGS2(fa, fb, lds, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned char* st = lds + (kt & 1) * STAGE;
    const unsigned char* stn = lds + ((kt + 1) & 1) * STAGE;
    const int knext = (kt + 2 < nkt) ? (kt + 2) * 32 : kt * 32;
    __builtin_amdgcn_sched_barrier(0);
    // first half: MFMAs of (tile kt, s=0); fragment reads of s=1; split + LDS stores of tile kt+1
    FRAGS2(ga, gb, st, 1);
    LSTORE((kt + 1) & 1);
    MFMA24X(fa, fb);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 LDS write
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // second half: MFMAs of (tile kt, s=1); fragment reads of (tile kt+1, s=0); global loads of tile kt+2
    FRAGS2(fa, fb, stn, 0);
    GLOAD(knext);
    MFMA24X(ga, gb);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm * 128 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int col = bn * 256 + 64 * wn + 32 * j + fr;
        C[(int64_t)row * N + col] = acc[i][j][r];
      }
}

// plain fp32 reference on the device (fmaf chain) for the error comparison
__global__ void ref_f32_kernel(const float* A, const float* W, float* C, const int* rows, const int* cols, int n, int K, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(A[(int64_t)rows[i] * K + k], W[(int64_t)cols[i] * K + k], s);
  C[i] = s;
}

template <class F> static double time_it(F f) {
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}
static void run(int M, int K, int N) {
  float *a, *w, *c;
  unsigned short* wp;
  hipMalloc(&a, (size_t)M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&c, (size_t)M * N * 4);
  hipMalloc(&wp, (size_t)3 * N * K * 2);
  std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
  for (auto& v : ha) v = rnd() * 1.7f;
  for (auto& v : hw) v = rnd() * 0.05f;
  hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int64_t pairs = (int64_t)N * K / 2;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, 0, w, (unsigned*)wp, pairs);
  const unsigned tiles = (M / 128) * (N / 128);
  const double us0 = time_it([&] { hipLaunchKernelGGL(gemm_x6_kernel, dim3(tiles), dim3(256), 0, 0, a, wp, c, M, N, K); });
  hipFuncSetAttribute((const void*)gemm_x6_v1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipMemset(c, 0, (size_t)M * N * 4);
  const unsigned tiles1 = (M / 128) * (N / 256);
  const double us = time_it([&] { hipLaunchKernelGGL(gemm_x6_v1_kernel, dim3(tiles1), dim3(512), 2 * STAGE, 0, a, wp, c, M, N, K); });
  const double us16 = time_it([&] { hipLaunchKernelGGL(gemm_x6_v0_16_kernel, dim3(tiles), dim3(256), 0, 0, a, wp, c, M, N, K); });
  {   // spot-check the 16x16x32 variant against the 32x32x16 one
    std::vector<float> c16(256), c32(256);
    hipMemcpy(c16.data(), c + (int64_t)777 * N, 256 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(gemm_x6_kernel, dim3(tiles), dim3(256), 0, 0, a, wp, c, M, N, K);
    hipMemcpy(c32.data(), c + (int64_t)777 * N, 256 * 4, hipMemcpyDeviceToHost);
    double md = 0; for (int i = 0; i < 256; ++i) md = std::fmax(md, std::fabs((double)c16[i] - c32[i]));
    printf("[16x16x32 vs 32x32x16 max diff %.2e] ", md);
  }
  printf("v0 %.1f us %.1f TF | v0/16x16x32 %.1f us %.1f TF | ", us0, 2.0 * M * K * N / us0 * 1e-6, us16, 2.0 * M * K * N / us16 * 1e-6);
  // error vs fp64 on a sample, beside the fp32 fmaf chain
  const int ns = 4096;
  std::vector<int> rows(ns), cols(ns);
  for (int i = 0; i < ns; ++i) { rows[i] = (int)((i * 2654435761ull) % M); cols[i] = (int)((i * 40503ull + 7) % N); }
  int *dr, *dc; float* dref;
  hipMalloc(&dr, ns * 4); hipMalloc(&dc, ns * 4); hipMalloc(&dref, ns * 4);
  hipMemcpy(dr, rows.data(), ns * 4, hipMemcpyHostToDevice);
  hipMemcpy(dc, cols.data(), ns * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(ref_f32_kernel, dim3(ns / 256), dim3(256), 0, 0, a, w, dref, dr, dc, ns, K, N);
  std::vector<float> href(ns), hc(ns);
  hipMemcpy(href.data(), dref, ns * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < ns; ++i) hipMemcpy(&hc[i], c + (int64_t)rows[i] * N + cols[i], 4, hipMemcpyDeviceToHost);
  double e_x6 = 0, e_f32 = 0, scale = 0, m_x6 = 0, m_f32 = 0;
  for (int i = 0; i < ns; ++i) {
    double ex = 0, ab = 0;
    for (int k = 0; k < K; ++k) {
      const double p = (double)ha[(size_t)rows[i] * K + k] * (double)hw[(size_t)cols[i] * K + k];
      ex += p; ab += std::fabs(p);
    }
    const double d1 = std::fabs(hc[i] - ex) / ab, d2 = std::fabs(href[i] - ex) / ab;
    e_x6 += d1; e_f32 += d2; m_x6 = std::fmax(m_x6, d1); m_f32 = std::fmax(m_f32, d2);
    scale += 1;
  }
  printf("M=%d K=%d N=%d: %.1f us  %.1f TF-equivalent | err/sum|ab|: x6 mean %.3g max %.3g ; fp32 fmaf chain mean %.3g max %.3g\n",
         M, K, N, us, 2.0 * M * K * N / us * 1e-6, e_x6 / scale, m_x6, e_f32 / scale, m_f32);
  hipFree(a); hipFree(w); hipFree(c); hipFree(wp); hipFree(dr); hipFree(dc); hipFree(dref);
}

int main() {
  run(40960, 512, 2048);
  run(40960, 2048, 512);
  run(40960, 512, 512);
  run(4096, 4096, 4096);
  return 0;
}
