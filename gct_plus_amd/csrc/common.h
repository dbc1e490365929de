// Shared device/host helpers for libgctplus_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gctplus_hip.h"

#define GCT_WAVE 64

void gct_set_error(const char* fmt, ...);

#define GCT_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      gct_set_error(__VA_ARGS__);       \
      return GCT_ERR_ARG;               \
    }                                   \
  } while (0)

#define GCT_LAUNCH_CHECK(name)                                            \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      gct_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return GCT_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

static inline bool gct_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------- Philox4x32-10
// key = (seed_lo, seed_hi ^ site); counter = caller-defined element coordinates.
struct GctRng {
  uint32_t k0, k1;
};

__host__ __device__ inline GctRng gct_rng_make(uint64_t seed, uint32_t site) {
  GctRng r;
  r.k0 = (uint32_t)seed;
  r.k1 = (uint32_t)(seed >> 32) ^ (site * 0x9E3779B9u + 0x7F4A7C15u);
  return r;
}

__device__ __forceinline__ uint4 gct_philox(GctRng rng, uint32_t c0, uint32_t c1, uint32_t c2,
                                            uint32_t c3) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t k0 = rng.k0, k1 = rng.k1;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return make_uint4(c0, c1, c2, c3);
}

// keep threshold: keep iff u32 >= thr, P(drop) = thr / 2^32
__host__ __device__ inline uint32_t gct_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}

// Row-tile convention for [rows][cols] activations: one Philox call serves the 4
// vertically adjacent elements (row&~3 .. +3, col); component = row & 3.
__device__ __forceinline__ uint4 gct_drop_bits(GctRng rng, uint32_t row4, uint32_t col) {
  return gct_philox(rng, row4, col, 0x243F6A88u, 0x85A308D3u);
}

__device__ __forceinline__ uint32_t gct_pick(uint4 v, int c) {
  return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ float gct_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gct_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (id % 8)
// get a contiguous range of logical ids (cdna guide T1).
__device__ __forceinline__ unsigned gct_xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, idx = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float gct_gelu(float x) {  // exact erf form (F.gelu default)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gct_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
