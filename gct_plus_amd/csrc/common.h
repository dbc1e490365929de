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

// Dropout convention for [rows][cols] activations: one Philox call serves an aligned patch of
// 4 rows x 2 columns -- element (row, col) takes the 16-bit lane (row & 3) * 2 + (col & 1) of
// philox(row >> 2, col >> 1) and is kept iff that lane >= thr >> 16 (P(drop) = floor(p * 2^16) / 2^16).
// Every kernel that applies or re-creates a mask goes through these two helpers.
__device__ __forceinline__ uint4 gct_drop_bits(GctRng rng, uint32_t row4, uint32_t col) {
  return gct_philox(rng, row4, col >> 1, 0x243F6A88u, 0x85A308D3u);
}

__device__ __forceinline__ uint32_t gct_pick(uint4 v, int c) {
  return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// e = row & 3; bits = gct_drop_bits(rng, row >> 2, col)
__device__ __forceinline__ bool gct_drop_keep(uint4 bits, int e, uint32_t col, uint32_t thr) {
  const uint32_t w = gct_pick(bits, e);
  return ((col & 1u) ? (w >> 16) : (w & 0xffffu)) >= (thr >> 16);
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

// erf(x) = sign(x) * (1 - exp(-|x| * P(|x|))), P of degree 7 fitted to -ln(erfc) on [0, 3.94] (beyond it
// erf rounds to 1 in fp32): branch-free, 8 fma + 1 v_exp_f32; max abs error 1.3e-7 (ocml's erff costs about
// twice the instructions because both of its branches are evaluated under divergence).  GELU built on it:
// max abs error 4.7e-7 over [-6, 6] against fp64.
__device__ __forceinline__ float gct_erf(float x) {
  const float a = fminf(fabsf(x), 3.9375f);
  float p = 3.144016591e-05f;
  p = fmaf(p, a, -3.088021767e-04f);
  p = fmaf(p, a, 1.032401458e-03f);
  p = fmaf(p, a, 5.369338905e-04f);
  p = fmaf(p, a, -1.958396100e-02f);
  p = fmaf(p, a, 1.029196158e-01f);
  p = fmaf(p, a, 6.365977526e-01f);
  p = fmaf(p, a, 1.128380299e+00f);
  const float e = 1.0f - __builtin_amdgcn_exp2f(p * a * -1.4426950408889634f);   // argument in [-32, 0]: no denormal handling needed
  return copysignf(e, x);
}
__device__ __forceinline__ float gct_gelu(float x) {  // erf form (F.gelu default)
  return 0.5f * x * (1.0f + gct_erf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gct_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + gct_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
