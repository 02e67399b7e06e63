// Shared device helpers for the gfx950 SPLIT-VAE kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/splitvae.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SV_LAUNCH_CHECK()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int EPP = 4; };     // elements per 16-B piece
template <> struct ElemTraits<bf16_t> { static constexpr int EPP = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 64-lane butterfly sum (wavefront shuffles; every lane ends with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// numerically stable helpers used by the ELBO kernels
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) {
  // 1/(1+exp(-x)) without overflow for large |x|
  float e = __expf(-fabsf(x));
  float s = 1.f / (1.f + e);
  return x >= 0.f ? s : e * s;
}

// ---------------------------------------------------------------- Philox4x32-10 (counter-based RNG)
struct Philox {
  uint32_t k0, k1;
  __host__ __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  __host__ __device__ inline void operator()(uint32_t c[4]) const {
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0, lo0, hi1, lo1;
      mulhilo(0xD2511F53u, c[0], hi0, lo0);
      mulhilo(0xCD9E8D57u, c[2], hi1, lo1);
      uint32_t n0 = hi1 ^ c[1] ^ a, n1 = lo1, n2 = hi0 ^ c[3] ^ b, n3 = lo0;
      c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
  }
};

__device__ __forceinline__ float u32_to_unit_open(uint32_t u) {   // (0,1]
  return ((float)(u >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

static inline int ilog2_exact(int v) {   // -1 if not a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
