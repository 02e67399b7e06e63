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
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// ---- 16-B piece <-> packed fp32 pairs (v_pk_* VALU ops take two floats per instruction)
template <typename T> struct Piece;
template <> struct Piece<bf16_t> {
  static constexpr int NP = 4;   // f32x2 pairs per piece
  static __device__ __forceinline__ void unpack(const uint4& u, f32x2 (&f)[4]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (f32x2){__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
  }
  static __device__ __forceinline__ uint4 pack(const f32x2 (&f)[4]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x2 b = __builtin_convertvector(f[i], bf16x2);   // v_cvt_pk_bf16_f32 (round to nearest even, NaN kept)
      w[i] = __builtin_bit_cast(uint32_t, b);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct Piece<float> {
  static constexpr int NP = 2;
  static __device__ __forceinline__ void unpack(const uint4& u, f32x2 (&f)[2]) {
    f[0] = (f32x2){__uint_as_float(u.x), __uint_as_float(u.y)};
    f[1] = (f32x2){__uint_as_float(u.z), __uint_as_float(u.w)};
  }
  static __device__ __forceinline__ uint4 pack(const f32x2 (&f)[2]) {
    return make_uint4(__float_as_uint(f[0][0]), __float_as_uint(f[0][1]), __float_as_uint(f[1][0]), __float_as_uint(f[1][1]));
  }
};
// a + (b - a) * w as one subtract and one fused multiply-add per pair: THE interpolation arithmetic of
// the 2x bilinear resize, shared by upsample2x_fwd_kernel and the fused tile staging (bitwise equal)
__device__ __forceinline__ f32x2 lerp2(f32x2 a, f32x2 b, float w) {
  return __builtin_elementwise_fma(b - a, (f32x2){w, w}, a);
}

// One hi-res row's share of the 2x bilinear resize adjoint (ResizeBilinearGrad, vae/model.py:163-167 backwards) for 8 bf16 channels:
//   acc[e] += wy * (.25 v0[e] + .75 v1[e] + .75 v2[e] + .25 v3[e]),   v0..v3 = the pixels at columns 2j-1 .. 2j+2 (clamped), wy = .25 | .75
// as v_dot2c_f32_bf16 on (column, column + 1) pairs: the products wy * wx (1/16, 3/16, 9/16) are exact in bf16, every product is exact in
// fp32, accumulation in fp32.  Half the VALU instructions of convert + fma per element (v_perm_b32 pairs the two columns' halves).
// upsample2x_bwd_kernel (pointwise.hip) and the row-ring kernel's fused epilogue (row_conv.hip) both go through here, rows a = -1..2 in
// that order: bitwise the same gradient either way.
__device__ __forceinline__ void adj2x_row_bf16(float acc[8], const uint4 v0, const uint4 v1, const uint4 v2, const uint4 v3, const bool outer) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  // (w(2j-1), w(2j)) and (w(2j+1), w(2j+2)) times wy, as packed bf16: 1/16 = 0x3D80, 3/16 = 0x3E40, 9/16 = 0x3F10
  const uint32_t w01 = outer ? 0x3E403D80u : 0x3F103E40u, w23 = outer ? 0x3D803E40u : 0x3E403F10u;
  const uint32_t a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w}, a3[4] = {v3.x, v3.y, v3.z, v3.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t lo01 = __builtin_amdgcn_perm(a1[k], a0[k], 0x05040100u), hi01 = __builtin_amdgcn_perm(a1[k], a0[k], 0x07060302u);
    const uint32_t lo23 = __builtin_amdgcn_perm(a3[k], a2[k], 0x05040100u), hi23 = __builtin_amdgcn_perm(a3[k], a2[k], 0x07060302u);
    acc[2 * k] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, lo01), __builtin_bit_cast(bf2_t, w01), acc[2 * k], false);
    acc[2 * k] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, lo23), __builtin_bit_cast(bf2_t, w23), acc[2 * k], false);
    acc[2 * k + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, hi01), __builtin_bit_cast(bf2_t, w01), acc[2 * k + 1], false);
    acc[2 * k + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, hi23), __builtin_bit_cast(bf2_t, w23), acc[2 * k + 1], false);
  }
}

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// GEMM row m -> (image, oy, ox) of an OY x OX iteration grid: shifts for power-of-two grids (every SPLIT-VAE layer),
// two integer divisions otherwise (wave-uniform choice)
__device__ __forceinline__ void sv_decode_row(int m, int lOY, int lOX, int OY, int OX, int& b, int& oy, int& ox) {
  if (lOY >= 0 && lOX >= 0) {
    b = m >> (lOY + lOX); oy = (m >> lOX) & (OY - 1); ox = m & (OX - 1);
  } else {
    const int pp = OY * OX;
    b = m / pp;
    const int r = m - b * pp;
    oy = r / OX; ox = r - oy * OX;
  }
}

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

// Raise a kernel's dynamic-LDS cap (hipFuncAttributeMaxDynamicSharedMemorySize) when a launch needs more than any launch
// of that kernel ON THAT DEVICE asked for before.  The only process-wide state of the library: a mutex-protected
// (device, kernel) -> bytes table, so concurrent host threads and several devices in one process are safe.
#include <map>
#include <mutex>
#include <utility>
inline void sv_ensure_dynamic_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> have;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& cur = have[std::make_pair(dev, kernel)];
  if (bytes > cur) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    cur = bytes;
  }
}

// SV_DETERMINISTIC=1: every reduction runs in a fixed order -- no split-K / m-split fp32 atomics (one workgroup owns an output tile, or
// partial sums go through slabs summed in index order), bias gradients through per-workgroup partials.  Same inputs twice -> identical
// bits (SURVEY section 5's determinism test; tests/test_gpu_determinism.py).  Off by default: the split-K dense layers are faster with
// atomics.  The environment is read once per process.
// sv_set_deterministic(1 / 0) overrides the environment for the launches that FOLLOW (-1: back to the environment); the choice is made per
// launch on the host, nothing about a plan or its workspace depends on it (tests flip it around individual comparisons).
#include <stdlib.h>
#include <atomic>
inline std::atomic<int>& sv_deterministic_override() {
  static std::atomic<int> v{-1};
  return v;
}
inline bool sv_deterministic() {
  const int o = sv_deterministic_override().load(std::memory_order_relaxed);
  if (o >= 0) return o != 0;
  static const bool d = getenv("SV_DETERMINISTIC") != nullptr && atoi(getenv("SV_DETERMINISTIC")) != 0;
  return d;
}

// Timing-ablation bits (skip staging / the MFMA loop / stores: WRONG results, for profiling only) exist only in builds
// with -DSV_DEBUG_KNOBS (SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py); the shipped library has none.
#ifdef SV_DEBUG_KNOBS
#define SV_DBG(x) (x)
#else
#define SV_DBG(x) 0
#endif

// (wgrad_tile.hip, wgrad_tile_f32.hip) index of (tap, ci, co) in the HWIO gradient; the x-packed conv (fold_kw > 0) has taps (ky, tx) of
// KH x (KW+1) and columns co = px*8 + c, which fold onto tap (ky, tx - px), channel c.  -1: padding.
// s2d3: the space-to-depth form of a 6 x 6 stride-2 conv over 3 channels (conv_geom.h: svg_s2d3): tap = ty * 3 + tx of the 3 x 3 kernel, ci = (py * 2 + px) * 3 + c
// -> kernel position (2 ty + py, 2 tx + px), channel c of the [6][6][3][N] gradient.
__device__ __forceinline__ int64_t dw_index(int tap, int ci, int co, int Cin_real, int N, int fold_kw, int fold_c, int s2d3 = 0) {
  if (s2d3) {
    const int ty = tap / 3, tx = tap - ty * 3, p = ci / 3, c = ci - p * 3;
    return ((int64_t)(((2 * ty + (p >> 1)) * 6 + 2 * tx + (p & 1)) * 3 + c)) * N + co;
  }
  if (!fold_kw) return ((int64_t)(tap * Cin_real + ci)) * N + co;
  const int ky = tap / (fold_kw + 1), tx = tap - ky * (fold_kw + 1), px = co >> 3, c = co & 7, kx = tx - px;
  if ((unsigned)kx >= (unsigned)fold_kw || c >= fold_c) return -1;
  return ((int64_t)((ky * fold_kw + kx) * Cin_real + ci)) * fold_c + c;
}

static inline int ilog2_exact(int v) {   // -1 if not a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
