// Weight gradient (Conv2DBackpropFilter + BiasAddGrad of tape.gradient, vae/trainer.py:137) on MFMA:
//
//   dW[(t, ci)][co] += sum_m A[pix(m) + (dy_t, dx_t), ci] * dY[m, co]        dbias[co] += sum_m dY[m, co]
//
// The contraction runs over m = (b, oy, ox) (up to 2M rows), so both MFMA operands are needed
// "k-major": they are staged in their natural NHWC row layout ([m][channels], coalesced 16-B
// loads) and transposed for free on the way out of LDS with ds_read_b64_tr_b16 (bf16) or read
// as 32-bit columns (fp32).  m is split across blockIdx.z; partial products are combined with
// fp32 global atomics into the zero-initialised gradient buffer.
#include "common.hip.h"
#include "kernels.h"

__device__ __forceinline__ short4_t lds_tr16_b64(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

template <typename T, int WR, int WC, int CF>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradMulti mg) {
  // several independent problems per launch (tap_gemm.hip has the same scheme): blockIdx.z = (problem, m-split)
  int inst = 0;
#pragma unroll
  for (int i = 1; i < SV_WGRAD_IM2COL_MAX_MULTI; ++i) inst += (i < mg.n && (int)blockIdx.z >= mg.zbase[i]) ? 1 : 0;
  const WgradArgs& g = mg.a[inst];
  const int zi = (int)blockIdx.z - mg.zbase[inst];
  const bool plain = mg.plain[inst] != 0;
  constexpr int EPP = ElemTraits<T>::EPP;
  constexpr int BR = 64 * WR;             // wrows (tap, ci) per block
  constexpr int BNW = 16 * CF * WC;       // output channels per block
  constexpr int MS = 128 / (int)sizeof(T);  // m rows per K-step (64 bf16 / 32 fp32)
  // LDS row pitches: an odd multiple of 32 B (bf16) so that the 8 consecutive m-rows touched by one
  // lane-half of a transposed read start on 8 different 32-B bank groups (see wgrad_tile.hip)
  constexpr int SA = BR * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
  constexpr int SB = BNW * (int)sizeof(T) + (sizeof(T) == 2 && BNW >= 32 ? 32 : 16);
  constexpr int PPRA = BR / EPP, PPRB = BNW / EPP;
  constexpr int APT = (MS * PPRA + 255) / 256;     // A pieces per thread
  constexpr int BPT = (MS * PPRB + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                          // [2][MS][SA]
  char* sB = smem + 2 * MS * SA;            // [2][MS][SB]
  int* sTap = (int*)(sB + 2 * MS * SB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int wrow0 = blockIdx.x * BR, n0 = blockIdx.y * BNW;
  if (wrow0 >= g.Nrows || n0 >= g.N) return;
  if (tid < g.ntaps) {
    sTap[tid * 3 + 0] = g.dy[tid];
    sTap[tid * 3 + 1] = g.dx[tid];
    sTap[tid * 3 + 2] = ((int)g.dy[tid] * g.IW + (int)g.dx[tid]) * g.lda;
  }
  const int mbeg = zi * g.msplit;
  const int mend = min(g.M, mbeg + g.msplit);
  const int nsteps = (mend - mbeg + MS - 1) / MS;
  const T* __restrict__ Ab = (const T*)g.A;
  const T* __restrict__ Yb = (const T*)g.dY;
  const int Ptot = g.Nrows / EPP;
  __syncthreads();

  // per-thread static part of the A pieces (which wrow piece, its tap)
  int a_row[APT], a_col[APT], a_dy[APT], a_dx[APT], a_off[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int q = tid + 256 * i;
    a_row[i] = q / PPRA;
    const int pc = q % PPRA;
    a_col[i] = pc;
    const int wp = wrow0 / EPP + pc;
    if (a_row[i] < MS && wp < Ptot) {
      const int tap = wp >> g.cl2;
      a_dy[i] = sTap[tap * 3];
      a_dx[i] = sTap[tap * 3 + 1];
      a_off[i] = sTap[tap * 3 + 2] + (wp & ((1 << g.cl2) - 1)) * EPP;
    } else {
      a_dy[i] = -(1 << 20); a_dx[i] = 0; a_off[i] = 0;
    }
  }
  uint4 ra[APT], rb[BPT];
  auto load_stage = [&](int step) {
    const int mb = mbeg + step * MS;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int m = mb + a_row[i];
      int b, oy, ox;
        sv_decode_row(m, g.lOY, g.lOX, g.OY, g.OX, b, oy, ox);
      const int iy = oy * g.S + a_dy[i], ix = ox * g.S + a_dx[i];
      const bool ok = m < mend && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
      const int64_t o = (int64_t)((b * g.IH + oy * g.S) * g.IW + ox * g.S) * g.lda + a_off[i];
      ra[i] = ok ? *(const uint4*)(Ab + o) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int q = tid + 256 * i;
      const int row = q / PPRB, pc = q % PPRB;
      const int m = mb + row;
      const bool ok = row < MS && m < mend && (n0 + pc * EPP) < g.ycols;
      rb[i] = ok ? *(const uint4*)(Yb + (int64_t)m * g.ldy + n0 + pc * EPP) : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APT; ++i)
      if (a_row[i] < MS) *(uint4*)(sA + buf * (MS * SA) + a_row[i] * SA + a_col[i] * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int q = tid + 256 * i;
      const int row = q / PPRB, pc = q % PPRB;
      if (row < MS) *(uint4*)(sB + buf * (MS * SB) + row * SB + pc * 16) = rb[i];
    }
  };

  f32x4 acc[4][CF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = g.dbias != nullptr && blockIdx.x == 0;
  constexpr int BG = 256 / BNW;            // row groups for the bias column sums
  const int bcol = tid % BNW, bgrp = tid / BNW;

  if (nsteps > 0) {
    load_stage(0);
    write_stage(0);
  }
  __syncthreads();
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3, lr = lane & 15;
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    const bool more = step + 1 < nsteps;
    if (more) load_stage(step + 1);
    const char* cA = sA + buf * (MS * SA);
    const char* cB = sB + buf * (MS * SB);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < MS / 32; ++kk) {
        // K index k = 8g + 4h + q  <->  m-row 16h + 4g + q (same permutation on both operands)
        const int mrow = kk * 32 + 4 * lg + lq;
        short8_t af[4], bfr[CF];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const char* p = cA + mrow * SA + (wr * 64 + i * 16 + 4 * lp) * 2;
          const short4_t lo = lds_tr16_b64(p), hi = lds_tr16_b64(p + 16 * SA);
          af[i] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < CF; ++j) {
          const char* p = cB + mrow * SB + ((wc * CF + j) * 16 + 4 * lp) * 2;
          const short4_t lo = lds_tr16_b64(p), hi = lds_tr16_b64(p + 16 * SB);
          bfr[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < CF; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                __builtin_bit_cast(bf16x8, bfr[j]), acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int k4 = 0; k4 < MS / 4; ++k4) {
        const int mrow = k4 * 4 + lg;
        float af[4], bfr[CF];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const float*)(cA + mrow * SA + (wr * 64 + i * 16 + lr) * 4);
#pragma unroll
        for (int j = 0; j < CF; ++j) bfr[j] = *(const float*)(cB + mrow * SB + ((wc * CF + j) * 16 + lr) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < CF; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (do_bias) {
      for (int r = bgrp; r < MS; r += BG) bsum += to_f32(*(const T*)(cB + r * SB + bcol * (int)sizeof(T)));
    }
    if (more) write_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: D row = wrow (lane>>4)*4+reg, col = channel lane&15
  const int cshift = g.cl2 + (EPP == 8 ? 3 : 2);   // log2(Cin_pad)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int wrow = wrow0 + wr * 64 + i * 16 + lg * 4 + r;
      if (wrow >= g.Nrows) continue;
      const int tap = wrow >> cshift, ci = wrow & (g.Cin_pad - 1);
      if (ci >= g.Cin_real) continue;
#pragma unroll
      for (int j = 0; j < CF; ++j) {
        const int n = n0 + (wc * CF + j) * 16 + lr;
        if (n < g.N) {
          float* dst = g.dW + ((int64_t)(tap * g.Cin_real + ci)) * g.N + n;
          if (plain) *dst += acc[i][j][r];
          else atomicAdd(dst, acc[i][j][r]);
        }
      }
    }
  }
  if (do_bias) {
    float* red = (float*)smem;   // all LDS reads finished at the loop's last barrier
    red[bgrp * BNW + bcol] = bsum;
    __syncthreads();
    if (tid < BNW) {
      float s = 0.f;
      for (int k = 0; k < BG; ++k) s += red[k * BNW + tid];
      if (n0 + tid < g.N) atomicAdd(g.dbias + n0 + tid, s);
    }
  }
}

template <typename T, int WR, int WC, int CF>
static int launch_wgrad(const WgradArgs* a, int n, hipStream_t st) {
  constexpr int BR = 64 * WR, BNW = 16 * CF * WC, MS = 128 / (int)sizeof(T);
  constexpr int SA = BR * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
  constexpr int SB = BNW * (int)sizeof(T) + (sizeof(T) == 2 && BNW >= 32 ? 32 : 16);
  const size_t lds = 2 * MS * SA + 2 * MS * SB + SV_MAX_TAPS * 3 * sizeof(int);
  static const bool no_plain = getenv("SV_WGRAD_NO_PLAIN") != nullptr;   // A/B knob: always atomics
  WgradMulti m;
  m.n = n;
  int gx = 0, gy = 0, gz = 0;
  for (int i = 0; i < n; ++i) {
    m.a[i] = a[i];
    m.zbase[i] = gz;
    const int z = (a[i].M + a[i].msplit - 1) / a[i].msplit;
    m.plain[i] = z == 1 && !no_plain;
    gx = max(gx, (a[i].Nrows + BR - 1) / BR);
    gy = max(gy, (a[i].N + BNW - 1) / BNW);
    gz += z;
  }
  for (int i = n; i < SV_WGRAD_IM2COL_MAX_MULTI; ++i) { m.zbase[i] = gz; m.plain[i] = 0; }
  dim3 grid(gx, gy, gz), block(256);
  sv_ensure_dynamic_lds((const void*)wgrad_kernel<T, WR, WC, CF>, lds);
  hipLaunchKernelGGL((wgrad_kernel<T, WR, WC, CF>), grid, block, lds, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int svk_wgrad_multi(const WgradArgs* a, int n, int dtype, int cfg, hipStream_t st) {
  if (n < 1 || n > SV_WGRAD_IM2COL_MAX_MULTI) return SV_E_BADARG;
  const int ms = dtype == SV_BF16 ? 64 : 32;
  for (int i = 0; i < n; ++i) {
    if (a[i].ntaps > SV_MAX_TAPS || a[i].msplit <= 0 || a[i].msplit % ms) return SV_E_BADARG;
    if (a[i].ups || a[i].fold_kw) return SV_E_UNSUPPORTED;   // needs the materialised hi-res tensor / cannot fold
  }
  if (dtype == SV_BF16) {
    switch (cfg) {
      case 0: return launch_wgrad<bf16_t, 1, 4, 2>(a, n, st);
      case 1: return launch_wgrad<bf16_t, 2, 2, 2>(a, n, st);
      case 2: return launch_wgrad<bf16_t, 4, 1, 2>(a, n, st);
      case 3: return launch_wgrad<bf16_t, 4, 1, 1>(a, n, st);
    }
  } else if (dtype == SV_F32) {
    switch (cfg) {
      case 0: return launch_wgrad<float, 1, 4, 2>(a, n, st);
      case 1: return launch_wgrad<float, 2, 2, 2>(a, n, st);
      case 2: return launch_wgrad<float, 4, 1, 2>(a, n, st);
      case 3: return launch_wgrad<float, 4, 1, 1>(a, n, st);
    }
  }
  return SV_E_BADARG;
}

int svk_wgrad(const WgradArgs& a, int dtype, int cfg, hipStream_t st) { return svk_wgrad_multi(&a, 1, dtype, cfg, st); }
