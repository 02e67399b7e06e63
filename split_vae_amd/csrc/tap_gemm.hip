// Implicit-GEMM NHWC convolution on the gfx950 matrix cores ("tap GEMM").
//
//   out[m, n] = act( bias[n] + sum_{tap t} sum_{c} A[pix(m) + (dy_t, dx_t), c] * Wt[n][t][c] )
//
// One kernel serves the forward convs (vae/model.py:36-38,:153-156), their data gradients
// (stride-1: flipped taps; stride-2: one launch per output parity class), and the dense layers
// (:41-42,:152) as the 1x1 / H=W=1 case.  K is walked in 16-byte "pieces" (8 bf16 / 4 fp32
// channels of one tap), eight pieces (128 B per row) per K-step:
//   global --(16 B/lane, zero-filled at the SAME-padding border)--> registers --> LDS (XOR-
//   swizzled 128-B rows, 2 buffers) --(ds_read_b128)--> MFMA 16x16x32 bf16 | 4x 16x16x4 f32.
// The next K-step's global loads are issued before the current step's MFMAs (register-staged
// software pipeline, one barrier per K-step).  256 threads = 4 waves, each wave owns WM rows of
// the BM = 4*WM row tile and all BN columns.
#include "common.hip.h"
#include "kernels.h"

template <typename T> struct MmaOp;
template <> struct MmaOp<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaOp<float> {
  // a 16-B piece holds k = 4g..4g+3 for lane group g; MFMA j consumes element j of every lane,
  // i.e. k-slot g <-> k = 4g + j on both operands, so four 16x16x4 MFMAs cover the 16 k's.
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    const float4 af = __builtin_bit_cast(float4, a), bf = __builtin_bit_cast(float4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, c, 0, 0, 0);
  }
};

template <typename T, int BN, int WM>
__global__ __launch_bounds__(256) void tap_gemm_kernel(const TapGemmMulti mg) {
  // up to SV_TAP_MAX_MULTI independent problems per launch (the twin networks' heads and dense layers: each alone
  // fills a fraction of the chip).  blockIdx.z = (problem, split-K slice); problems may differ in every field,
  // the grid is the largest of them and the surplus workgroups of the smaller ones leave at once.
  int inst = 0;
#pragma unroll
  for (int i = 1; i < SV_TAP_MAX_MULTI; ++i) inst += (i < mg.n && (int)blockIdx.z >= mg.zbase[i]) ? 1 : 0;
  const TapGemmArgs& g = mg.a[inst];
  const int zi = (int)blockIdx.z - mg.zbase[inst];
  constexpr int BM = 4 * WM, MF = WM / 16, NF = BN / 16;
  constexpr int AR = BM / 32;                       // A pieces per thread per K-step
  constexpr int BRN = BN >= 32 ? BN / 32 : 1;       // B pieces per thread per K-step
  constexpr int EPP = ElemTraits<T>::EPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + 2 * BM * 128;
  int* sTap = (int*)(sB + 2 * BN * 128);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= g.M || n0 >= ((g.N + BN - 1) / BN) * BN) return;
  if (tid < g.ntaps) {
    sTap[tid * 3 + 0] = g.dy[tid];
    sTap[tid * 3 + 1] = g.dx[tid];
    sTap[tid * 3 + 2] = ((int)g.dy[tid] * g.IW + (int)g.dx[tid]) * g.lda;
  }
  const int pp = tid & 7, r0 = tid >> 3;
  int iy0[AR], ix0[AR], roff[AR];
  {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int m = m0 + r0 + 32 * i;
      if (m < g.M) {
        int b, oy, ox;
        sv_decode_row(m, g.lOY, g.lOX, g.OY, g.OX, b, oy, ox);
        iy0[i] = oy * g.S;
        ix0[i] = ox * g.S;
        roff[i] = ((b * g.IH + iy0[i]) * g.IW + ix0[i]) * g.lda;
      } else {
        iy0[i] = -(1 << 20); ix0[i] = 0; roff[i] = 0;
      }
    }
  }
  const T* __restrict__ Ab = (const T*)g.A;
  const T* __restrict__ Wb = (const T*)g.Wt;
  const int nk_all = (g.P + 7) >> 3;
  const int ks_begin = (int)(((int64_t)nk_all * zi) / g.splitk);
  const int ks_end = (int)(((int64_t)nk_all * (zi + 1)) / g.splitk);
  __syncthreads();   // tap table visible

  uint4 ra[AR], rb[BRN];
  auto load_stage = [&](int ks) {
    const int p = ks * 8 + pp;
    const bool pv = p < g.P;
    const int tap = pv ? (p >> g.cl2) : 0;
    const int ci0 = (p & ((1 << g.cl2) - 1)) * EPP;
    const int tdy = sTap[tap * 3], tdx = sTap[tap * 3 + 1], toff = sTap[tap * 3 + 2] + ci0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int iy = iy0[i] + tdy, ix = ix0[i] + tdx;
      const bool ok = pv && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
      ra[i] = ok ? *(const uint4*)(Ab + (int64_t)(roff[i] + toff)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BRN; ++i) {
      const int n = r0 + 32 * i;
      const bool ok = pv && n < BN;
      rb[i] = ok ? *(const uint4*)(Wb + (int64_t)(n0 + n) * g.Ktot + (int64_t)p * EPP) : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = r0 + 32 * i;
      *(uint4*)(sA + buf * (BM * 128) + r * 128 + ((pp ^ (r & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BRN; ++i) {
      const int n = r0 + 32 * i;
      if (n < BN) *(uint4*)(sB + buf * (BN * 128) + n * 128 + ((pp ^ (n & 7)) << 4)) = rb[i];
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (ks_begin < ks_end) {
    load_stage(ks_begin);
    write_stage(0);
  }
  __syncthreads();
  const int lr = lane & 15, lg = lane >> 4;
  for (int ks = ks_begin; ks < ks_end; ++ks) {
    const int buf = (ks - ks_begin) & 1;
    const bool more = ks + 1 < ks_end;
    if (more) load_stage(ks + 1);
    const char* cA = sA + buf * (BM * 128);
    const char* cB = sB + buf * (BN * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[MF], bfr[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int r = wave * WM + i * 16 + lr;
        af[i] = *(const uint4*)(cA + r * 128 + (((kk * 4 + lg) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int n = j * 16 + lr;
        bfr[j] = *(const uint4*)(cB + n * 128 + (((kk * 4 + lg) ^ (n & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) MmaOp<T>::run(bfr[j], af[i], acc[i][j]);   // D = W x rows: a lane gets 4 channels of one row
    }
    if (more) write_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  Operands swapped: D rows = channels, cols = GEMM rows, so a lane holds channels
  // n0 + j*16 + (lane>>4)*4 + {0..3} of row lane&15 of each fragment (see tile_conv.hip).
  // columns stored: the real channels, and -- for a low-precision output whose channel count is not a multiple of 8 (SPAIR's
  // 100-channel z3) -- the zero pad channels up to the tensor's 8-channel pitch (weight rows >= N are zero, the bias is skipped)
  const int Nst = (!g.out_f32 && g.splitk == 1 && (g.N & 7)) ? min(g.ldo, (g.N + 7) & ~7) : g.N;
  const int ncols = min(BN, Nst - n0);
  {
    // transposed through LDS into row-contiguous 16-B (8-B for narrow fp32 rows) stores; split-K
    // partial sums take the same route so that their fp32 atomics are issued row-contiguously
    const int oesz = g.out_f32 ? 4 : (int)sizeof(T);
    const int rowb = ncols * oesz;
    if (!(rowb & 7)) {
      const int srow = ((rowb + 15) & ~15) + 16;
      char* sC = smem;                         // all LDS reads finished at the loop's last barrier
      float bv[NF][4];
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int nl = j * 16 + lg * 4 + e;
          bv[j][e] = (g.bias && n0 + nl < g.N && zi == 0) ? g.bias[n0 + nl] : 0.f;   // split-K: the first K slice carries the bias
        }
      const bool relu = g.act == SV_ACT_RELU;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int rl = wave * WM + i * 16 + lr;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int nl = j * 16 + lg * 4;
          if (nl >= ncols) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[i][j][e] + bv[j][e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
          }
          if (g.out_f32) *(float4*)(sC + rl * srow + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
          else if constexpr (sizeof(T) == 2) {
            T pk[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
            *(uint2*)(sC + rl * srow + nl * 2) = *(uint2*)pk;
          } else *(float4*)(sC + rl * srow + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      __syncthreads();
      if (g.splitk > 1) {
        // fp32 partial sums (no activation / mask: checked at launch; the bias rides on K slice 0): one dword per lane,
        // consecutive lanes on consecutive channels, so an atomic instruction covers whole 128-B lines
        const float rinv = 1.0f / (float)ncols;
        for (int q = tid; q < BM * ncols; q += 256) {
          const int rl = (int)(((float)q + 0.5f) * rinv), n = q - rl * ncols;
          const int m = m0 + rl;
          if (m >= g.M) continue;
          int b, oy, ox;
        sv_decode_row(m, g.lOY, g.lOX, g.OY, g.OX, b, oy, ox);
          const int64_t pix = ((int64_t)b * g.OHF + oy * g.OS + g.ooy) * g.OWF + ox * g.OS + g.oox;
          atomicAdd((float*)g.out + pix * g.ldo + n0 + n, *(const float*)(sC + rl * srow + n * 4));
        }
        return;
      }
      const int psz = (rowb & 15) ? 8 : 16, ppr_o = rowb / psz;
      for (int q = tid; q < BM * ppr_o; q += 256) {
        const int rl = q / ppr_o, c = q - rl * ppr_o;
        const int m = m0 + rl;
        if (m >= g.M) continue;
        int b, oy, ox;
        sv_decode_row(m, g.lOY, g.lOX, g.OY, g.OX, b, oy, ox);
        const int64_t pix = ((int64_t)b * g.OHF + oy * g.OS + g.ooy) * g.OWF + ox * g.OS + g.oox;
        const int64_t ob = (pix * g.ldo + n0) * oesz + c * psz;
        if (psz == 16) {
          uint4 v = *(const uint4*)(sC + rl * srow + c * 16);
          if (g.mask) {
            const uint4 mv = *(const uint4*)((const char*)g.mask + ob);
            T ve[EPP], me[EPP];
            *(uint4*)ve = v; *(uint4*)me = mv;
#pragma unroll
            for (int e = 0; e < EPP; ++e) ve[e] = to_f32(me[e]) > 0.f ? ve[e] : from_f32<T>(0.f);
            v = *(uint4*)ve;
          }
          if (g.accum && g.out_f32) {                   // one workgroup owns the tile: a plain add is race-free and order-free
            const float4 o4 = *(const float4*)((const char*)g.out + ob);
            float4 v4 = *(float4*)&v;
            v4.x += o4.x; v4.y += o4.y; v4.z += o4.z; v4.w += o4.w;
            v = *(uint4*)&v4;
          }
          *(uint4*)((char*)g.out + ob) = v;
        } else {
          uint2 v2 = *(const uint2*)(sC + rl * srow + c * 8);
          if (g.accum && g.out_f32) {
            const float2 o2 = *(const float2*)((const char*)g.out + ob);
            float2 f2 = *(float2*)&v2;
            f2.x += o2.x; f2.y += o2.y;
            v2 = *(uint2*)&f2;
          }
          *(uint2*)((char*)g.out + ob) = v2;
        }
      }
      return;
    }
  }
  // odd row widths: straight from the registers
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int m = m0 + wave * WM + i * 16 + lr;
    if (m >= g.M) continue;
    int b, oy, ox;
        sv_decode_row(m, g.lOY, g.lOX, g.OY, g.OX, b, oy, ox);
    const int64_t pix = ((int64_t)b * g.OHF + oy * g.OS + g.ooy) * g.OWF + ox * g.OS + g.oox;
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + j * 16 + lg * 4 + e;
        if (n >= g.N) continue;
        float v = acc[i][j][e];
        const int64_t o = pix * g.ldo + n;
        if (g.splitk > 1) {
          atomicAdd((float*)g.out + o, (g.bias && zi == 0) ? v + g.bias[n] : v);
          continue;
        }
        if (g.bias) v += g.bias[n];
        if (g.act == SV_ACT_RELU) v = fmaxf(v, 0.f);
        if (g.mask) v = to_f32(((const T*)g.mask)[o]) > 0.f ? v : 0.f;
        if (g.out_f32) ((float*)g.out)[o] = g.accum ? ((float*)g.out)[o] + v : v;
        else ((T*)g.out)[o] = from_f32<T>(v);
      }
  }
}

template <typename T, int BN, int WM>
static int launch_tap(const TapGemmArgs* a, int n, hipStream_t st) {
  constexpr int BM = 4 * WM;
  TapGemmMulti m;
  m.n = n;
  int gx = 0, gy = 0, gz = 0;
  size_t lds = 2 * BM * 128 + 2 * BN * 128 + SV_MAX_TAPS * 3 * sizeof(int);
  for (int i = 0; i < n; ++i) {
    m.a[i] = a[i];
    m.zbase[i] = gz;
    gx = max(gx, (a[i].M + BM - 1) / BM);
    gy = max(gy, round_up(a[i].N, BN) / BN);
    gz += a[i].splitk;
    const size_t epi = (size_t)BM * (((BN * (a[i].out_f32 ? 4 : sizeof(T)) + 15) & ~(size_t)15) + 16);   // epilogue transpose tile
    if (lds < epi) lds = epi;
  }
  for (int i = n; i < SV_TAP_MAX_MULTI; ++i) m.zbase[i] = gz;
  dim3 grid(gx, gy, gz), block(256);
  sv_ensure_dynamic_lds((const void*)tap_gemm_kernel<T, BN, WM>, lds);
  hipLaunchKernelGGL((tap_gemm_kernel<T, BN, WM>), grid, block, lds, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int svk_tap_gemm_multi(const TapGemmArgs* a, int n, int dtype, int cfg, hipStream_t st) {
  if (n < 1 || n > SV_TAP_MAX_MULTI) return SV_E_BADARG;
  for (int i = 0; i < n; ++i) {
    if (a[i].ntaps > SV_MAX_TAPS || a[i].splitk < 1) return SV_E_BADARG;
    if (a[i].splitk > 1 && (!a[i].out_f32 || a[i].act != SV_ACT_NONE || a[i].mask)) return SV_E_BADARG;
  }
  if (dtype == SV_BF16) {
    switch (cfg) {
      case 0: return launch_tap<bf16_t, 128, 32>(a, n, st);
      case 1: return launch_tap<bf16_t, 64, 32>(a, n, st);
      case 2: return launch_tap<bf16_t, 32, 64>(a, n, st);
      case 3: return launch_tap<bf16_t, 16, 64>(a, n, st);
      case 4: return launch_tap<bf16_t, 32, 16>(a, n, st);   // 64 x 32: skinny split-K problems (heads, d1 dgrad)
    }
  } else if (dtype == SV_F32) {
    switch (cfg) {
      case 0: return launch_tap<float, 128, 32>(a, n, st);
      case 1: return launch_tap<float, 64, 32>(a, n, st);
      case 2: return launch_tap<float, 32, 64>(a, n, st);
      case 3: return launch_tap<float, 16, 64>(a, n, st);
      case 4: return launch_tap<float, 32, 16>(a, n, st);
    }
  }
  return SV_E_BADARG;
}

int svk_tap_gemm(const TapGemmArgs& a, int dtype, int cfg, hipStream_t st) { return svk_tap_gemm_multi(&a, 1, dtype, cfg, st); }
