// Border correction of the POLYPHASE decoder head (conv_geom.h: svg_poly).
//
// The polyphase conv over the edge-clamped low-res tensor equals the 6x6 conv over the REPLICATE-padded 2x upsample; the
// reference (vae/model.py:163-167: UpSampling2D(bilinear) -> Conv2D(padding='same')) ZERO-pads it.  The difference is the
// contribution of the taps that leave the image, which only the hi-res output rows 0, 1, 2H-3, 2H-2, 2H-1 (H = 2h) and
// the same five columns have.  Split disjointly:
//   rows:    D(r, c) = sum_{ky in Ey(r)} sum_kx w[ky,kx] . Rrow[c + kx - 2]      Rrow = the first / last upsampled ROW,
//                                                                                 replicate-extended in x (covers the corners)
//   columns: D(r, c) = sum_{kx in Ex(c)} sum_ky w[ky,kx] . Zcol[r + ky - 2]      Zcol = the first / last upsampled COLUMN,
//                                                                                 ZERO outside the image (row-in taps only)
// i.e. ten 1-D 6-tap convs along 2h- / 2w-pixel lines per image, K = 6 x 32, on MFMA with the class weights -(sum over the
// excluded ky or kx) prepared by prep_poly (conv_api.hip).  One workgroup per image: the four lines are interpolated into
// LDS (same lerp arithmetic and bf16 rounding as the fused-upsample staging); a wave takes whole classes (its six weight
// fragments stay in registers for the line's fragments).  Two delivery modes:
//   fixbuf != null: the terms go to a workspace laid out like the conv's output (plain stores, no ordering constraint: the kernel
//                   runs BEFORE the polyphase conv, whose epilogue adds them to its border pixels -- tile_conv.hip, TileConvArgs::fix)
//   fixbuf == null: they are added to out6 with fp32 atomics AFTER the conv (a corner pixel receives a row and a column term);
//                   the path of callers without a workspace (sv_conv2d_nhwc_fwd), ~10x slower
#include "common.hip.h"
#include "kernels.h"
#include "fix_mma.hip.h"
#include <stdlib.h>

namespace {

struct PolyFixMulti { const void* x[2]; const void* wfix[2]; float* out6[2]; float* fixbuf[2]; };   // blockIdx.y: the twin networks

// one pixel of an upsampled edge line: hi coordinate u (clamped by the caller) of a line of n low-res pixels -> the two low-res indices and the weight of the second
__device__ __forceinline__ void line_src(int u, int n, int& i0, int& i1, float& f) {
  const int m = u >> 1;
  i0 = (u & 1) ? m : max(m - 1, 0); i1 = (u & 1) ? min(m + 1, n - 1) : m;
  f = (u & 1) ? 0.25f : 0.75f;                             // weight of the second sample (common.hip.h: lerp2)
}

template <typename T>
__global__ __launch_bounds__(256) void poly_fix_kernel(const PolyFixMulti mg, int h, int w, int lda, int Cout, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPP = ElemTraits<T>::EPP, NPC = 32 / EPP, PSB = 32 * (int)sizeof(T) + (sizeof(T) == 4 ? 16 : 0), NG = 32 / FixMma<T>::CPG;   // pieces / bytes per line pixel (fp32: +16 B, the
                                                                                     // 16 pixels of a fragment read on different banks), MFMA groups per tap
  const T* __restrict__ x = (const T*)mg.x[blockIdx.y];
  const T* __restrict__ wfix = (const T*)mg.wfix[blockIdx.y];
  float* __restrict__ out6 = mg.out6[blockIdx.y];
  float* __restrict__ fixbuf = mg.fixbuf[blockIdx.y];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction: as an SGPR the class loop below compiles to scalar branches.  (With a per-lane `wave` the
  // fragment predicates became EXEC-masked regions around the MFMAs and their AGPR moves, and the kernel was not run-to-run
  // reproducible once two workgroups shared a CU.)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int H2 = 2 * h, W2 = 2 * w;
  const int L = H2 > W2 ? H2 : W2, LW = L + 5;            // line index li = hi coordinate + 2, hi coordinate in -2 .. 2n+2
  // this wave's first class: its weight fragments are in flight while the lines are built
  uint4 wv[6][NG];
  auto load_w = [&](int cls) {
    const T* wp = wfix + ((int64_t)cls * 6 * 16 + lr) * 32 + lg * EPP;
#pragma unroll
    for (int tap = 0; tap < 6; ++tap)
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) wv[tap][gq] = *(const uint4*)(wp + tap * 16 * 32 + gq * FixMma<T>::CPG);
  };
  load_w(wave);
  const T* xb = x + (int64_t)b * h * w * lda;
  // ---- the four lines: [line][li][32 channels] of T
  for (int it = tid; it < 4 * LW * NPC && !(SV_DBG(dbg) & 1); it += 256) {
    const int ch = it % NPC, li = (it / NPC) % LW, line = (it / NPC) / LW;
    const bool is_row = line < 2;
    const int n = is_row ? w : h;                          // low-res extent along the line
    int u = li - 2;                                        // hi coordinate
    uint4 v = make_uint4(0, 0, 0, 0);
    if (li < 2 * n + 5 && (is_row || (u >= 0 && u < 2 * n))) {
      u = min(max(u, 0), 2 * n - 1);
      const int m = u >> 1;
      const int i0 = (u & 1) ? m : max(m - 1, 0), i1 = (u & 1) ? min(m + 1, n - 1) : m;
      const float f = (u & 1) ? 0.25f : 0.75f;             // weight of the second sample (common.hip.h: lerp2)
      const int64_t o0 = is_row ? ((int64_t)(line == 0 ? 0 : h - 1) * w + i0) * lda : ((int64_t)i0 * w + (line == 2 ? 0 : w - 1)) * lda;
      const int64_t o1 = is_row ? ((int64_t)(line == 0 ? 0 : h - 1) * w + i1) * lda : ((int64_t)i1 * w + (line == 2 ? 0 : w - 1)) * lda;
      const uint4 a0 = *(const uint4*)(xb + o0 + ch * EPP), a1 = *(const uint4*)(xb + o1 + ch * EPP);
      f32x2 p0[Piece<T>::NP], p1[Piece<T>::NP], r[Piece<T>::NP];
      Piece<T>::unpack(a0, p0); Piece<T>::unpack(a1, p1);
#pragma unroll
      for (int e = 0; e < Piece<T>::NP; ++e) r[e] = lerp2(p0[e], p1[e], f);
      v = Piece<T>::pack(r);
    }
    *(uint4*)(smem + (line * LW + li) * PSB + ch * 16) = v;
  }
  __syncthreads();
  // ---- classes 0..4: hi-res rows 0, 1, 2h-3, 2h-2, 2h-1 (tap = kx); 5..9: the columns (tap = ky)
  for (int cls = wave; cls < 10 && !(SV_DBG(dbg) & 2); cls += 4) {
    const bool rows = cls < 5;
    const int c5 = cls % 5, line = (rows ? 0 : 2) + (c5 >= 2 ? 1 : 0);
    const int n2 = rows ? W2 : H2, nf = n2 >> 4;           // pixels / fragments along the line
    const int m2 = rows ? H2 : W2;
    const int edge = c5 == 0 ? 0 : c5 == 1 ? 1 : m2 - 5 + c5;                   // 0, 1, m-3, m-2, m-1
    uint4 wc[6][NG];
#pragma unroll
    for (int tap = 0; tap < 6; ++tap)
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) wc[tap][gq] = wv[tap][gq];
    if (cls + 4 < 10) load_w(cls + 4);                     // next class of this wave
    for (int f0 = 0; f0 < nf; f0 += 4) {
      f32x4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 6; ++tap)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (f0 + q >= nf) continue;
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) {
            const uint4 pv = *(const uint4*)(smem + (line * LW + 16 * (f0 + q) + lr + tap) * PSB + gq * (FixMma<T>::CPG * (int)sizeof(T)) + lg * 16);
            FixMma<T>::run(wc[tap][gq], pv, acc[q]);
          }
        }
      // lane (lr, lg): channels lg*4 .. lg*4+3 of line pixel 16 f + lr
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (f0 + q >= nf) continue;
        const int pos = 16 * (f0 + q) + lr;
        if (SV_DBG(dbg) & 4) continue;
        if (fixbuf) {
          // workspace in the conv's OUTPUT layout, so that its epilogue adds whole 16-B pieces (tile_conv.hip):
          //   rows:    [b][class 0..4][hi col][Cout]                          = an out6 row per class
          //   columns: [b][hi row][group 0..2][2 pixels][Cout], group = the low-res column 0, w-2, w-1 whose pixel pair holds the
          //            border column(s); classes 0..4 = (group, pixel) (0,0) (0,1) (1,1) (2,0) (2,1); (1,0) is no border column: zero
          float* fb = fixbuf + (int64_t)b * (5 * W2 + 6 * H2) * Cout;
          float* p = rows ? fb + ((int64_t)c5 * W2 + pos) * Cout
                          : fb + (int64_t)5 * W2 * Cout + (((int64_t)pos * 3 + (c5 < 2 ? 0 : c5 == 2 ? 1 : 2)) * 2 + ((c5 == 0 || c5 == 3) ? 0 : 1)) * Cout;
          // Cout is even (svg_packx): 8-byte stores
          if (lg * 4 < Cout) *(f32x2*)(p + lg * 4) = (f32x2){acc[q][0], acc[q][1]};
          if (lg * 4 + 2 < Cout) *(f32x2*)(p + lg * 4 + 2) = (f32x2){acc[q][2], acc[q][3]};
          if (!rows && c5 == 2) {
            if (lg * 4 < Cout) *(f32x2*)(p + lg * 4 - Cout) = (f32x2){0.f, 0.f};
            if (lg * 4 + 2 < Cout) *(f32x2*)(p + lg * 4 + 2 - Cout) = (f32x2){0.f, 0.f};
          }
        } else {
          const int r = rows ? edge : pos, c = rows ? pos : edge;
          float* op = out6 + (((int64_t)b * H2 + r) * W2 + c) * Cout;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (lg * 4 + e < Cout) atomicAdd(op + lg * 4 + e, acc[q][e]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same border terms for the PER-CLASS polyphase form (conv_geom.h: svg_polyc; d4 / d3): any kernel size K in {4, 6} (K - 1 border classes per
// direction: hi-res rows 0 .. pad-1 and 2h-nb .. 2h-1, nb = K-1-pad), Cin a multiple of the MFMA group, Cout a multiple of 16.  One workgroup per image:
// the four upsampled edge lines go to LDS, then (class, 16 output channels, 16 line pixels) units are dealt to the waves; a unit is K taps x Cin / CPG
// operand pairs, the class weights (-sum over the excluded taps, conv_api.hip prep poly == 4: [2(K-1)][K][Cout][Cin]) come straight from L2.
//   fixrow[b][c][X][co]  (c < K-1: hi-res row class, X < 2w)        fixcol[b][Y][c][co]  (Y < 2h, c: hi-res column class)
// added by the conv's epilogue before the activation (tile_conv.hip).
struct PolycFixMulti { const void* x[2]; const void* wfix[2]; float* frow[2]; float* fcol[2]; };

// A workgroup owns ONE border class and a group of images: every wave keeps the class weights of its 16 output channels in registers (K taps x NGRP operand
// pieces) for all of them; the class's edge line of NB images at a time goes to LDS (one barrier pair per NB images), a wave then runs its pixel fragments
// of each image: K x NGRP operand pairs per fragment.  (The first version -- one workgroup per image, weights re-read from L2 for every fragment -- took
// 122 us per launch at 2 x 512 images; the work is 4 GFLOP.)
template <typename T, int K, int NGRP, int NCF, int NB>
__global__ __launch_bounds__(256) void polyc_fix_kernel(const PolycFixMulti mg, int B, int h, int w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPP = ElemTraits<T>::EPP, CPG = FixMma<T>::CPG, CIN = NGRP * CPG, COUT = NCF * 16, NPC = CIN / EPP;
  constexpr int PSB = CIN * (int)sizeof(T) + 16, PAD = (K - 1) / 2, NC = K - 1;   // +16: the pixels of a fragment on different banks
  static_assert(4 % NCF == 0, "a wave keeps one channel fragment");
  const T* __restrict__ x = (const T*)mg.x[blockIdx.z];
  const T* __restrict__ wfix = (const T*)mg.wfix[blockIdx.z];
  float* __restrict__ frow = mg.frow[blockIdx.z];
  float* __restrict__ fcol = mg.fcol[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int cls = blockIdx.x, c = cls % NC;
  const bool rows = cls < NC;
  const int H2 = 2 * h, W2 = 2 * w, L = H2 > W2 ? H2 : W2, LW = L + K - 1;
  const int npos = rows ? W2 : H2, n = rows ? w : h, line = (rows ? 0 : 2) + (c >= PAD ? 1 : 0);
  const int cf = wave % NCF;
  uint4 wv[K][NGRP];
  {
    const T* wp = wfix + (((int64_t)cls * K) * COUT + cf * 16 + lr) * CIN + lg * EPP;
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int gq = 0; gq < NGRP; ++gq) wv[tap][gq] = *(const uint4*)(wp + (int64_t)tap * COUT * CIN + gq * CPG);
  }
  const int per = (B + (int)gridDim.y - 1) / (int)gridDim.y, b_lo = (int)blockIdx.y * per, b_hi = min(B, b_lo + per);
  for (int b0 = b_lo; b0 < b_hi; b0 += NB) {
    __syncthreads();                     // the previous batch is consumed
    for (int it = tid; it < NB * LW * NPC; it += 256) {
      const int ch = it % NPC, li = (it / NPC) % LW, ib = it / (NPC * LW), b = b0 + ib;
      int u = li - PAD;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b < b_hi && li < 2 * n + K - 1 && (rows || (u >= 0 && u < 2 * n))) {
        u = min(max(u, 0), 2 * n - 1);
        int i0, i1;
        float f;
        line_src(u, n, i0, i1, f);
        const T* xb = x + (int64_t)b * h * w * CIN;
        const int64_t o0 = rows ? ((int64_t)(line == 0 ? 0 : h - 1) * w + i0) * CIN : ((int64_t)i0 * w + (line == 2 ? 0 : w - 1)) * CIN;
        const int64_t o1 = rows ? ((int64_t)(line == 0 ? 0 : h - 1) * w + i1) * CIN : ((int64_t)i1 * w + (line == 2 ? 0 : w - 1)) * CIN;
        const uint4 a0 = *(const uint4*)(xb + o0 + ch * EPP), a1 = *(const uint4*)(xb + o1 + ch * EPP);
        f32x2 p0[Piece<T>::NP], p1[Piece<T>::NP], r[Piece<T>::NP];
        Piece<T>::unpack(a0, p0); Piece<T>::unpack(a1, p1);
#pragma unroll
        for (int e = 0; e < Piece<T>::NP; ++e) r[e] = lerp2(p0[e], p1[e], f);
        v = Piece<T>::pack(r);
      }
      *(uint4*)(smem + (ib * LW + li) * PSB + ch * 16) = v;
    }
    __syncthreads();
    const int npf = npos >> 4;
    for (int u = wave / NCF; u < NB * npf; u += 4 / NCF) {       // (image of the batch, pixel fragment)
      const int ib = u / npf, pf = u - ib * npf, b = b0 + ib;
      if (b >= b_hi) break;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* lp = smem + (ib * LW + 16 * pf + lr) * PSB + lg * 16;
#pragma unroll
      for (int tap = 0; tap < K; ++tap)
#pragma unroll
        for (int gq = 0; gq < NGRP; ++gq) FixMma<T>::run(wv[tap][gq], *(const uint4*)(lp + tap * PSB + gq * (CPG * (int)sizeof(T))), acc);   // D rows = output channels 4 * lg .., columns = line pixels
      const int pos = 16 * pf + lr, co = cf * 16 + lg * 4;
      float* p = rows ? frow + (((int64_t)b * NC + c) * W2 + pos) * COUT + co : fcol + (((int64_t)b * H2 + pos) * NC + c) * COUT + co;
      *(float4*)p = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

template <typename T, int K, int NGRP, int NCF>
static int launch_polyc_fix(const PolycFixMulti& m, int n, int B, int h, int w, hipStream_t st) {
  constexpr int NB = 4;
  const int LW = 2 * (h > w ? h : w) + K - 1;
  const size_t lds = (size_t)NB * LW * (NGRP * FixMma<T>::CPG * sizeof(T) + 16);
  if (lds > 150 * 1024) return SV_E_UNSUPPORTED;
  int groups = (B + 15) / 16;                               // ~16 images per workgroup: four batches of NB
  // small shards (config 4: 64 images per network): 16 images per workgroup left 80 workgroups on 256 CUs (53 us of the 64-image fp32 step for 0.5 GFLOP);
  // one batch of NB images per workgroup until the launch holds ~2 workgroups per CU (the class weights are 24-48 registers per lane: cheap to re-load)
  const int want = (512 + 2 * (K - 1) * n - 1) / (2 * (K - 1) * n), most = (B + NB - 1) / NB;
  if (groups < want) groups = want < most ? want : most;
  if (groups < 1) groups = 1;
  sv_ensure_dynamic_lds((const void*)polyc_fix_kernel<T, K, NGRP, NCF, NB>, lds);
  hipLaunchKernelGGL((polyc_fix_kernel<T, K, NGRP, NCF, NB>), dim3(2 * (K - 1), groups, n), dim3(256), lds, st, m, B, h, w);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // namespace

int svk_polyc_fix_multi(int n, const void* const* x_lo, const void* const* wfix, float* const* fixrow, float* const* fixcol, int B, int h, int w,
                        int Cin, int Cout, int K, int dtype, hipStream_t st) {
  if (n < 1 || n > 2 || B < 1 || h < 8 || w < 8 || (h & 7) || (w & 7) || (Cout & 15) || (K != 4 && K != 6)) return SV_E_BADARG;
  PolycFixMulti m;
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    m.x[i] = x_lo[k]; m.wfix[i] = wfix[k]; m.frow[i] = fixrow[k]; m.fcol[i] = fixcol[k];
    if (!m.x[i] || !m.wfix[i] || !m.frow[i] || !m.fcol[i]) return SV_E_BADARG;
  }
  // instantiations: the layers of the model (d4: 6 x 6, 64 -> 32; d3: 4 x 4, 128 -> 64) and the 32-channel variant of d4
  if (dtype == SV_F32) {
    if (K == 6 && Cin == 64 && Cout == 32) return launch_polyc_fix<float, 6, 4, 2>(m, n, B, h, w, st);
    if (K == 6 && Cin == 32 && Cout == 32) return launch_polyc_fix<float, 6, 2, 2>(m, n, B, h, w, st);
    if (K == 4 && Cin == 128 && Cout == 64) return launch_polyc_fix<float, 4, 8, 4>(m, n, B, h, w, st);
  } else {
    if (K == 6 && Cin == 64 && Cout == 32) return launch_polyc_fix<bf16_t, 6, 2, 2>(m, n, B, h, w, st);
    if (K == 6 && Cin == 32 && Cout == 32) return launch_polyc_fix<bf16_t, 6, 1, 2>(m, n, B, h, w, st);
    if (K == 4 && Cin == 128 && Cout == 64) return launch_polyc_fix<bf16_t, 4, 4, 4>(m, n, B, h, w, st);
  }
  return SV_E_UNSUPPORTED;
}

int64_t svk_poly_fix_ws_bytes(int B, int h, int w) { return (int64_t)B * (5 * 2 * w + 6 * 2 * h) * 8 * 4; }   // sized for Cout <= 8

int svk_poly_fix_multi(int n, const void* const* x_lo, const void* const* wfix, float* const* out6, float* const* fixbuf, int B,
                       int h, int w, int lda, int Cout, hipStream_t st, int dtype) {
  if (n < 1 || n > 2 || B < 1 || h < 8 || w < 8 || (h & 7) || (w & 7) || Cout < 2 || Cout > 8 || (Cout & 1) || lda < 32) return SV_E_BADARG;
  const int LW = 2 * (h > w ? h : w) + 5;
  const size_t lds = (size_t)4 * LW * (dtype == SV_BF16 ? 64 : 144);
  if (lds > 150 * 1024) return SV_E_UNSUPPORTED;
  PolyFixMulti m;
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    m.x[i] = x_lo[k]; m.wfix[i] = wfix[k];
    m.out6[i] = out6 ? out6[k] : nullptr; m.fixbuf[i] = fixbuf ? fixbuf[k] : nullptr;
    if (!m.out6[i] && !m.fixbuf[i]) return SV_E_BADARG;
  }
  static const int dbg = SV_DBG(getenv("SV_PF_DBG") ? atoi(getenv("SV_PF_DBG")) : 0);   // ablation: 1 skip the lines, 2 skip the classes, 4 skip the stores
  if (dtype == SV_BF16) {
    sv_ensure_dynamic_lds((const void*)poly_fix_kernel<bf16_t>, lds);
    hipLaunchKernelGGL(poly_fix_kernel<bf16_t>, dim3(B, n), dim3(256), lds, st, m, h, w, lda, Cout, dbg);
  } else {
    sv_ensure_dynamic_lds((const void*)poly_fix_kernel<float>, lds);
    hipLaunchKernelGGL(poly_fix_kernel<float>, dim3(B, n), dim3(256), lds, st, m, h, w, lda, Cout, dbg);
  }
  SV_LAUNCH_CHECK();
  return SV_OK;
}
