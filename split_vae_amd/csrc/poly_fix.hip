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
#include <stdlib.h>

namespace {

struct PolyFixMulti { const bf16_t* x[2]; const bf16_t* wfix[2]; float* out6[2]; float* fixbuf[2]; };   // blockIdx.y: the twin networks

__global__ __launch_bounds__(256) void poly_fix_kernel(const PolyFixMulti mg, int h, int w, int lda, int Cout, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16_t* __restrict__ x = mg.x[blockIdx.y];
  const bf16_t* __restrict__ wfix = mg.wfix[blockIdx.y];
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
  uint4 wv[6];
  auto load_w = [&](int cls) {
    const bf16_t* wp = wfix + ((int64_t)cls * 6 * 16 + lr) * 32 + lg * 8;
#pragma unroll
    for (int tap = 0; tap < 6; ++tap) wv[tap] = *(const uint4*)(wp + tap * 16 * 32);
  };
  load_w(wave);
  const bf16_t* xb = x + (int64_t)b * h * w * lda;
  // ---- the four lines: [line][li][32 channels] bf16
  for (int it = tid; it < 4 * LW * 4 && !(SV_DBG(dbg) & 1); it += 256) {
    const int ch = it & 3, li = (it >> 2) % LW, line = (it >> 2) / LW;
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
      const uint4 a0 = *(const uint4*)(xb + o0 + ch * 8), a1 = *(const uint4*)(xb + o1 + ch * 8);
      f32x2 p0[4], p1[4], r[4];
      Piece<bf16_t>::unpack(a0, p0); Piece<bf16_t>::unpack(a1, p1);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = lerp2(p0[e], p1[e], f);
      v = Piece<bf16_t>::pack(r);
    }
    *(uint4*)(smem + ((line * LW + li) * 4 + ch) * 16) = v;
  }
  __syncthreads();
  // ---- classes 0..4: hi-res rows 0, 1, 2h-3, 2h-2, 2h-1 (tap = kx); 5..9: the columns (tap = ky)
  for (int cls = wave; cls < 10 && !(SV_DBG(dbg) & 2); cls += 4) {
    const bool rows = cls < 5;
    const int c5 = cls % 5, line = (rows ? 0 : 2) + (c5 >= 2 ? 1 : 0);
    const int n2 = rows ? W2 : H2, nf = n2 >> 4;           // pixels / fragments along the line
    const int m2 = rows ? H2 : W2;
    const int edge = c5 == 0 ? 0 : c5 == 1 ? 1 : m2 - 5 + c5;                   // 0, 1, m-3, m-2, m-1
    uint4 wc[6];
#pragma unroll
    for (int tap = 0; tap < 6; ++tap) wc[tap] = wv[tap];
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
          const uint4 pv = *(const uint4*)(smem + (line * LW + 16 * (f0 + q) + lr + tap) * 64 + lg * 16);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wc[tap]), __builtin_bit_cast(bf16x8, pv), acc[q], 0, 0, 0);
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

}  // namespace

int64_t svk_poly_fix_ws_bytes(int B, int h, int w) { return (int64_t)B * (5 * 2 * w + 6 * 2 * h) * 8 * 4; }   // sized for Cout <= 8

int svk_poly_fix_multi(int n, const void* const* x_lo, const void* const* wfix, float* const* out6, float* const* fixbuf, int B,
                       int h, int w, int lda, int Cout, hipStream_t st) {
  if (n < 1 || n > 2 || B < 1 || h < 8 || w < 8 || (h & 7) || (w & 7) || Cout < 2 || Cout > 8 || (Cout & 1) || lda < 32) return SV_E_BADARG;
  const int LW = 2 * (h > w ? h : w) + 5;
  const size_t lds = (size_t)4 * LW * 64;
  if (lds > 64 * 1024) return SV_E_UNSUPPORTED;
  PolyFixMulti m;
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    m.x[i] = (const bf16_t*)x_lo[k]; m.wfix[i] = (const bf16_t*)wfix[k];
    m.out6[i] = out6 ? out6[k] : nullptr; m.fixbuf[i] = fixbuf ? fixbuf[k] : nullptr;
    if (!m.out6[i] && !m.fixbuf[i]) return SV_E_BADARG;
  }
  static const int dbg = SV_DBG(getenv("SV_PF_DBG") ? atoi(getenv("SV_PF_DBG")) : 0);   // ablation: 1 skip the lines, 2 skip the classes, 4 skip the stores
  hipLaunchKernelGGL(poly_fix_kernel, dim3(B, n), dim3(256), lds, st, m, h, w, lda, Cout, dbg);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
