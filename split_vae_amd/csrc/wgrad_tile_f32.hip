// Weight gradient with LDS-resident tiles at the REFERENCE's precision (fp32 operands, fp32 accumulate: v_mfma_f32_16x16x4_f32):
//
//   dW[t][ci][co] += sum_pixels X[pix + tap t][ci] * dY[pix][co]          dbias[co] += sum_pixels dY[pix][co]
//
// (Conv2DBackpropFilter + BiasAddGrad of vae/trainer.py:137's tape.gradient, as the fp32 step runs them.)  The im2col form (wgrad.hip) gathers
// every input pixel once per tap through L2 -- d5: 4.2 M pixels x 36 taps x 32 channels x 4 B = 19 GB per launch, 3.9 ms at ~5 TB/s, 9 % of the
// fp32 matrix peak.  Here, as in wgrad_tile.hip (bf16), a workgroup stages a spatial tile of a 16-channel slice of the input with its halo ONCE
// (plain coalesced 16-B loads, natural NHWC order) and the matching dY tile; every tap's A operand is a shifted window of the same LDS patch.
// With K = 4 pixels per MFMA the operands are plain 4-B LDS reads (lane = (row / column lane & 15, pixel lane >> 4)): no transposed reads, and
// one read feeds COF (A) or TPW (B) 32-cycle MFMAs, so the loop is matrix-pipe-bound and the simple stage -> barrier -> multiply form is enough.
// Accumulators leave through the per-workgroup partial-sum slabs of wgrad_tile.hip in its fragment order <TPW, CIF = 1, COF>, summed by
// svk_wgrad_reduce_all in workgroup order: the fp32 step's weight gradients are bit-reproducible without SV_DETERMINISTIC.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"
#include "tile_stage.hip.h"
#include "conv_geom.h"

namespace {

// one LDS-DMA wave-instruction as inline assembly (base: wave-uniform tensor pointer, off: this lane's byte offset): the compiler orders every LDS access behind
// the builtin's vmcnt(0); with the assembly form the double-buffered loop waits where IT wants to (latent_gemm.hip: nt_dma16)
__device__ __forceinline__ void dma16_asm(const void* base, uint32_t off, const char* lds) {
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}

// TPW taps per wave (4 waves: 4 * TPW taps per workgroup = all of them), COF 16-column fragments of dY.  A workgroup owns a 16-channel (8 for
// an 8-channel input) slice of the input, blockIdx.y, and walks a contiguous run of tiles, blockIdx.x.
// LDY (dY floats per pixel), CW (channels of the slice), SX (x stride) and G4 (pixel groups of four per tile row) are compile-time: inside a tile row
// every operand address is the row's base register + an immediate, the reads of a whole row are issued before its MFMAs, and nothing but the
// MFMAs and one address add per tap and row is left in the loop (the first version -- run-time pitches -- issued ~35 VALU instructions per
// 18 MFMAs: address adds, masks, register rotation).
template <int TPW, int COF, int LDY, int CW, int SX, int G4>
__global__ __launch_bounds__(256, 2) void wgrad_tile_f32_kernel(const WgradTileMulti mg) {
  constexpr int PS = 4 * CW, YS = 4 * LDY;
  constexpr bool MASKB = LDY < 16 * COF;              // the last column fragment is half empty (the direct 8-column head)
  const WgradTileArgs& g = mg.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sIn = smem;                       // [NB][TIH][TIW] pixels of PS = 4 * CW bytes
  char* sDy = smem + g.in_bytes;          // [BM] pixels of YS = 4 * ldy bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int tap0 = wave * TPW, ci0 = (int)blockIdx.y * g.CW;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH, NB = 1 << g.lNB, BM = TW * TH * NB;
  const int ycols = g.ldy;

  // lane part of the operand addresses: pixel kq of a group of four consecutive tile pixels (one tile row: TW % 4 == 0), row / column lr
  const int in_lane = kq * SX * PS + lr * 4, dy_lane = kq * YS + lr * 4;
  // an 8-channel slice (e1) fills the fragment's rows 8..15 with the NEXT pixel in x (the pixel records are 32 B: lane lr reads channel lr & 7 of
  // pixel + (lr >> 3)) = the operand of tap (ky, kx + 1): one MFMA serves two taps, the tap list holds every other x tap (pairx; the host
  // only takes 8-channel inputs in this form)
  int tapoff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = min(tap0 + t, g.ntaps - 1);
    tapoff[t] = (((int)g.dy[tap] - g.y_lo) * g.TIW + ((int)g.dx[tap] - g.x_lo)) * PS + in_lane;
  }
  f32x4 acc[TPW][COF];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int j = 0; j < COF; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = ones^T . dY on the fragments the loop already holds: wave w takes the column fragments j = w, w + 4, ...
  constexpr int BJ = (COF + 3) / 4;
  f32x4 bacc[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) bacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = g.dbias != nullptr && blockIdx.y == 0;

  const float* __restrict__ Ab = (const float*)g.A + ci0;
  const float* __restrict__ Yb = (const float*)g.dY;
  const int lycp = g.lycp, dy_total = BM << lycp;
  const int per_wg = (g.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile_lo = (int)blockIdx.x * per_wg, tile_hi = min(g.ntiles, tile_lo + per_wg);
  const bool dma = g.dma != 0;
  // the MFMA rows of one staged tile
  auto rows = [&](const char* sIn, const char* sDy) {
      // ---- MFMA: K = pixels, four per instruction; one tile row (G4 groups of four pixels) at a time
      const int nrow = (SV_DBG(g.dbg) & 2) ? 0 : BM / (4 * G4);
      for (int row = 0; row < nrow; ++row) {
        const int ty = row & (TH - 1), bl = row >> g.lTH;                                   // wave-uniform
        const char* pin = sIn + ((bl * g.TIH + ty * g.S) * g.TIW) * PS;
        const char* pdy = sDy + dy_lane + row * (4 * G4 * YS);
        float bfr[G4][COF], afr[G4][TPW];
#pragma unroll
        for (int q = 0; q < G4; ++q)
#pragma unroll
          for (int j = 0; j < COF; ++j) {
            const float v = *(const float*)(pdy + q * 4 * YS + j * 64);
            bfr[q][j] = (MASKB && j * 16 + lr >= LDY) ? 0.f : v;
          }
#pragma unroll
        for (int t2 = 0; t2 < TPW; ++t2) {
          const char* pa = pin + tapoff[t2];
#pragma unroll
          for (int q = 0; q < G4; ++q) afr[q][t2] = *(const float*)(pa + q * 4 * SX * PS);
        }
#pragma unroll
        for (int q = 0; q < G4; ++q) {
          if (do_bias) {
#pragma unroll
            for (int j = 0; j < COF; ++j)
              if ((j & 3) == wave) bacc[j >> 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bfr[q][j], bacc[j >> 2], 0, 0, 0);
          }
#pragma unroll
          for (int t2 = 0; t2 < TPW; ++t2)
#pragma unroll
            for (int j = 0; j < COF; ++j) acc[t2][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[q][t2], bfr[q][j], acc[t2][j], 0, 0, 0);
        }
      }
  };
  if (g.db) {
    // DOUBLE-BUFFERED (plain / clamped inputs): the transfers of tile t + 1 -- inline-assembly LDS-DMA, which the compiler does not order the LDS reads behind --
    // are in flight beside the MFMAs of tile t; one barrier per tile.  (Single-buffered, the staging of a tile was exposed unless the CU's other workgroup
    // happened to be in its MFMA rows: 0.12 of the d4 weight gradient's 1.03 ms.)
    const int bufb = g.in_bytes + g.dy_bytes;
    const float inv_row = 1.0f / (float)(g.TIW << g.cl2), inv_h = 1.0f / (float)g.TIH;
    auto stage_db = [&](int tile, char* sI) {
      int t = tile;
      const int tx0 = (t % g.tilesX) << g.lTW; t /= g.tilesX;
      const int ty0 = (t % g.tilesY) << g.lTH; t /= g.tilesY;
      const int b0 = t << g.lNB;
      const int iy_base = ty0 * g.S + g.y_lo, ix_base = tx0 * g.SX + g.x_lo;
      const int cpp = 1 << g.cl2, ppr = g.TIW << g.cl2, total = NB * g.TIH * ppr;
      for (int base = wave * 64; base < total; base += 256) {
        const int L = base + lane;
        if (L >= total) continue;
        const int row = (int)(((float)L + 0.5f) * inv_row), r = L - row * ppr;
        const int bl = (int)(((float)row + 0.5f) * inv_h), iyl = row - bl * g.TIH;
        const int ixl = r >> g.cl2, c = r & (cpp - 1), b = b0 + bl;
        int iy = iy_base + iyl, ix = ix_base + ixl;
        bool ok = b < g.B;
        if (g.clampin) { iy = min(max(iy, 0), g.IH - 1); ix = min(max(ix, 0), g.IW - 1); }
        else ok = ok && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
        if (ok) dma16_asm(Ab, (uint32_t)((((b * g.IH + iy) * g.IW + ix) * g.lda + c * 4) * 4), sI + base * 16);
        else *(uint4*)(sI + L * 16) = make_uint4(0, 0, 0, 0);
      }
      char* sY = sI + g.in_bytes;
      for (int base = wave * 64; base < dy_total; base += 256) {
        const int q = base + lane;
        if (q >= dy_total) continue;
        const int r = q >> lycp, c = q & ((1 << lycp) - 1);
        const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
        const int b = b0 + bl;
        uint32_t off;
        if (g.dy_s2d) off = (uint32_t)((((b * 2 * g.OY + 2 * (ty0 + ty) + (c >> 2)) * (2 * g.OX) + 2 * (tx0 + tx) + ((c >> 1) & 1)) * 8 + (c & 1) * 4) * 4);
        else if (g.dy_os) off = (uint32_t)((((b * 2 * g.OY + 2 * (ty0 + ty) + g.dy_oy) * (2 * g.OX) + 2 * (tx0 + tx) + g.dy_ox) * g.ldy + c * 4) * 4);
        else off = (uint32_t)((((b * g.OY + ty0 + ty) * g.OX + tx0 + tx) * g.ldy + c * 4) * 4);
        if (b < g.B) dma16_asm(Yb, off, sY + base * 16);
        else *(uint4*)(sY + q * 16) = make_uint4(0, 0, 0, 0);
      }
    };
    int cur = 0;
    if (tile_lo < tile_hi) stage_db(tile_lo, smem);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                    // tile `tile` has landed for every wave; tile - 1 is consumed: the other buffer is free
      if (tile + 1 < tile_hi) stage_db(tile + 1, smem + (cur ^ 1) * bufb);
      rows(smem + cur * bufb, smem + cur * bufb + g.in_bytes);
      cur ^= 1;
    }
  } else
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    int t = tile;
    const int tx0 = (t % g.tilesX) << g.lTW; t /= g.tilesX;
    const int ty0 = (t % g.tilesY) << g.lTH; t /= g.tilesY;
    const int b0 = t << g.lNB;
    __syncthreads();                      // the previous tile is consumed
    if (!(SV_DBG(g.dbg) & 4)) {           // (ablation builds: bit 4 skips the staging, bit 2 the MFMA rows)
      const TileStageGeom sg = {g.B, g.IH, g.IW, g.lda, g.cl2, g.TIW, g.TIH, g.PS, NB, 0};
      // (ups: the layer's input is the 2x bilinear resize of the LOW-RES tensor A, blended on the fly with upsample2x_fwd's own arithmetic)
      if (g.s2d3) stage_tile_s2d3<256>((const float*)g.A, sg, b0, ty0 + g.y_lo, tx0 + g.x_lo, sIn, tid);      // e1: the padded RGB tensor through its space-to-depth view
      else if (g.ups) stage_tile_upsampled<float>(Ab, sg, b0, ty0 * g.S + g.y_lo, tx0 * g.SX + g.x_lo, sIn, tid);
      else if (dma) {                     // LDS-DMA: the whole tile in flight at once (tile_stage.hip.h)
        if (g.clampin) stage_tile_plain_dma<float, 4, true>(Ab, sg, b0, ty0 * g.S + g.y_lo, tx0 * g.SX + g.x_lo, sIn, lane, wave);   // polyphase forms: the edge-clamped low-res tensor
        else stage_tile_plain_dma<float, 4, false>(Ab, sg, b0, ty0 * g.S + g.y_lo, tx0 * g.SX + g.x_lo, sIn, lane, wave);
      }
      else if (g.clampin) stage_tile_plain<float, 256, true>(Ab, sg, b0, ty0 * g.S + g.y_lo, tx0 * g.SX + g.x_lo, sIn, tid);
      else stage_tile_plain<float>(Ab, sg, b0, ty0 * g.S + g.y_lo, tx0 * g.SX + g.x_lo, sIn, tid);
    }
    // dY tile: piece q = pixel * (ldy / 4) + c lives at byte 16 q (YS = 4 ldy): by LDS-DMA too (the register form was ONE load in flight per lane: eight dependent round trips per tile)
    for (int base = dma ? wave * 64 : tid; base < ((SV_DBG(g.dbg) & 4) ? 0 : dy_total); base += 256) {
      const int q = dma ? base + lane : base;
      if (q >= dy_total) continue;
      const int r = q >> lycp, c = q & ((1 << lycp) - 1);
      const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
      const int b = b0 + bl;
      const float* src;
      if (g.dy_s2d)          // merged polyphase head: column c * 4 = (py * 2 + px) * 8 + co of low-res pixel (i, j) is channel co of hi-res pixel (2i + py, 2j + px)
        src = Yb + ((int64_t)(b * 2 * g.OY + 2 * (ty0 + ty) + (c >> 2)) * (2 * g.OX) + 2 * (tx0 + tx) + ((c >> 1) & 1)) * 8 + (c & 1) * 4;
      else if (g.dy_os)      // one parity class of the per-class polyphase form: hi-res pixel (2i + dy_oy, 2j + dy_ox)
        src = Yb + ((int64_t)(b * 2 * g.OY + 2 * (ty0 + ty) + g.dy_oy) * (2 * g.OX) + 2 * (tx0 + tx) + g.dy_ox) * g.ldy + c * 4;
      else src = Yb + ((int64_t)(b * g.OY + ty0 + ty) * g.OX + tx0 + tx) * g.ldy + c * 4;
      if (dma && b < g.B) stage_dma16(src, sDy + base * 16);
      else *(uint4*)(sDy + q * 16) = b < g.B ? *(const uint4*)src : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    rows(sIn, sDy);
  }

  // ---- flush: one slab per workgroup in the fragment order of wgrad_reduce <TPW, 1, COF>; without a workspace, fp32 atomics into dW
  if (!g.slab) {                          // D row = channel (lane >> 4) * 4 + register, column = output channel lane & 15
#pragma unroll
    for (int t2 = 0; t2 < TPW; ++t2) {
      const int tap = tap0 + t2;
      if (tap >= g.ntaps) continue;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int cl = kq * 4 + r4, ci = g.pairx ? (cl & 7) : ci0 + cl;
        if (ci >= g.Cin_real || (!g.pairx && cl >= g.CW)) continue;
        const int otap = g.pairx ? 2 * tap + (cl >> 3) : tap;
#pragma unroll
        for (int j = 0; j < COF; ++j) {
          const int co = j * 16 + lr;
          const int64_t di = co < g.N ? dw_index(otap, ci, co, g.Cin_real, g.N, g.fold_kw, g.fold_c, g.s2d3) : -1;
          if (di >= 0) atomicAdd(g.dW + di, acc[t2][j][r4]);
        }
      }
    }
  } else {
    float* sl = g.slab + ((((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave) * (TPW * COF)) * 256 + lane;
#pragma unroll
    for (int t2 = 0; t2 < TPW; ++t2)
#pragma unroll
      for (int j = 0; j < COF; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) sl[((t2 * COF + j) * 4 + r4) * 64] = acc[t2][j][r4];
  }
  if (do_bias) {                          // every row of a column-sum fragment is the column sum: row 0 = lanes 0..15, register 0
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < COF; ++j)
        if ((j & 3) == wave && j * 16 + lane < ycols && j * 16 + lane < g.N) {
          const int col = j * 16 + lane;
          if (g.bslab) g.bslab[(int64_t)blockIdx.x * 128 + col] = bacc[j >> 2][0];     // (folded columns: the reduce adds px 0 and px 1)
          else if (!g.fold_kw || (col & 7) < g.fold_c) atomicAdd(g.dbias + (g.fold_kw ? (col & 7) : col), bacc[j >> 2][0]);
        }
    }
  }
}

// ---- the four parity classes of a per-class polyphase layer (conv_geom.h: svg_polyc; polyc_wgrad.hip) in ONE launch: the clamped low-res input tile is staged ONCE
// per tile and serves the 25 + 20 + 20 + 16 taps of the four classes; only dY -- each class reads its own sub-pixel of the hi-res gradient -- is re-staged per class.
// As four launches the input moved four times (PMC: 713 MB per d4 weight gradient against 201 MB algorithmic).  Accumulators of all four classes live in registers
// (42 fragments x 4 = 168); slabs leave per class in the single-class kernel's layout, so the reduce and the projection are unchanged.
struct WgradPolycArgs {
  WgradTileArgs g;                 // geometry (taps unused), A, bslab, dbias
  const float* dY; float* slab[4];
  int ntaps[4];
  int8_t cdy[4][28], cdx[4][28];
};
struct WgradPolycMulti { WgradPolycArgs a[SV_WGRAD_MAX_MULTI]; };

template <int TPW, int COF, int G4, int YS, int SXPS>
__device__ __forceinline__ void polyc_rows(f32x4 (&acc)[TPW][COF], f32x4 (&bacc)[(COF + 3) / 4], const int (&tapoff)[TPW], const char* sIn, const char* sDy,
                                           int dy_lane, int nrow, int TH, int lTH, int TIH, int TIW, int PS, bool do_bias, int wave) {
  for (int row = 0; row < nrow; ++row) {
    const int ty = row & (TH - 1), bl = row >> lTH;
    const char* pin = sIn + ((bl * TIH + ty) * TIW) * PS;
    const char* pdy = sDy + dy_lane + row * (4 * G4 * YS);
    float bfr[G4][COF], afr[G4][TPW];
#pragma unroll
    for (int q = 0; q < G4; ++q)
#pragma unroll
      for (int j = 0; j < COF; ++j) bfr[q][j] = *(const float*)(pdy + q * 4 * YS + j * 64);
#pragma unroll
    for (int t2 = 0; t2 < TPW; ++t2) {
      const char* pa = pin + tapoff[t2];
#pragma unroll
      for (int q = 0; q < G4; ++q) afr[q][t2] = *(const float*)(pa + q * SXPS);
    }
#pragma unroll
    for (int q = 0; q < G4; ++q) {
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < COF; ++j)
          if ((j & 3) == wave) bacc[j >> 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bfr[q][j], bacc[j >> 2], 0, 0, 0);
      }
#pragma unroll
      for (int t2 = 0; t2 < TPW; ++t2)
#pragma unroll
        for (int j = 0; j < COF; ++j) acc[t2][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[q][t2], bfr[q][j], acc[t2][j], 0, 0, 0);
    }
  }
}

template <int T0, int T1, int T2, int T3, int COF, int LDY, int CW, int G4>
__global__ __launch_bounds__(256, 2) void wgrad_polyc_f32_kernel(const WgradPolycMulti mg) {
  constexpr int PS = 4 * CW, YS = 4 * LDY;
  const WgradPolycArgs& pa = mg.a[blockIdx.z];
  const WgradTileArgs& g = pa.g;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sIn = smem;
  char* sDy = smem + g.in_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int ci0 = (int)blockIdx.y * g.CW;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH, NB = 1 << g.lNB, BM = TW * TH * NB;
  const int in_lane = kq * PS + lr * 4, dy_lane = kq * YS + lr * 4;
  // a class with T == 0 is not part of this launch (the two-pair form: classes {0, 3} and {1, 2} as two launches of 88 / 80 accumulator registers)
  int off0[T0 ? T0 : 1], off1[T1 ? T1 : 1], off2[T2 ? T2 : 1], off3[T3 ? T3 : 1];
  auto taps = [&](int c, int TPWc, int* out) {
    for (int t = 0; t < TPWc; ++t) {
      const int tap = min(wave * TPWc + t, pa.ntaps[c] - 1);
      out[t] = (((int)pa.cdy[c][tap] - g.y_lo) * g.TIW + ((int)pa.cdx[c][tap] - g.x_lo)) * PS + in_lane;
    }
  };
  taps(0, T0, off0); taps(1, T1, off1); taps(2, T2, off2); taps(3, T3, off3);
  f32x4 a0[T0 ? T0 : 1][COF], a1[T1 ? T1 : 1][COF], a2[T2 ? T2 : 1][COF], a3[T3 ? T3 : 1][COF];
#pragma unroll
  for (int j = 0; j < COF; ++j) {
#pragma unroll
    for (int t = 0; t < T0; ++t) a0[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T1; ++t) a1[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T2; ++t) a2[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T3; ++t) a3[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  constexpr int BJ = (COF + 3) / 4;
  f32x4 bacc[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) bacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = g.dbias != nullptr && blockIdx.y == 0;
  const float* __restrict__ Ab = (const float*)g.A + ci0;
  const float* __restrict__ Yb = pa.dY;
  const int lycp = g.lycp, dy_total = BM << lycp;
  const int per_wg = (g.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile_lo = (int)blockIdx.x * per_wg, tile_hi = min(g.ntiles, tile_lo + per_wg);
  const int nrow = BM / (4 * G4);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    int t = tile;
    const int tx0 = (t % g.tilesX) << g.lTW; t /= g.tilesX;
    const int ty0 = (t % g.tilesY) << g.lTH; t /= g.tilesY;
    const int b0 = t << g.lNB;
    __syncthreads();                      // the previous tile (its last class) is consumed
    {
      const TileStageGeom sg = {g.B, g.IH, g.IW, g.lda, g.cl2, g.TIW, g.TIH, g.PS, NB, 0};
      stage_tile_plain<float, 256, true>(Ab, sg, b0, ty0 + g.y_lo, tx0 + g.x_lo, sIn, tid);
    }
    bool first = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if ((c == 0 && !T0) || (c == 1 && !T1) || (c == 2 && !T2) || (c == 3 && !T3)) continue;
      if (!first) __syncthreads();        // the previous class's dY tile is consumed
      first = false;
      for (int q = tid; q < dy_total; q += 256) {
        const int r = q >> lycp, cc = q & ((1 << lycp) - 1);
        const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
        const int b = b0 + bl;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b < g.B) v = *(const uint4*)(Yb + ((int64_t)(b * 2 * g.OY + 2 * (ty0 + ty) + (c >> 1)) * (2 * g.OX) + 2 * (tx0 + tx) + (c & 1)) * g.ldy + cc * 4);
        *(uint4*)(sDy + r * YS + cc * 16) = v;
      }
      __syncthreads();
      if constexpr (T0 > 0) if (c == 0) polyc_rows<T0, COF, G4, YS, 4 * PS>(a0, bacc, off0, sIn, sDy, dy_lane, nrow, TH, g.lTH, g.TIH, g.TIW, PS, do_bias, wave);
      if constexpr (T1 > 0) if (c == 1) polyc_rows<T1, COF, G4, YS, 4 * PS>(a1, bacc, off1, sIn, sDy, dy_lane, nrow, TH, g.lTH, g.TIH, g.TIW, PS, do_bias, wave);
      if constexpr (T2 > 0) if (c == 2) polyc_rows<T2, COF, G4, YS, 4 * PS>(a2, bacc, off2, sIn, sDy, dy_lane, nrow, TH, g.lTH, g.TIH, g.TIW, PS, do_bias, wave);
      if constexpr (T3 > 0) if (c == 3) polyc_rows<T3, COF, G4, YS, 4 * PS>(a3, bacc, off3, sIn, sDy, dy_lane, nrow, TH, g.lTH, g.TIH, g.TIW, PS, do_bias, wave);
    }
  }
  auto flush = [&](int c, int TPWc, auto& acc) {
    float* sl = pa.slab[c] + ((((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave) * (TPWc * COF)) * 256 + lane;
#pragma unroll
    for (int t2 = 0; t2 < (int)(sizeof(acc) / sizeof(acc[0])); ++t2)
#pragma unroll
      for (int j = 0; j < COF; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) sl[((t2 * COF + j) * 4 + r4) * 64] = acc[t2][j][r4];
  };
  if constexpr (T0 > 0) flush(0, T0, a0);
  if constexpr (T1 > 0) flush(1, T1, a1);
  if constexpr (T2 > 0) flush(2, T2, a2);
  if constexpr (T3 > 0) flush(3, T3, a3);
  if (do_bias && lane < 16) {
#pragma unroll
    for (int j = 0; j < COF; ++j)
      if ((j & 3) == wave && j * 16 + lane < g.N) g.bslab[(int64_t)blockIdx.x * 128 + j * 16 + lane] = bacc[j >> 2][0];
  }
}

template <int TPW, int COF, int LDY, int CW, int SX, int G4>
int launch_f32(const WgradTileArgs* a, int n, int msplit, int groups, size_t lds, hipStream_t st) {
  WgradTileMulti m;
  for (int i = 0; i < n; ++i) m.a[i] = a[i];
  sv_ensure_dynamic_lds((const void*)wgrad_tile_f32_kernel<TPW, COF, LDY, CW, SX, G4>, lds);
  hipLaunchKernelGGL((wgrad_tile_f32_kernel<TPW, COF, LDY, CW, SX, G4>), dim3(msplit, groups, n), dim3(256), lds, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // namespace

// The seven conv layers of the SPLIT-VAE encoders / decoders and the 3 x 3 layers of LG-SPAIR's object networks.  SV_E_UNSUPPORTED:
// any other shape, a missing / small workspace -- the caller falls back to the im2col kernel.
#define F32_REJ(why) do { if (trace) fprintf(stderr, "wgrad_tile_f32: not taken (%s)\n", why); return SV_E_UNSUPPORTED; } while (0)
int svk_wgrad_tile_f32_multi(const WgradArgs* wv, int n, hipStream_t st) {
  static const bool trace = getenv("SV_TRACE_DISPATCH") != nullptr;
  static const bool off = getenv("SV_NO_WGRAD_TILE_F32") != nullptr;
  if (off || n < 1 || n > SV_WGRAD_MAX_MULTI) F32_REJ("off / problems");
  const WgradArgs& w = wv[0];
  if (w.lOY < 0 || w.lOX < 0 || w.S > 2 || (w.S != w.SX && !w.fold_kw) || (w.ups && w.S != 1) || w.ycols != w.ldy) F32_REJ("form");
  // polyphase forms (conv_geom.h: svg_poly / svg_polyc): edge-clamped low-res input, dY through its space-to-depth view (merged head: 32 columns) or one parity class
  const bool polyf = w.clampin != 0;
  if (polyf && (w.S != 1 || w.ups || w.fold_kw || !(w.dy_s2d == 8 ? w.ldy == 32 : w.dy_os == 2))) F32_REJ("polyphase form");
  if (!polyf && (w.dy_s2d || w.dy_os)) F32_REJ("dY view");
  if (w.s2d3 && (w.S != 1 || w.ups || polyf || w.fold_kw || w.Cin_pad != 16 || w.ntaps != 9)) F32_REJ("space-to-depth form");
  const int OY = 1 << w.lOY, OX = 1 << w.lOX, cin = w.Cin_pad, ldy = w.ldy, nt = w.ntaps;
  if (OX < 4 || OY * OX < 16 || ldy > 128 || (ldy & 7) || (nt != 36 && nt != 16 && nt != 9 && !(nt == 42 && w.fold_kw) && !(polyf && (nt == 25 || nt == 20 || nt == 12 || nt == 9)))) {
    if (trace) fprintf(stderr, "wgrad_tile_f32: OY %d OX %d ldy %d taps %d cin %d\n", OY, OX, ldy, nt, cin);
    F32_REJ("grid / taps");
  }
  const int CW = cin >= 16 ? 16 : 8;
  if (cin % CW || (cin != 8 && ilog2_exact(cin) < 0)) F32_REJ("channels");
  // 8-channel inputs (e1's padded RGB), even KW, taps y-major: tap pairs (wgrad_tile.hip's pairx)
  static const bool no_pairx = getenv("SV_WTF32_NO_PAIRX") != nullptr;
  bool pairx = CW == 8 && cin == 8 && nt == 36 && !w.fold_kw && !no_pairx;
  for (int u = 0; u < nt / 2 && pairx; ++u) pairx = w.dy[2 * u + 1] == w.dy[2 * u] && w.dx[2 * u + 1] == w.dx[2 * u] + 1;
  const int ntk = pairx ? nt / 2 : nt;                      // taps the kernel walks
  const int TPW = (ntk + 3) / 4, COF = (ldy + 15) / 16;      // (42 folded taps: 11 per wave, the last two slots repeat tap 41 and are dropped by the reduce)
  // (an 8-channel input with an odd kernel width -- the first 3 x 3 layer of LG-SPAIR's object encoder -- runs without pairs: fragment rows 8..15
  //  hold the next pixel's channels and are dropped by the flush / the reduce)
  if (CW == 8 && !pairx && nt != 9) F32_REJ("8-channel input without tap pairs");
  int y_lo = 127, y_hi = -127, x_lo = 127, x_hi = -127;
  for (int i = 0; i < nt; ++i) {
    y_lo = w.dy[i] < y_lo ? w.dy[i] : y_lo; y_hi = w.dy[i] > y_hi ? w.dy[i] : y_hi;
    x_lo = w.dx[i] < x_lo ? w.dx[i] : x_lo; x_hi = w.dx[i] > x_hi ? w.dx[i] : x_hi;
  }
  // pixels per tile: the largest of 256 .. 32 whose input patch + dY tile leave two workgroups per CU their LDS
  WgradTileArgs a;
  static const int bm_max = getenv("SV_WTF32_BM") ? atoi(getenv("SV_WTF32_BM")) : 256;      // A/B knobs
  static const int wgs = getenv("SV_WTF32_WGS") ? atoi(getenv("SV_WTF32_WGS")) : 512;
  // LDS per workgroup the tile may take.  78 KB (two workgroups per CU) gives the best launch ALONE on the chip; in the step the weight gradients run on the side
  // stream beside the input-gradient chain, and 52 KB tiles (three to five workgroups per CU fit beside the other stream's) make the 512-image fp32 step 2.3 %
  // shorter: 9.82-9.90 -> 9.60-9.62 ms (40 KB: 9.68; 128-pixel tiles at 78 KB: 9.77; profiles/r05_wtf32_lds_ab.txt), the serial rows 0-3 % longer
  static const int lds_max = getenv("SV_WTF32_LDS") ? atoi(getenv("SV_WTF32_LDS")) : 52000;
  // double-buffered staging (plain / clamped inputs; the kernel's g.db loop): both buffers inside SV_WTF32_DB_LDS bytes.  OPT-IN (default 0 = single buffer).  Measured
  // (2 x 512 images): alone on the chip the d4 weight gradient goes 1.020 -> 0.990 ms and d5's 0.605 -> 0.590 at 64 KB (e2: 0.335 -> 0.359, its stride-2 tile halves),
  // but the STEP goes 9.19 -> 9.35 ms (48 KB: 9.32, 78 KB: 9.38, 120 KB: 9.60): the second buffer takes the LDS that the other streams' workgroups lived in
  static const int db_lds = getenv("SV_WTF32_DB_LDS") ? atoi(getenv("SV_WTF32_DB_LDS")) : 0;
  static const bool no_dma = getenv("SV_WT32_NO_DMA") != nullptr;          // A/B: the register staging
  const bool db = db_lds > 0 && !no_dma && !w.ups && !w.s2d3;
  int BM = bm_max;
  for (;; BM >>= 1) {
    if (BM < 32) F32_REJ("tile");
    if (OY * OX < BM && (BM % (OY * OX))) continue;
    const int lTW = OX >= 16 ? 4 : w.lOX;
    int lTH = 0;
    while ((1 << (lTW + lTH)) < BM && (1 << lTH) < OY) ++lTH;
    int lNB = 0;
    while ((1 << (lTW + lTH + lNB)) < BM) ++lNB;
    const int TW = 1 << lTW, TH = 1 << lTH, NB = 1 << lNB;
    memset(&a, 0, sizeof(a));
    a.lTW = lTW; a.lTH = lTH; a.lNB = lNB;
    a.TIW = (TW - 1) * w.SX + (x_hi - x_lo) + 1; a.TIH = (TH - 1) * w.S + (y_hi - y_lo) + 1;
    a.PS = CW * 4; a.YS = ldy * 4;
    a.in_bytes = (NB * a.TIH * a.TIW * a.PS + 64 + 15) / 16 * 16;      // slack: the masked lanes of a half-filled fragment read past the last pixel
    a.dy_bytes = BM * a.YS + 64;
    if (db ? 2 * (a.in_bytes + a.dy_bytes) <= db_lds : a.in_bytes + a.dy_bytes <= lds_max) break;
  }
  const int TW = 1 << a.lTW, TH = 1 << a.lTH, NB = 1 << a.lNB;
  const int B = w.M >> (w.lOY + w.lOX);
  a.B = B; a.IH = w.IH; a.IW = w.IW; a.lda = w.lda; a.S = w.S; a.SX = w.SX; a.assign = w.assign; a.ups = w.ups;
  a.fold_kw = w.fold_kw; a.fold_c = w.fold_c;
  a.clampin = w.clampin; a.dy_s2d = w.dy_s2d; a.dy_os = w.dy_os; a.dy_oy = w.dy_oy; a.dy_ox = w.dy_ox; a.s2d3 = w.s2d3;
  a.contig = 1; a.CW = CW; a.ncg = cin / CW; a.cl2 = ilog2_exact(CW / 4);
  a.dma = no_dma ? 0 : 1;
  a.db = db ? 1 : 0;
  static const int dbg = getenv("SV_WT32_DBG") ? atoi(getenv("SV_WT32_DBG")) : 0;       // (read by SV_DEBUG_KNOBS builds only)
  a.dbg = dbg;
  a.OY = OY; a.OX = OX; a.tilesX = OX / TW; a.tilesY = OY / TH;
  a.ntiles = a.tilesX * a.tilesY * ((B + NB - 1) / NB);
  a.y_lo = y_lo; a.x_lo = x_lo;
  a.ldy = ldy; a.lycp = ilog2_exact(ldy / 4);
  if (a.lycp < 0) F32_REJ("dY pitch");
  a.Cin_real = w.Cin_real; a.N = w.N; a.ntaps = ntk; a.pairx = pairx ? 1 : 0;
  for (int u = 0; u < ntk; ++u) { a.dy[u] = w.dy[pairx ? 2 * u : u]; a.dx[u] = w.dx[pairx ? 2 * u : u]; }
  const int groups = a.ncg;
  // two resident workgroups per CU over the whole launch, every workgroup at least one tile
  int msplit = (wgs + groups * n - 1) / (groups * n);
  if (msplit > a.ntiles) msplit = a.ntiles;
  if (msplit < 1) msplit = 1;
  const int64_t PER = 4LL * TPW * COF * 256;
  const int64_t need = (int64_t)msplit * groups * PER * 4 + (int64_t)msplit * 128 * 4;
  bool slab = true;                      // no (or a small) workspace: fp32 atomics straight into dW -- kept for the x-packed head and the layers behind
  for (int i = 0; i < n; ++i) slab = slab && wv[i].ws && wv[i].ws_bytes >= need;          // a fused resize, which the im2col kernel cannot do; every other shape falls back to it
  if (!slab && !w.fold_kw && !w.ups && !w.s2d3) F32_REJ("workspace");        // (s2d3: the im2col kernel cannot form the space-to-depth view: fp32 atomics flush)
  WgradTileArgs av[SV_WGRAD_MAX_MULTI];
  WgradReduceDesc rd[SV_WGRAD_MAX_MULTI];
  for (int i = 0; i < n; ++i) {
    av[i] = a;
    av[i].A = wv[i].A; av[i].dY = wv[i].dY; av[i].dW = wv[i].dW; av[i].dbias = wv[i].dbias;
    av[i].slab = slab ? wv[i].ws : nullptr;
    av[i].bslab = slab && wv[i].dbias ? wv[i].ws + (int64_t)msplit * groups * PER : nullptr;
    rd[i] = WgradReduceDesc{av[i].slab, wv[i].dW, av[i].bslab, wv[i].dbias, msplit, groups, a.ncg, CW, a.Cin_real, a.N, ntk, w.fold_kw, w.fold_c, a.pairx, a.assign, TPW, 1, COF, w.s2d3};
  }
  const size_t lds = ((size_t)a.in_bytes + a.dy_bytes) * (a.db ? 2 : 1);
  // <taps per wave, column fragments, dY floats per pixel, slice channels, x stride, pixel groups per tile row>: the layers of the model
  const int G4 = TW / 4, SXv = w.SX;
  int rc = SV_E_UNSUPPORTED;
#define F32_CASE(T, C, L, W, X, G) else if (TPW == T && COF == C && ldy == L && CW == W && SXv == X && G4 == G) rc = launch_f32<T, C, L, W, X, G>(av, n, msplit, groups, lds, st)
#define F32_LAYER(T, C, L, W, X) F32_CASE(T, C, L, W, X, 4); F32_CASE(T, C, L, W, X, 2); F32_CASE(T, C, L, W, X, 1)     // 16-, 8-, 4-pixel tile rows
  if (false) {}
  F32_LAYER(11, 1, 16, 16, 2);      // d5, x-packed (pixel pairs)
  F32_LAYER(9, 1, 8, 16, 1);        // d5, direct form (SV_NO_PACKX)
  F32_LAYER(9, 2, 32, 16, 1);       // d4
  F32_LAYER(7, 2, 32, 16, 1);       // polyphase: the head's merged 25-tap form (32 = 4 parities x 8 columns) and d4's class (0, 0) (25 taps on 28 slots)
  F32_LAYER(5, 2, 32, 16, 1);       // d4's classes (0, 1) and (1, 0): 20 taps
  F32_LAYER(4, 2, 32, 16, 1);       // d4's class (1, 1): 16 taps
  F32_LAYER(4, 4, 64, 16, 1);       // d3
  F32_LAYER(4, 8, 128, 16, 1);      // d2
  F32_LAYER(4, 8, 128, 16, 2);      // e3
  F32_LAYER(9, 4, 64, 16, 2);       // e2
  F32_LAYER(5, 2, 32, 8, 2);        // e1 (tap pairs)
  // the 3 x 3 layers of LG-SPAIR's object encoder / decoder on 32 x 32 glimpses (9 taps on 12 slots: three per wave, the reduce drops the padding)
  F32_LAYER(3, 4, 64, 16, 2);       // object encoder conv2
  F32_LAYER(3, 4, 64, 16, 1);       // object decoder d2
  F32_LAYER(3, 2, 32, 16, 1);       // object decoder d3; e1 in space-to-depth form (svg_s2d3)
  F32_LAYER(3, 8, 128, 16, 1);      // SPLIT-GMVAE's first encoder layer (3 -> 128, vae/model.py:50) in space-to-depth form
  F32_LAYER(3, 1, 8, 16, 1);        // object decoder d5 (RGB + alpha)
  F32_LAYER(3, 2, 32, 8, 2);        // object encoder conv1 (RGB padded to 8 channels, no tap pairs: half of every fragment is dropped)
#undef F32_LAYER
#undef F32_CASE
  else F32_REJ("no instantiation");
  if (rc != SV_OK) return rc;
  if (w.ev_mid[0]) { (void)hipEventRecord(w.ev_mid[0], st); (void)hipEventRecord(w.ev_mid[1], st); }
  if (!slab) return SV_OK;
  if (w.defer && w.n_defer && *w.n_defer + n <= 64) {
    for (int i = 0; i < n; ++i) w.defer[(*w.n_defer)++] = rd[i];
    return SV_OK;
  }
  return svk_wgrad_reduce_all(rd, n, st);
}


// The four class problems of n <= 2 per-class polyphase layers as ONE launch (see wgrad_polyc_f32_kernel).  cls[c * n + i]: class c of network i (svg_polyc_wgrad_args
// + pointers; ws = the class's slab region); descriptors of the 4 n slab reduces are appended to rd.  SV_E_UNSUPPORTED: no instantiation / small workspace (nothing launched).
int svk_wgrad_polyc_f32_multi(const WgradArgs* cls, int n, int mask, WgradReduceDesc* rd, int* nrd, hipStream_t st) {
  static const bool trace = getenv("SV_TRACE_DISPATCH") != nullptr;
  // OPT-IN (SV_WGRAD_POLYC_FUSED=1).  Measured (2 x 512 images, profiles/r05_polyc_fused_ab.txt): alone on the chip the d4 weight gradient goes 1.078 -> 0.949 ms (the input
  // tile moves once), but the 512-image STEP goes 9.355 -> 9.54 ms: at 239 VGPRs the kernel leaves no room beside it for the other streams' workgroups, and the step lives
  // on that co-residency (same finding as the 52-KB tiles).  The default keeps the four class launches.
  // mask: the classes of THIS launch (bit c).  0xF: all four; 0x9 and 0x6: the two-pair form (SV_WGRAD_POLYC_FUSED=2) -- classes {0, 3} (25 + 16 taps, 88 accumulator
  // registers) and {1, 2} (20 + 20, 80): the input tile moves twice instead of four times and the workgroup keeps the register budget of a co-resident one.  The bias partial
  // rides with the lowest class of the mask (that class's dbias / slab region).
  if (n < 1 || n > SV_WGRAD_MAX_MULTI || (mask != 0xF && mask != 0x9 && mask != 0x6)) F32_REJ("problems / classes");
  const WgradArgs& w = cls[0];
  if (w.lOY < 0 || w.lOX < 0 || w.S != 1 || !w.clampin || w.dy_os != 2 || w.ldy != 32 || w.ycols != 32 || w.N != 32) F32_REJ("form");
  const int ntc[4] = {cls[0].ntaps, cls[n].ntaps, cls[2 * n].ntaps, cls[3 * n].ntaps};
  if (ntc[0] != 25 || ntc[1] != 20 || ntc[2] != 20 || ntc[3] != 16) F32_REJ("taps");
  const int OY = 1 << w.lOY, OX = 1 << w.lOX, cin = w.Cin_pad, CW = 16;
  if (OX < 4 || OY * OX < 16 || cin % CW) F32_REJ("grid / channels");
  static const int bm_max = getenv("SV_WTF32_BM") ? atoi(getenv("SV_WTF32_BM")) : 256;
  static const int wgs = getenv("SV_WTF32_WGS") ? atoi(getenv("SV_WTF32_WGS")) : 512;
  static const int lds_max = getenv("SV_WTF32_LDS") ? atoi(getenv("SV_WTF32_LDS")) : 52000;
  WgradTileArgs a;
  int BM = bm_max;
  for (;; BM >>= 1) {
    if (BM < 32) F32_REJ("tile");
    if (OY * OX < BM && (BM % (OY * OX))) continue;
    const int lTW = OX >= 16 ? 4 : w.lOX;
    int lTH = 0;
    while ((1 << (lTW + lTH)) < BM && (1 << lTH) < OY) ++lTH;
    int lNB = 0;
    while ((1 << (lTW + lTH + lNB)) < BM) ++lNB;
    const int TW = 1 << lTW, TH = 1 << lTH, NB = 1 << lNB;
    memset(&a, 0, sizeof(a));
    a.lTW = lTW; a.lTH = lTH; a.lNB = lNB;
    a.TIW = TW + 4; a.TIH = TH + 4;                          // the union of the classes' windows: offsets -2 .. 2
    a.PS = CW * 4; a.YS = 32 * 4;
    a.in_bytes = (NB * a.TIH * a.TIW * a.PS + 64 + 15) / 16 * 16;
    a.dy_bytes = BM * a.YS + 64;
    if (a.in_bytes + a.dy_bytes <= lds_max) break;
  }
  const int TW = 1 << a.lTW, TH = 1 << a.lTH, NB = 1 << a.lNB, G4 = TW / 4;
  const int B = w.M >> (w.lOY + w.lOX);
  a.B = B; a.IH = w.IH; a.IW = w.IW; a.lda = w.lda; a.S = 1; a.SX = 1; a.assign = 1; a.clampin = 1;
  a.contig = 1; a.CW = CW; a.ncg = cin / CW; a.cl2 = 2;
  a.OY = OY; a.OX = OX; a.tilesX = OX / TW; a.tilesY = OY / TH;
  a.ntiles = a.tilesX * a.tilesY * ((B + NB - 1) / NB);
  a.y_lo = -2; a.x_lo = -2; a.ldy = 32; a.lycp = 3; a.Cin_real = w.Cin_real; a.N = 32;
  const int groups = a.ncg;
  int msplit = (wgs + groups * n - 1) / (groups * n);
  if (msplit > a.ntiles) msplit = a.ntiles;
  if (msplit < 1) msplit = 1;
  static const int tpw[4] = {7, 5, 5, 4};
  const int c_lead = mask & 1 ? 0 : 1, nrd0 = *nrd;
  WgradPolycMulti m;
  for (int i = 0; i < n; ++i) {
    WgradPolycArgs& p = m.a[i];
    p.g = a;
    p.g.A = cls[i].A; p.dY = (const float*)cls[i].dY;
    for (int c = 0; c < 4; ++c) {
      p.slab[c] = nullptr; p.ntaps[c] = 0;
      if (!(mask >> c & 1)) continue;
      const WgradArgs& wc = cls[c * n + i];
      const bool lead = c == c_lead;
      const int64_t PER = 4LL * tpw[c] * 2 * 256;
      const int64_t need = (int64_t)msplit * groups * PER * 4 + (lead ? (int64_t)msplit * 128 * 4 : 0);
      if (!wc.ws || wc.ws_bytes < need) { *nrd = nrd0; F32_REJ("workspace"); }
      p.slab[c] = wc.ws;
      p.ntaps[c] = wc.ntaps;
      for (int t = 0; t < wc.ntaps; ++t) { p.cdy[c][t] = wc.dy[t]; p.cdx[c][t] = wc.dx[t]; }
      if (lead) { p.g.dbias = wc.dbias; p.g.bslab = wc.dbias ? wc.ws + (int64_t)msplit * groups * PER : nullptr; }
      rd[(*nrd)++] = WgradReduceDesc{p.slab[c], wc.dW, lead ? p.g.bslab : nullptr, lead ? wc.dbias : nullptr, msplit, groups, a.ncg, CW, a.Cin_real, 32,
                                     wc.ntaps, 0, 0, 0, 1, tpw[c], 1, 2, 0};
    }
    if (!p.g.bslab) p.g.dbias = nullptr;
  }
  const size_t lds = ((size_t)a.in_bytes + a.dy_bytes) * (a.db ? 2 : 1);
  const dim3 grid(msplit, groups, n);
#define POLYC_LAUNCH(T0, T1, T2, T3, G) { sv_ensure_dynamic_lds((const void*)wgrad_polyc_f32_kernel<T0, T1, T2, T3, 2, 32, 16, G>, lds); \
    hipLaunchKernelGGL((wgrad_polyc_f32_kernel<T0, T1, T2, T3, 2, 32, 16, G>), grid, dim3(256), lds, st, m); }
#define POLYC_CASE(G) if (G4 == G) { if (mask == 0xF) POLYC_LAUNCH(7, 5, 5, 4, G) else if (mask == 0x9) POLYC_LAUNCH(7, 0, 0, 4, G) else POLYC_LAUNCH(0, 5, 5, 0, G) }
  POLYC_CASE(4) else POLYC_CASE(2) else POLYC_CASE(1) else { *nrd = nrd0; F32_REJ("tile row"); }
#undef POLYC_CASE
#undef POLYC_LAUNCH
  SV_LAUNCH_CHECK();
  return SV_OK;
}
