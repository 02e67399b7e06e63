// Weight gradient with LDS-resident tiles (bf16):  dW[t][ci][co] += sum_pixels X[pix + tap t][ci] * dY[pix][co]
//
// The im2col GEMM form (wgrad.hip) re-gathers the input once per tap through L2 and its tiles have
// an arithmetic intensity of ~43 FLOP/B: L2-bound.  Here a workgroup owns a CHANNEL SLICE of the
// input (CW = 16 or 32 channels) x a tap group (all taps when the accumulators fit) x all output
// channels, and walks a strided subset of the spatial tiles.  Per tile it stages, once, the input
// patch slice with its halo and the matching dY patch (coalesced 16-B loads, natural NHWC order);
// every tap's k-major MFMA operand is then a SHIFTED WINDOW of the same LDS patch, read transposed
// with ds_read_b64_tr_b16 (pixels are the MFMA K dimension).  The four waves split the taps.
// Accumulators (up to 36 fragments = 144 VGPRs per wave) stay in registers across all of the
// workgroup's tiles and are flushed once with fp32 atomics.
//
// LDS layout: pixel records of PS bytes with PS/32 odd (32-B slices as they are, 64-B slices padded
// to 96; dY 64 -> 96, 128 -> 160, 256 -> 288), and the MFMA K index is mapped to pixels as
// k = 8g + 4h + q  <->  pixel 16h + 4g + q, so the two lane-halves of each transposed read touch
// 8 consecutive pixels on 8 different 32-B bank groups: conflict-free.
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"
#include "tile_stage.hip.h"
#include "conv_geom.h"

#ifndef SV_WT_PF
#define SV_WT_PF 3      // LDS prefetch depth (fragments) of the transposed A-operand reads (re-measured round 2: 3 beats 4 by 0.5 % of the step, 1-2 and 6-8 lose)
#endif

__device__ __forceinline__ short4_t tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

// TPW taps per wave (4 waves: tap group = 4*TPW taps), CIF ci-fragments (16 channels) per
// workgroup slice, COF co-fragments (all of Cout_pad16), KC = 32-pixel K chunks per tile.
// NG wave groups of 4 waves: with NG = 2 the workgroup has two tile buffers, each group walks every other tile of the
// run with the SAME tap split, and group 1's accumulators are added to group 0's through LDS before the flush --
// the same waves per SIMD as two 4-wave workgroups per CU, but half the partial-sum slabs to write and reduce
// (the slab traffic was 25-40 % of these kernels).
template <int TPW, int CIF, int COF, int KC, int NG, int OCC>
__global__ __launch_bounds__(256 * NG, NG == 1 ? OCC : 1) void wgrad_tile_kernel(const WgradTileMulti mg) {
  const WgradTileArgs& g = mg.a[blockIdx.z];      // twin layers (x / x-hat networks) share one launch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int grp = NG == 1 ? 0 : (int)(threadIdx.x >> 8);
  char* sIn = smem + grp * (g.in_bytes + g.dy_bytes);   // [NB][TIH][TIW] pixels of PS bytes (+ slack)
  char* sDy = sIn + g.in_bytes;                         // [BM] pixels of YS bytes (+ slack)
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // thread / wave WITHIN the group
  const int tg = blockIdx.y / g.ncg, cg = blockIdx.y - tg * g.ncg;   // tap group, channel slice
  const int tap0 = tg * (4 * TPW) + wave * TPW;                       // this wave's first tap
  const int ci0 = cg * g.CW;                                          // first input channel of the slice
  const int TW = 1 << g.lTW, TH = 1 << g.lTH, NB = 1 << g.lNB;
  const int cpp = 1 << g.cl2;                                         // 16-B chunks per pixel IN THE SLICE
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3, lr = lane & 15;

  // LDS byte offsets for the transposed reads: read h of chunk kc -> tile pixel r = 32*kc + 16*h + (4*lg + lq),
  // channel block 4*lp (see the K <-> pixel map above).  A tile holds >= 16 pixels per image row group, so the
  // lane part (4*lg + lq < 16) and the chunk part (a multiple of 16) fill disjoint bit fields of (bl, ty, tx) and
  // the offset splits into one per-lane register plus wave-uniform terms (SGPRs): 4*KC fewer VGPRs.
  int in_lane, dy_lane;
  {
    const int r = 4 * lg + lq;
    const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1);
    in_lane = (ty * g.S * g.TIW + tx * g.SX) * g.PS + 4 * lp * 2;
    dy_lane = r * g.YS + 4 * lp * 2;
  }
  auto in_chunk = [&](int kc, int h) -> int {       // uniform
    const int r = kc * 32 + 16 * h;
    const int ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
    return ((bl * g.TIH + ty * g.S) * g.TIW) * g.PS;
  };
  auto dy_chunk = [&](int kc, int h) -> int { return (kc * 32 + 16 * h) * g.YS; };
  int tapoff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = min(tap0 + t, g.ntaps - 1);
    tapoff[t] = (((int)g.dy[tap] - g.y_lo) * g.TIW + ((int)g.dx[tap] - g.x_lo)) * g.PS;
  }

  f32x4 acc[TPW][CIF][COF];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < CIF; ++i)
#pragma unroll
      for (int j = 0; j < COF; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dY = ones^T . dY: one more "tap" whose A operand is all ones, on the dY fragments the loop
  // already holds (wave w takes the output-channel fragments j = w, w + 4, ...: KC * ceil(COF / 4) extra MFMAs per wave and tile and at
  // most two more accumulators -- a full set of COF per wave cost the 8-fragment layers their occupancy --; every row of the result
  // is the column sum).
  // (The first version summed the columns with scalar LDS reads: 5-30 us of the launch -- e1 91 -> 61 us without it -- and only on the
  // channel-slice-0 workgroups, i.e. on the launch's critical path.)
  // Measured per layer (B = 512, serial table): d4 0.180 -> 0.172 ms, d5 0.143 -> 0.136, d3 0.098 -> 0.087 with the MFMA form; e2 0.076 ->
  // 0.106, e1 0.076 -> 0.084, d2 0.079 -> 0.083 (their register budgets are the tight ones): those keep the scalar column sums.
  constexpr bool BIASM = NG == 1 && (OCC == 3 || (TPW == 4 && CIF == 2));       // the d4 / d5 instantiations (three workgroups per CU) and d3's
  constexpr int BJ = (COF + 3) / 4;
  f32x4 bacc[BJ];
  float bsum = 0.f;
  const int bcol = tid & (g.ldy - 1), bgrp = tid / g.ldy, nbg = 256 / g.ldy;
#pragma unroll
  for (int j = 0; j < BJ; ++j) bacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = g.dbias != nullptr && blockIdx.y == 0;
  const int ycols = g.ldy;                          // dY channels per pixel (power of two >= 8)
  const short8_t ones = (short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0

  const bf16_t* __restrict__ Ab = (const bf16_t*)g.A + ci0;
  const bf16_t* __restrict__ Yb = (const bf16_t*)g.dY;
  const int lycp = g.lycp;                          // log2(16-B pieces per dY pixel)
  const int dy_total = (32 * KC) << lycp;

  // each workgroup walks a CONTIGUOUS run of tiles (neighbours share halos: the re-reads hit this XCD's L2)
  const int per_wg = (g.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile_lo = g.contig ? (int)blockIdx.x * per_wg : (int)blockIdx.x;
  const int tile_hi = g.contig ? min(g.ntiles, tile_lo + per_wg) : g.ntiles;
  const int tile_step = g.contig ? 1 : (int)gridDim.x;
  for (int tile0 = tile_lo; tile0 < tile_hi; tile0 += tile_step * NG) {
    const int tile = tile0 + grp * tile_step;
    const bool has = tile < tile_hi;                // the last pass of an odd run leaves group 1 idle
    int t = has ? tile : tile_lo;
    const int tx0 = (t % g.tilesX) << g.lTW; t /= g.tilesX;
    const int ty0 = (t % g.tilesY) << g.lTH; t /= g.tilesY;
    const int b0 = t << g.lNB;
    __syncthreads();                                // previous tile fully consumed
    // ---- stage input patch slice (+halo, zero outside the image; optionally through the fused upsample)
    {
      const TileStageGeom sg = {g.B, g.IH, g.IW, g.lda, g.cl2, g.TIW, g.TIH, g.PS, NB, 0};
      const int iy_base = ty0 * g.S + g.y_lo, ix_base = tx0 * g.SX + g.x_lo;
      if ((SV_DBG(g.dbg) & 2) || !has) {}
      else if (g.ups) stage_tile_upsampled<bf16_t>(Ab, sg, b0, iy_base, ix_base, sIn, tid);
      else if (g.clampin) stage_tile_plain<bf16_t, 256, true>(Ab, sg, b0, iy_base, ix_base, sIn, tid);
      else stage_tile_plain<bf16_t>(Ab, sg, b0, iy_base, ix_base, sIn, tid);
    }
    // ---- stage dY patch
    for (int q = tid; q < dy_total && !(SV_DBG(g.dbg) & 4) && has; q += 256) {
      const int r = q >> lycp, c = q & ((1 << lycp) - 1);
      const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
      const int b = b0 + bl;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b < g.B) {
        if (g.dy_s2d) {   // dy_s2d = channels per hi-res pixel (8 / 32): piece c = (parity (py, px), 16-B piece within the pixel)
          const int pp = g.dy_s2d >> 3, par = c / pp, sub = c - par * pp;
          v = *(const uint4*)(Yb + (((int64_t)b * 2 * g.OY + 2 * (ty0 + ty) + (par >> 1)) * (2 * g.OX) + 2 * (tx0 + tx) + (par & 1)) * g.dy_s2d + sub * 8);
        }
        else v = *(const uint4*)(Yb + ((int64_t)(b * g.OY + ty0 + ty) * g.OX + tx0 + tx) * g.ldy + c * 8);
      }
      *(uint4*)(sDy + r * g.YS + c * 16) = v;
    }
    __syncthreads();
    // ---- MFMA: K = pixels
    if (!(SV_DBG(g.dbg) & 8) && has)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      short8_t bfr[COF];
#pragma unroll
      for (int j = 0; j < COF; ++j) {
        const short4_t lo = tr16(sDy + dy_lane + dy_chunk(kc, 0) + j * 32), hi = tr16(sDy + dy_lane + dy_chunk(kc, 1) + j * 32);
        bfr[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      if (BIASM && do_bias) {
#pragma unroll
        for (int j = 0; j < COF; ++j)
          if ((j & 3) == wave)          // wave-uniform
            bacc[j >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bfr[j]), bacc[j >> 2], 0, 0, 0);
      }
      // A fragments of the (tap, ci-fragment) sequence u = t2*CIF + i, prefetched PF deep so the
      // LDS round trip (~100+ cycles) hides under the MFMAs of earlier fragments
      constexpr int U = TPW * CIF, PF = U < SV_WT_PF ? U : SV_WT_PF;
      short4_t alo[PF], ahi[PF];
      const int ic0 = in_chunk(kc, 0), ic1 = in_chunk(kc, 1);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        alo[u] = tr16(sIn + in_lane + ic0 + tapoff[u / CIF] + (u % CIF) * 32);
        ahi[u] = tr16(sIn + in_lane + ic1 + tapoff[u / CIF] + (u % CIF) * 32);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const short4_t lo = alo[u % PF], hi = ahi[u % PF];
        const short8_t af = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (u + PF < U) {
          alo[u % PF] = tr16(sIn + in_lane + ic0 + tapoff[(u + PF) / CIF] + ((u + PF) % CIF) * 32);
          ahi[u % PF] = tr16(sIn + in_lane + ic1 + tapoff[(u + PF) / CIF] + ((u + PF) % CIF) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch PF fragments ahead of its MFMAs (hipcc sinks it otherwise)
#pragma unroll
        for (int j = 0; j < COF; ++j)
          acc[u / CIF][u % CIF][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[j]), acc[u / CIF][u % CIF][j], 0, 0, 0);
      }
    }
    if (!BIASM && do_bias && has) {
      for (int r = bgrp; r < 32 * KC; r += nbg) bsum += (float)*(const bf16_t*)(sDy + r * g.YS + bcol * 2);
    }
  }

  if constexpr (NG == 2) {
    // group 1 -> group 0 through LDS, half of the fragments at a time (the tile buffers are free now)
    constexpr int NFR_ = TPW * CIF * COF, HF = (NFR_ + 1) / 2;
    float* xch = (float*)smem + (wave * HF * 4) * 64 + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      if (grp == 1) {
#pragma unroll
        for (int f = 0; f < NFR_; ++f)
          if (f / HF == h)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) xch[((f - h * HF) * 4 + r4) * 64] = acc[f / (CIF * COF)][(f / COF) % CIF][f % COF][r4];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int f = 0; f < NFR_; ++f)
          if (f / HF == h)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) acc[f / (CIF * COF)][(f / COF) % CIF][f % COF][r4] += xch[((f - h * HF) * 4 + r4) * 64];
      }
    }
  }
  const bool flusher = grp == 0;
  // ---- flush.  With a partial-sum slab (two-stage, deterministic): every accumulator register goes
  // out as one fully coalesced 256-B store in fragment order; wgrad_reduce_kernel sums the slabs in
  // a fixed order.  (fp32 atomics straight into dW run at ~1.3 TB/s chip-wide and were 40-60 % of
  // this kernel's time: SV_WT_NOFLUSH ablation.)
  if (g.slab) {
    if ((SV_DBG(g.dbg) & 1) || !flusher) goto bias_part;
    {
    float* sl = g.slab + ((((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave) * (TPW * CIF * COF)) * 256 + lane;
#pragma unroll
    for (int t2 = 0; t2 < TPW; ++t2)
#pragma unroll
      for (int i = 0; i < CIF; ++i)
#pragma unroll
        for (int j = 0; j < COF; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) sl[(((t2 * CIF + i) * COF + j) * 4 + r4) * 64] = acc[t2][i][j][r4];
    }
  } else if (flusher)
  // atomics: D row = ci (lane>>4)*4+reg within the fragment, col = co lane&15
#pragma unroll
  for (int t2 = 0; t2 < TPW; ++t2) {
    const int tap = tap0 + t2;
    if (tap >= g.ntaps) continue;
#pragma unroll
    for (int i = 0; i < CIF; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int cl = i * 16 + lg * 4 + r4;
        const int ci = g.pairx ? (cl & 7) : ci0 + cl;
        if (ci >= g.Cin_real || (!g.pairx && cl >= g.CW)) continue;
#pragma unroll
        for (int j = 0; j < COF; ++j) {
          const int co = j * 16 + lr;
          const int otap = g.pairx ? 2 * tap + (cl >> 3) : tap;
          const int64_t di = co < g.N ? dw_index(otap, ci, co, g.Cin_real, g.N, g.fold_kw, g.fold_c) : -1;
          if (di >= 0 && !(SV_DBG(g.dbg) & 1)) atomicAdd(g.dW + di, acc[t2][i][j][r4]);
        }
      }
  }
bias_part:
  if (do_bias) {
    __syncthreads();
    float* red = (float*)smem;
    int nred;
    if constexpr (BIASM) {                          // [NG][ycols]: row 0 of the column-sum fragments (fragment j lives in wave j & 3)
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < COF; ++j)
          if ((j & 3) == wave && j * 16 + lane < ycols) red[grp * ycols + j * 16 + lane] = bacc[j >> 2][0];
      }
      nred = NG;
    } else {
      red[(grp * nbg + bgrp) * ycols + bcol] = bsum;
      nred = nbg * NG;
    }
    __syncthreads();
    if (grp == 0 && tid < ycols && tid < g.N && (!g.fold_kw || (tid & 7) < g.fold_c)) {
      float s = 0.f;
      for (int k = 0; k < nred; ++k) s += red[k * ycols + tid];
      // slab path: one partial per workgroup behind the dW slabs, summed in workgroup order by the reduce kernel (deterministic)
      if (g.bslab) g.bslab[(int64_t)blockIdx.x * 128 + tid] = s;
      else atomicAdd(g.dbias + (g.fold_kw ? (tid & 7) : tid), s);
    }
  }
}

// ---- the same kernel as a PIPELINE, for the layers whose tiles are whole images (d2: 8 x 8 x 128 -> 128, e3: 16 x 16 x 64 -> 8 x 8 x 128) ----
// Phase ablation of the form above on these layers (profiles/r04_abl_wt.txt, 1 024 images): staging 27 us, MFMA loop 20 us, flush + reduce 15 us
// of d2's 93 -- ADDITIVE: the two wave groups of the NG = 2 form stage together, wait together and multiply together, nothing overlaps.  Here
//   * ONE tile at a time for all eight waves: group g (waves 4g .. 4g + 3) multiplies the K chunks kc = g * KC/2 .. of the tile with the same
//     tap split and the same 32 accumulators per wave as before (summed through LDS at the end: the slab order is unchanged);
//   * the tile buffers are a ring of NBUF (2..4, whatever 160 KB holds) and the tiles arrive by global_load_lds_dwordx4 issued NBUF - 1 tiles
//     ahead, counted with s_waitcnt vmcnt: ONE barrier per tile, no VGPR round trip, no address arithmetic in the loop -- every lane's source
//     offsets (relative to the tile's first image) are the same for every tile and computed once (<= 8 registers);
//   * the LDS layout is the one above (pixel records of PS / YS bytes, padded for the transposed reads): a transfer writes 64 consecutive 16-B
//     slots, lanes on padding slots or on pixels outside the image are masked off (the ring is zeroed once: those slots stay zero).
__device__ __forceinline__ void wt_dma16(const void* base, uint32_t off, const char* lds) {    // base: wave-uniform; off: this lane's byte offset
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ void wt_wait_vm(int n) {      // s_waitcnt vmcnt(n), n wave-uniform (the count is an immediate)
  switch (n) {
#define SV_WT_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    SV_WT_VM(1) SV_WT_VM(2) SV_WT_VM(3) SV_WT_VM(4) SV_WT_VM(5) SV_WT_VM(6) SV_WT_VM(7) SV_WT_VM(8) SV_WT_VM(9) SV_WT_VM(10) SV_WT_VM(11) SV_WT_VM(12)
    SV_WT_VM(13) SV_WT_VM(14) SV_WT_VM(15) SV_WT_VM(16) SV_WT_VM(17) SV_WT_VM(18) SV_WT_VM(19) SV_WT_VM(20) SV_WT_VM(21) SV_WT_VM(22) SV_WT_VM(23) SV_WT_VM(24)
#undef SV_WT_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
constexpr int WTP_NI = 10;     // transfers per wave and tile, at most (d3: 34 for the 19 x 19 x 96-B patch + 40 for 256 dY pixels of 160 B, over 8 waves)

template <int TPW, int CIF, int COF, int KC>
__global__ __launch_bounds__(512, 1) void wgrad_tile_pipe_kernel(const WgradTileMulti mg, int nbuf) {
  static_assert(KC % 2 == 0, "the two wave groups split the K chunks of a tile");
  const WgradTileArgs& g = mg.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int stage = g.in_bytes + g.dy_bytes;
  const int grp = (int)(threadIdx.x >> 8), gw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave group, wave of the workgroup (0..7)
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = gw & 3;                                           // thread / wave WITHIN the group
  const int tg = blockIdx.y / g.ncg, cg = blockIdx.y - tg * g.ncg;
  const int tap0 = tg * (4 * TPW) + wave * TPW;
  const int ci0 = cg * g.CW;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH, NB = 1 << g.lNB;
  const int cpp = 1 << g.cl2;
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;

  for (int q = threadIdx.x; q < (nbuf * stage) / 16; q += 512) *(uint4*)(smem + q * 16) = make_uint4(0, 0, 0, 0);

  int in_lane, dy_lane;
  {
    const int r = 4 * lg + lq;
    const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1);
    in_lane = (ty * g.S * g.TIW + tx * g.SX) * g.PS + 4 * lp * 2;
    dy_lane = r * g.YS + 4 * lp * 2;
  }
  auto in_chunk = [&](int kc, int h) -> int {
    const int r = kc * 32 + 16 * h;
    const int ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
    return ((bl * g.TIH + ty * g.S) * g.TIW) * g.PS;
  };
  auto dy_chunk = [&](int kc, int h) -> int { return (kc * 32 + 16 * h) * g.YS; };
  int tapoff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = min(tap0 + t, g.ntaps - 1);
    tapoff[t] = (((int)g.dy[tap] - g.y_lo) * g.TIW + ((int)g.dx[tap] - g.x_lo)) * g.PS;
  }

  // ---- this lane's transfers of a tile: wave-transfer k (64 slots of 16 B) = gw, gw + 8, ...; k < nin: the input patch, else the dY patch
  const int spp_in = g.PS >> 4, spp_dy = g.YS >> 4;                   // 16-B slots per pixel record (padding included)
  const int nin = (NB * g.TIH * g.TIW * spp_in + 63) >> 6, ndy = (32 * KC * spp_dy + 63) >> 6;
  const int ni = (nin + ndy - gw + 7) >> 3;                           // transfers of this wave per tile (wave-uniform, <= WTP_NI: checked on the host)
  uint32_t soff[WTP_NI];                                              // source byte offset from the tile's first image / first dY pixel; ~0: lane off
#pragma unroll
  for (int i = 0; i < WTP_NI; ++i) {
    const int k = gw + 8 * i;
    soff[i] = 0xFFFFFFFFu;
    if (k < nin) {
      const int q = k * 64 + lane, rec = q / spp_in, c = q - rec * spp_in;
      const int per_img = g.TIH * g.TIW, bl = rec / per_img, r2 = rec - bl * per_img, iyl = r2 / g.TIW, ixl = r2 - iyl * g.TIW;
      const int iy = g.y_lo + iyl, ix = g.x_lo + ixl;
      if (bl < NB && c < cpp && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
        soff[i] = (uint32_t)((((bl * g.IH + iy) * g.IW + ix) * g.lda + c * 8) * 2);
    } else if (k < nin + ndy) {
      const int q = (k - nin) * 64 + lane, rec = q / spp_dy, c = q - rec * spp_dy;
      if (rec < 32 * KC && c < (g.ldy >> 3)) soff[i] = (uint32_t)((rec * g.ldy + c * 8) * 2);
    }
  }
  int nact = 0;                                                       // transfers this wave really issues per tile (one with every lane off is
#pragma unroll                                                        // branched over: it must not be counted by the s_waitcnt arithmetic)
  for (int i = 0; i < WTP_NI; ++i)
    if (i < ni && __builtin_amdgcn_ballot_w64(soff[i] != 0xFFFFFFFFu) != 0) ++nact;
  nact = __builtin_amdgcn_readfirstlane(nact);
  const bf16_t* __restrict__ Ab = (const bf16_t*)g.A + ci0;
  const bf16_t* __restrict__ Yb = (const bf16_t*)g.dY;
  const int64_t in_tile = (int64_t)NB * g.IH * g.IW * g.lda, dy_tile = (int64_t)32 * KC * g.ldy;     // elements per tile
  auto issue = [&](int tile, int slot) {
    char* buf = smem + slot * stage;
    const bf16_t* ab = Ab + tile * in_tile;
    const bf16_t* yb = Yb + tile * dy_tile;
#pragma unroll
    for (int i = 0; i < WTP_NI; ++i) {
      const int k = gw + 8 * i;
      if (i < ni) {                                                   // wave-uniform
        const bool isin = k < nin;
        const char* dst = isin ? buf + k * 1024 : buf + g.in_bytes + (k - nin) * 1024;
        const void* base = isin ? (const void*)ab : (const void*)yb;
        if (soff[i] != 0xFFFFFFFFu) wt_dma16(base, soff[i], dst);
      }
    }
  };

  f32x4 acc[TPW][CIF][COF];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < CIF; ++i)
#pragma unroll
      for (int j = 0; j < COF; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ycols = g.ldy;
  auto bias_split = [&]() -> int {
    int bsp = 1;
    while (2 * bsp <= (int)gridDim.y && 16 * bsp <= ycols) bsp *= 2;
    return bsp;
  };
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool do_bias = g.dbias != nullptr && (int)blockIdx.y < bias_split();

  const int per_wg = (g.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile_lo = (int)blockIdx.x * per_wg, tile_hi = min(g.ntiles, tile_lo + per_wg);
  __syncthreads();                                                    // the ring is zero
  for (int i = 0; i < nbuf - 1; ++i)
    if (tile_lo + i < tile_hi) issue(tile_lo + i, i);
  int slot = 0;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    // my transfers of this tile have landed: everything but those of the (at most nbuf - 2) later tiles already issued
    wt_wait_vm(min(nbuf - 2, tile_hi - 1 - tile) * nact);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // ... and everybody's; the previous tile is consumed
    asm volatile("" ::: "memory");
    {
      int ns = slot + nbuf - 1;
      if (ns >= nbuf) ns -= nbuf;
      if (tile + nbuf - 1 < tile_hi) issue(tile + nbuf - 1, ns);      // into the buffer of the previous tile
    }
    const char* sIn = smem + slot * stage;
    const char* sDy = sIn + g.in_bytes;
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      const int kc = grp * (KC / 2) + kk;
      short8_t bfr[COF];
#pragma unroll
      for (int j = 0; j < COF; ++j) {
        const short4_t lo = tr16(sDy + dy_lane + dy_chunk(kc, 0) + j * 32), hi = tr16(sDy + dy_lane + dy_chunk(kc, 1) + j * 32);
        bfr[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      constexpr int U = TPW * CIF, PF = U < SV_WT_PF ? U : SV_WT_PF;
      short4_t alo[PF], ahi[PF];
      const int ic0 = in_chunk(kc, 0), ic1 = in_chunk(kc, 1);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        alo[u] = tr16(sIn + in_lane + ic0 + tapoff[u / CIF] + (u % CIF) * 32);
        ahi[u] = tr16(sIn + in_lane + ic1 + tapoff[u / CIF] + (u % CIF) * 32);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const short4_t lo = alo[u % PF], hi = ahi[u % PF];
        const short8_t af = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (u + PF < U) {
          alo[u % PF] = tr16(sIn + in_lane + ic0 + tapoff[(u + PF) / CIF] + ((u + PF) % CIF) * 32);
          ahi[u % PF] = tr16(sIn + in_lane + ic1 + tapoff[(u + PF) / CIF] + ((u + PF) % CIF) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < COF; ++j)
          acc[u / CIF][u % CIF][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[j]), acc[u / CIF][u % CIF][j], 0, 0, 0);
      }
    }
    if (do_bias) {                                                    // column sums of the dY patch: the first `bias_split()` workgroups of a grid column (they stage the
                                                                      // same dY tiles) take ycols / split columns each, a thread sums a column PAIR (4-B reads) over every
                                                                      // (512 / pairs)-th pixel
      const int bpc = (ycols >> 1) / bias_split();
      const int t5 = threadIdx.x, bch = t5 & (bpc - 1), brow = t5 / bpc, nbrow = 512 / bpc, bc0 = (int)blockIdx.y * bpc;
      for (int r = brow; r < 32 * KC; r += nbrow) {
        const uint32_t v = *(const uint32_t*)(sDy + r * g.YS + (bc0 + bch) * 4);
        bsum0 += __uint_as_float(v << 16);
        bsum1 += __uint_as_float(v & 0xFFFF0000u);
      }
    }
    if (++slot == nbuf) slot = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- group 1 -> group 0 through LDS, half of the fragments at a time; flush in the slab order of the form above
  {
    constexpr int NFR_ = TPW * CIF * COF, HF = (NFR_ + 1) / 2;
    float* xch = (float*)smem + (wave * HF * 4) * 64 + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      if (grp == 1) {
#pragma unroll
        for (int f = 0; f < NFR_; ++f)
          if (f / HF == h)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) xch[((f - h * HF) * 4 + r4) * 64] = acc[f / (CIF * COF)][(f / COF) % CIF][f % COF][r4];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int f = 0; f < NFR_; ++f)
          if (f / HF == h)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) acc[f / (CIF * COF)][(f / COF) % CIF][f % COF][r4] += xch[((f - h * HF) * 4 + r4) * 64];
      }
    }
  }
  if (grp == 0) {
    float* sl = g.slab + ((((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave) * (TPW * CIF * COF)) * 256 + lane;
#pragma unroll
    for (int t2 = 0; t2 < TPW; ++t2)
#pragma unroll
      for (int i = 0; i < CIF; ++i)
#pragma unroll
        for (int j = 0; j < COF; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) sl[(((t2 * CIF + i) * COF + j) * 4 + r4) * 64] = acc[t2][i][j][r4];
  }
  if (do_bias) {
    __syncthreads();
    float* red = (float*)smem;                                        // [512 / bpc][2 * bpc]
    const int bpc = (ycols >> 1) / bias_split();
    const int t5 = threadIdx.x, bch = t5 & (bpc - 1), brow = t5 / bpc, nbrow = 512 / bpc, bc0 = (int)blockIdx.y * bpc;
    red[(brow * bpc + bch) * 2] = bsum0;
    red[(brow * bpc + bch) * 2 + 1] = bsum1;
    __syncthreads();
    const int bw = 2 * bpc, col = 2 * bc0 + t5;
    if (t5 < bw && col < g.N) {
      float sum = 0.f;
      for (int k = 0; k < nbrow; ++k) sum += red[k * bw + t5];
      if (g.bslab) g.bslab[(int64_t)blockIdx.x * 128 + col] = sum;
      else atomicAdd(g.dbias + col, sum);
    }
  }
}

// second stage of the slab path: dW[...] += sum over the m-splits (fixed order: deterministic).
// A block sums 32 float4 columns; its 8 thread rows take the splits x = row, row+8, ... with 16-B
// loads (many in flight: the first version, one scalar column per thread, ran at 1.2 TB/s) and are
// combined through LDS in row order.
template <int TPW, int CIF, int COF>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceMulti m, int msplit, int groups, int ncg,
                                                           int CW, int Cin_real, int N, int ntaps, int fold_kw,
                                                           int fold_c, int pairx, int assign) {
  const float* __restrict__ slab = m.slab[blockIdx.z];
  float* __restrict__ dW = m.dW[blockIdx.z];
  __shared__ float4 part[8][32];
  if (blockIdx.x == 0 && blockIdx.y == 0 && m.bslab[blockIdx.z]) {
    // bias gradient: the workgroups' partial column sums [msplit][128], always in the same order: like the dW slabs below, eight thread
    // rows take the splits x = row, row + 8, ... with 16-B loads and are combined through LDS in row order
    const float* __restrict__ bs = m.bslab[blockIdx.z];
    const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int x = row; x < msplit; x += 8) {
      const float4 v = *(const float4*)(bs + x * 128 + col * 4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    part[row][col] = s4;
    __syncthreads();
    if (!row) {
#pragma unroll
      for (int r = 1; r < 8; ++r) { const float4 v = part[r][col]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
      part[0][col] = s4;
    }
    __syncthreads();
    const float* tot = (const float*)&part[0][0];             // [128] column sums
    const int c = threadIdx.x;
    if (c < N && c < 128 && (!fold_kw || c < fold_c)) m.dbias[blockIdx.z][c] += tot[c] + (fold_kw ? tot[c + 8] : 0.f);
    __syncthreads();
  }
  constexpr int NFR = TPW * CIF * COF, PER = 4 * NFR * 256;      // floats per (split, group)
  const int y = blockIdx.y, col = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int e = (blockIdx.x * 32 + col) * 4;                     // first of this thread's 4 floats (PER % 128 == 0)
  const float* p = slab + (int64_t)y * PER + e;
  const int64_t xs = (int64_t)groups * PER;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int x = row; x < msplit; x += 8) {
    const float4 v = *(const float4*)(p + x * xs);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[row][col] = s;
  __syncthreads();
  if (row) return;
#pragma unroll
  for (int r = 1; r < 8; ++r) {
    const float4 v = part[r][col];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  // e = ((wave*NFR + f)*4 + r4)*64 + lane: four consecutive lanes = four consecutive output channels
  const int wave = e / (NFR * 256), rem = e - wave * (NFR * 256), f = rem >> 8, r4 = (rem >> 6) & 3, lane = rem & 63;
  const int t2 = f / (CIF * COF), i = (f / COF) % CIF, j = f % COF;
  const int tg = y / ncg, cg = y - tg * ncg;
  const int tap = tg * 4 * TPW + wave * TPW + t2;
  const int cl = i * 16 + (lane >> 4) * 4 + r4, ci = pairx ? (cl & 7) : cg * CW + cl, co = j * 16 + (lane & 15);
  if (tap >= ntaps || (!pairx && cl >= CW) || ci >= Cin_real) return;
  const int otap = pairx ? 2 * tap + (cl >> 3) : tap;
  const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (co + k >= N) continue;
    const int64_t di = dw_index(otap, ci, co + k, Cin_real, N, fold_kw, fold_c);
    if (di < 0) continue;
    // folded: the two pixel-parity columns of a tap pair land on one element from two threads; two
    // atomic adds onto the zeroed gradient commute exactly, so the result is still run-to-run identical
    if (fold_kw) atomicAdd(dW + di, sv[k]); else if (assign) dW[di] = sv[k]; else dW[di] += sv[k];
  }
}

// The same reduce for ANY layer with its fragment counts as run-time values: all the pending reduces of a backward pass in one launch
// (blockIdx.z = descriptor).  At 64 images per GPU the seven per-layer reduce launches were ~11 us each for a few hundred KB.
__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(const WgradReduceAll m) {
  int z = 0;                               // flat grid: descriptor z owns blocks [first[z], first[z+1]) (no empty workgroups: a padded 3-D grid of
  while (z + 1 < m.n && (int)blockIdx.x >= m.first[z + 1]) ++z;      // 60 k early-exit workgroups cost 60 us by itself)
  const WgradReduceDesc& g = m.d[z];
  const int NFR = g.TPW * g.CIF * g.COF, PER = 4 * NFR * 256;
  const int lb = blockIdx.x - m.first[z], bpg = PER / 128;
  const int by = lb / bpg, bx = lb - by * bpg;
  __shared__ float4 part[8][32];
  const int msplit = g.msplit, fold_kw = g.fold_kw, N = g.N;
  if (bx == 0 && by == 0 && g.bslab) {
    const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int x = row; x < msplit; x += 8) {
      const float4 v = *(const float4*)(g.bslab + x * 128 + col * 4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    part[row][col] = s4;
    __syncthreads();
    if (!row) {
#pragma unroll
      for (int r = 1; r < 8; ++r) { const float4 v = part[r][col]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
      part[0][col] = s4;
    }
    __syncthreads();
    const float* tot = (const float*)&part[0][0];
    const int c = threadIdx.x;
    if (c < N && c < 128 && (!fold_kw || c < g.fold_c)) g.dbias[c] += tot[c] + (fold_kw ? tot[c + 8] : 0.f);
    __syncthreads();
  }
  const int y = by, col = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int e = (bx * 32 + col) * 4;
  const float* p = g.slab + (int64_t)y * PER + e;
  const int64_t xs = (int64_t)g.groups * PER;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int x = row; x < msplit; x += 8) {
    const float4 v = *(const float4*)(p + x * xs);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[row][col] = s;
  __syncthreads();
  if (row) return;
#pragma unroll
  for (int r = 1; r < 8; ++r) {
    const float4 v = part[r][col];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int wave = e / (NFR * 256), rem = e - wave * (NFR * 256), f = rem >> 8, r4 = (rem >> 6) & 3, lane = rem & 63;
  const int t2 = f / (g.CIF * g.COF), i = (f / g.COF) % g.CIF, j = f % g.COF;
  const int tg = y / g.ncg, cg = y - tg * g.ncg;
  const int tap = tg * 4 * g.TPW + wave * g.TPW + t2;
  const int cl = i * 16 + (lane >> 4) * 4 + r4, ci = g.pairx ? (cl & 7) : cg * g.CW + cl, co = j * 16 + (lane & 15);
  if (tap >= g.ntaps || (!g.pairx && cl >= g.CW) || ci >= g.Cin_real) return;
  const int otap = g.pairx ? 2 * tap + (cl >> 3) : tap;
  const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (co + k >= N) continue;
    const int64_t di = dw_index(otap, ci, co + k, g.Cin_real, N, fold_kw, g.fold_c, g.s2d3);
    if (di < 0) continue;
    if (fold_kw) atomicAdd(g.dW + di, sv[k]); else if (g.assign) g.dW[di] = sv[k]; else g.dW[di] += sv[k];
  }
}

int svk_wgrad_reduce_all(const WgradReduceDesc* d, int n, hipStream_t st) {
  for (int b = 0; b < n; b += SV_WGRAD_DEFER_MAX) {
    WgradReduceAll m;
    m.n = n - b < SV_WGRAD_DEFER_MAX ? n - b : SV_WGRAD_DEFER_MAX;
    int total = 0;
    for (int i = 0; i < m.n; ++i) {
      m.d[i] = d[b + i];
      const int per = 4 * d[b + i].TPW * d[b + i].CIF * d[b + i].COF * 256;
      m.first[i] = total;
      total += (per / 128) * d[b + i].groups;
    }
    m.first[m.n] = total;
    hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(total), dim3(256), 0, st, m);
    SV_LAUNCH_CHECK();
  }
  return SV_OK;
}

template <int TPW, int CIF, int COF, int KC, int NG, int OCC = 2>
static int launch_wt_ng(const WgradTileArgs* a, int n, int groups, hipStream_t st, const hipEvent_t* ev_mid) {
  constexpr int HFB = ((TPW * CIF * COF + 1) / 2) * 4 * 1024;     // bytes of the cross-group exchange (NG = 2)
  size_t lds = NG * ((size_t)a[0].in_bytes + a[0].dy_bytes);
  if (NG == 2 && lds < (size_t)HFB) lds = HFB;
  sv_ensure_dynamic_lds((const void*)wgrad_tile_kernel<TPW, CIF, COF, KC, NG, OCC>, lds);
  // resident workgroups per CU: LDS, and 2 waves per SIMD (accumulator-heavy waves)
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > OCC / NG) per_cu = OCC / NG;
  if (per_cu < 1) per_cu = 1;
  static const int force_pc = getenv("SV_WT_PERCU") ? atoi(getenv("SV_WT_PERCU")) : 0;   // profiling knob
  if (force_pc > 0 && force_pc < per_cu) per_cu = force_pc;
  // ablation bits: 1 skip the flush (+reduce), 2 skip input staging, 4 skip dY staging, 8 skip the MFMA loop
  static const int dbg = SV_DBG(getenv("SV_WT_DBG") ? atoi(getenv("SV_WT_DBG")) : (getenv("SV_WT_NOFLUSH") ? 1 : 0));
  // ONE resident round of workgroups over the whole launch (all n problems): every workgroup flushes one slab, so
  // a second round doubles the slab writes and the reduce's reads for no extra parallelism (SV_WT_ROUNDS=2: the old
  // one round PER PROBLEM, for A/B)
  static const int rounds = getenv("SV_WT_ROUNDS") ? atoi(getenv("SV_WT_ROUNDS")) : 1;
  const int share = rounds >= 2 ? 1 : n;
  int msplit = (256 * per_cu + groups * share - 1) / (groups * share);
  if (msplit < 1) msplit = 1;
  if (msplit > a[0].ntiles) msplit = a[0].ntiles;
  // small batches: a workgroup should own several tiles before it pays for a slab flush (the slab is as large for one tile as
  // for forty: at 64 images per network d2's 512 one-tile workgroups wrote and re-read 67 MB of partial sums)
  static const int min_tiles = getenv("SV_WT_MIN_TILES") ? atoi(getenv("SV_WT_MIN_TILES")) : 4;   // B = 64: 0.768 -> 0.746 ms; 8: worse (too few workgroups)
  if (min_tiles > 1) {
    int cand = (a[0].ntiles + min_tiles - 1) / min_tiles;
    const int floor_wgs = (256 + groups * n - 1) / (groups * n);     // ... but never fewer than one workgroup per CU in the launch
    if (cand < floor_wgs) cand = floor_wgs;                          // (SVHN-32, 64 images: 4 tiles per workgroup left 32 workgroups: +5 %)
    if (msplit > cand) msplit = cand;
  }
  dim3 grid(msplit, groups, n), block(256 * NG);
  constexpr int PER = 4 * TPW * CIF * COF * 256;
  const int64_t need = (int64_t)msplit * groups * PER * 4 + (int64_t)msplit * 128 * 4;      // + one (<= 128-column) bias partial per workgroup row
  static const bool no_slab = getenv("SV_WT_ATOMICS") != nullptr;
  WgradTileMulti m;
  WgradReduceMulti r;
  bool slab = !no_slab;
  for (int i = 0; i < n; ++i) slab = slab && a[i].ws && a[i].ws_bytes >= need;
  for (int i = 0; i < n; ++i) {
    m.a[i] = a[i];
    m.a[i].dbg = dbg;
    m.a[i].slab = slab ? a[i].ws : nullptr;
    m.a[i].bslab = (slab && a[i].dbias && a[i].ldy <= 128) ? a[i].ws + (int64_t)msplit * groups * PER : nullptr;
    r.slab[i] = m.a[i].slab; r.dW[i] = a[i].dW; r.bslab[i] = m.a[i].bslab; r.dbias[i] = a[i].dbias;
  }
  bool piped = false;
  if constexpr (NG == 2 && KC % 2 == 0) {
    // whole-image tiles, plain staging, slabs: the pipelined form (ring of 2..4 tile buffers filled by LDS-DMA)
    static const bool no_pipe = getenv("SV_WT_NO_PIPE") != nullptr;
    static const int force_nbuf = getenv("SV_WT_PIPE_NBUF") ? atoi(getenv("SV_WT_PIPE_NBUF")) : 0;
    const WgradTileArgs& q = a[0];
    const size_t stage = (size_t)q.in_bytes + q.dy_bytes;
    int nbuf = (int)((160 * 1024) / stage);
    if (nbuf > 4) nbuf = 4;
    if (force_nbuf >= 2 && force_nbuf < nbuf) nbuf = force_nbuf;
    const int NBi = 1 << q.lNB;
    const int nin = (NBi * q.TIH * q.TIW * (q.PS >> 4) + 63) >> 6, ndy = (32 * KC * (q.YS >> 4) + 63) >> 6;
    if (!no_pipe && slab && nbuf >= 2 && q.tilesX == 1 && q.tilesY == 1 && !q.ups && !q.clampin && !q.dy_s2d && !q.pairx && !q.fold_kw && q.contig &&
        q.B % NBi == 0 && nin + ndy <= 8 * WTP_NI && (q.PS & 15) == 0 && (q.YS & 15) == 0 && (size_t)HFB <= nbuf * stage) {
      const size_t plds = nbuf * stage;
      sv_ensure_dynamic_lds((const void*)wgrad_tile_pipe_kernel<TPW, CIF, COF, KC>, plds);
      hipLaunchKernelGGL((wgrad_tile_pipe_kernel<TPW, CIF, COF, KC>), grid, block, plds, st, m, nbuf);
      piped = true;
    }
  }
  if (!piped) hipLaunchKernelGGL((wgrad_tile_kernel<TPW, CIF, COF, KC, NG, OCC>), grid, block, lds, st, m);
  SV_LAUNCH_CHECK();
  if (ev_mid && ev_mid[0]) { (void)hipEventRecord(ev_mid[0], st); (void)hipEventRecord(ev_mid[1], st); }
  if (slab && !(dbg & 1) && a[0].defer && a[0].n_defer && *a[0].n_defer + n <= 64) {      // the caller reduces every layer's slabs in one launch later
    for (int i = 0; i < n; ++i)
      a[0].defer[(*a[0].n_defer)++] = WgradReduceDesc{m.a[i].slab, a[i].dW, m.a[i].bslab, a[i].dbias, msplit, groups, a[0].ncg, a[0].CW, a[0].Cin_real,
                                                      a[0].N, a[0].ntaps, a[0].fold_kw, a[0].fold_c, a[0].pairx, a[0].assign, TPW, CIF, COF};
    return SV_OK;
  }
  if (slab && !(dbg & 1)) {
    hipLaunchKernelGGL((wgrad_reduce_kernel<TPW, CIF, COF>), dim3(PER / 128, groups, n), dim3(256), 0, st, r, msplit, groups,
                       a[0].ncg, a[0].CW, a[0].Cin_real, a[0].N, a[0].ntaps, a[0].fold_kw, a[0].fold_c, a[0].pairx, a[0].assign);
    SV_LAUNCH_CHECK();
  }
  return SV_OK;
}

template <int TPW, int CIF, int COF, int KC>
static int launch_wt(const WgradTileArgs* a, int n, int groups, hipStream_t st, const hipEvent_t* ev_mid) {
  static const bool ng1 = getenv("SV_WT_NG1") != nullptr;      // A/B knob: 4-wave workgroups, two per CU
  constexpr size_t HFB = ((TPW * CIF * COF + 1) / 2) * 4 * 1024;
  const size_t two = 2 * ((size_t)a[0].in_bytes + a[0].dy_bytes);
  const bool slab = a[0].ws != nullptr;                         // halving the slabs is the point; atomics keep NG = 1
  // measured per layer (B = 512): e2 -13 %, d2 -8 %, e1 -4 %, d3 0; d4 / d5 (the longest MFMA sections) lose
  // 5-12 % to the lockstep of the two groups, so the wide-tile layers keep two independent workgroups per CU
  static const char* ng2 = getenv("SV_WT_NG2") ? getenv("SV_WT_NG2") : "3456";     // layer ids (see the table above)
  const bool want = strchr(ng2, '0' + a[0].layer_id) != nullptr || (a[0].layer_id == 2 && !a[0].ups);    // d3 on its written-out resized input: the pipelined form
  if (!ng1 && want && slab && two <= 160 * 1024 && HFB <= 160 * 1024 && a[0].ntiles >= 64)
    return launch_wt_ng<TPW, CIF, COF, KC, 2>(a, n, groups, st, ev_mid);
  return launch_wt_ng<TPW, CIF, COF, KC, 1>(a, n, groups, st, ev_mid);
}

// Returns SV_E_UNSUPPORTED when the layer shape has no tile instantiation (caller falls back to
// the im2col wgrad).  Shapes: the seven conv layers of the SPLIT-VAE encoder/decoder.
int svk_wgrad_tile_multi(const WgradArgs* wv, int n, hipStream_t st) {
  if (n < 1 || n > SV_WGRAD_MAX_MULTI) return SV_E_BADARG;
  const WgradArgs& w = wv[0];                       // the n problems share one geometry, pointers differ
  static const bool force_old = getenv("SV_FORCE_IM2COL") != nullptr;
  static const char* skip = getenv("SV_WGRAD_IM2COL_IDS");    // e.g. "23": these layer ids use the im2col kernel (A/B)
  if (force_old || w.lOY < 0 || w.lOX < 0 || w.S > 2) return SV_E_UNSUPPORTED;   // power-of-two grids, stride <= 2
  if (svk_wgrad_roll_supported(wv, n)) return svk_wgrad_roll_multi(wv, n, st);
  if (svk_wgrad_p5_supported(wv, n)) return svk_wgrad_p5_multi(wv, n, st);
  if (svk_wgrad_e1_supported(wv, n)) return svk_wgrad_e1_multi(wv, n, st);
  if (svk_wgrad_e2_supported(wv, n)) return svk_wgrad_e2_multi(wv, n, st);
  const int OY = 1 << w.lOY, OX = 1 << w.lOX;
  if (OY * OX < 16 || w.ycols != w.ldy || (w.ups && w.S != 1)) return SV_E_UNSUPPORTED;
  const int cin = w.Cin_pad, cout = w.ldy, nt = w.ntaps;
  // id, pixels per tile, channel slice width, taps per group
  int id = -1, BM = 0, CW = 0, TT = 0;
  if (nt == 36 && cin == 32 && cout == 8) { id = 0; BM = 256; CW = 32; TT = 36; }          // d5
  else if (nt == 36 && cin == 64 && cout == 32) { id = 1; BM = 256; CW = 32; TT = 36; }    // d4
  else if (nt == 16 && cin == 128 && cout == 64) { id = 2; BM = 256; CW = 32; TT = 16; }   // d3
  else if (nt == 16 && cin == 128 && cout == 128) { id = 3; BM = 128; CW = 16; TT = 16; }  // d2
  else if (nt == 16 && cin == 64 && cout == 128) { id = 4; BM = 128; CW = 16; TT = 16; }   // e3
  else if (nt == 36 && cin == 32 && cout == 64) { id = 5; BM = 128; CW = 16; TT = 36; }    // e2
  else if (nt == 36 && cin == 8 && cout == 32) { id = 6; BM = 256; CW = 8; TT = 36; }      // e1 (taps halved below: pairx)
  else if (nt == 42 && cin == 32 && cout == 16 && w.fold_kw) { id = 7; BM = 256; CW = 32; TT = 42; }   // d5, x-packed
  else if (nt == 25 && cin == 32 && cout == 32 && w.dy_s2d) { id = 8; BM = 256; CW = 16; TT = 25; }     // d5, polyphase (low-res grid)
  else return SV_E_UNSUPPORTED;
  if (skip && strchr(skip, '0' + id)) return SV_E_UNSUPPORTED;
  // 16-channel slices for d4 / d3 / packed d5: half the accumulators per wave, so three workgroups (3 waves per
  // SIMD) fit per CU instead of two; e1 fits four.  Measured together: 185.5k -> 188.7k images/s (SV_WT_CW16= /
  // SV_WT_HIOCC= with an empty list restore the wide variants).
  static const char* cw16 = getenv("SV_WT_CW16") ? getenv("SV_WT_CW16") : "17";   // (round 2: d3 (id 2) back on 32-channel slices, step -1.3 %)
  const bool narrow = strchr(cw16, '0' + id) && (id == 1 || id == 2 || id == 7);
  static const char* hiocc = getenv("SV_WT_HIOCC") ? getenv("SV_WT_HIOCC") : "6";
  const bool hi = strchr(hiocc, '0' + id) != nullptr;
  if (narrow) CW = 16;
  // (e3 stayed on the im2col GEMM in round 1 -- "no faster here"; with one resident round of workgroups per launch and the
  // two-group form it is: 0.097 -> 0.064 ms incl. the reduce, B = 512.  SV_WGRAD_IM2COL_IDS=4 restores the GEMM.)
  // every layer takes the two-stage flush when a workspace is given: even d5 (6 of 16 slab columns
  // real) and e1 (3 of 16 rows real) beat the fp32 atomics (d5: 183 -> 133 us) since the reduce runs at HBM speed
  const bool allow_slab = true;
  while (OY * OX < BM && (BM % (OY * OX))) BM >>= 1;
  if (BM < 64) return SV_E_UNSUPPORTED;
  int y_lo = 127, y_hi = -127, x_lo = 127, x_hi = -127;
  for (int i = 0; i < nt; ++i) {
    y_lo = w.dy[i] < y_lo ? w.dy[i] : y_lo; y_hi = w.dy[i] > y_hi ? w.dy[i] : y_hi;
    x_lo = w.dx[i] < x_lo ? w.dx[i] : x_lo; x_hi = w.dx[i] > x_hi ? w.dx[i] : x_hi;
  }
  const int lTW = OX >= 16 ? 4 : w.lOX;
  int lTH = 0;
  while ((1 << (lTW + lTH)) < BM && (1 << lTH) < OY) ++lTH;
  int lNB = 0;
  while ((1 << (lTW + lTH + lNB)) < BM) ++lNB;
  const int TW = 1 << lTW, TH = 1 << lTH, NB = 1 << lNB;
  const int B = w.M >> (w.lOY + w.lOX);
  WgradTileArgs av[SV_WGRAD_MAX_MULTI];
  WgradTileArgs& a = av[0];
  memset(&a, 0, sizeof(a));
  a.B = B; a.IH = w.IH; a.IW = w.IW; a.lda = w.lda; a.S = w.S; a.SX = w.SX; a.ups = w.ups;
  a.fold_kw = w.fold_kw; a.fold_c = w.fold_c; a.layer_id = id;
  a.clampin = w.clampin; a.dy_s2d = w.dy_s2d; a.assign = w.assign;
  // tile walk of a workgroup: a contiguous run (neighbours share halos in one XCD's L2) or strided by the grid.  Re-measured per
  // layer (round 2): strided wins for the layers listed in SV_WT_STRIDED_IDS (SV_WT_STRIDED=1: every layer)
  static const bool all_strided = getenv("SV_WT_STRIDED") != nullptr;
  static const char* strided_ids = getenv("SV_WT_STRIDED_IDS") ? getenv("SV_WT_STRIDED_IDS") : "567";   // d5 (x-packed), e2, e1: -0.6 % of the step (almost all of it d5)
  a.contig = (all_strided || strchr(strided_ids, '0' + id)) ? 0 : 1;
  a.CW = CW; a.ncg = cin / CW;
  a.cl2 = ilog2_exact(CW / 8);
  a.lTW = lTW; a.lTH = lTH; a.lNB = lNB; a.OY = OY; a.OX = OX;
  a.tilesX = OX / TW; a.tilesY = OY / TH;
  a.ntiles = a.tilesX * a.tilesY * ((B + NB - 1) / NB);
  a.TIW = (TW - 1) * w.SX + (x_hi - x_lo) + 1; a.TIH = (TH - 1) * w.S + (y_hi - y_lo) + 1;
  a.y_lo = y_lo; a.x_lo = x_lo;
  // 16 / 32 / 96 B: PS/32 odd (or a single 16-B chunk); at x stride 2 the K pixels are 2*PS apart: 80 B
  // SV_WT_PAD16: 16-channel slices at x stride 2 (packed d5) on a 96-B K-pixel pitch.  Measured: LDS bank conflicts
  // 46 % -> 1.3 %, LDS-active cycles -45 %, kernel time unchanged (110 us): the loop is not LDS-throughput-bound.
  static const bool pad16 = getenv("SV_WT_PAD16") != nullptr;
  a.PS = CW * 2 + (CW == 32 ? (w.SX == 2 ? 16 : 32) : (CW == 16 && w.SX == 2 && pad16) ? 16 : 0);
  a.ldy = cout; a.YS = cout * 2 + (cout >= 32 ? 32 : 0);
  a.lycp = ilog2_exact(cout / 8);
  a.in_bytes = (NB * a.TIH * a.TIW * a.PS + 64 + 15) / 16 * 16;   // slack: 16-column transposed reads of narrow pixels
  a.dy_bytes = ((32 * (BM / 32)) * a.YS + 64 + 15) / 16 * 16;
  if (a.in_bytes + a.dy_bytes > 150 * 1024) return SV_E_UNSUPPORTED;
  a.Cin_real = w.Cin_real; a.N = w.N; a.ntaps = nt;
  memcpy(a.dy, w.dy, sizeof(a.dy));
  memcpy(a.dx, w.dx, sizeof(a.dx));
  static const bool no_pairx = getenv("SV_WT_NO_PAIRX") != nullptr;   // A/B knob
  const bool pairx = id == 6 && !no_pairx;
  if (pairx) {
    // 8-channel pixel records: fragment rows 8..15 (channel block 8..15 of the transposed read) are the next
    // pixel in x, i.e. the operand of tap (ky, kx+1) -- one MFMA serves two taps.  Keep the even-kx taps
    // (taps are ky-major, KW = 6 even, so original tap = 2 * kept index + (row >> 3)).
    for (int u = 0; u < nt / 2; ++u) { a.dy[u] = w.dy[2 * u]; a.dx[u] = w.dx[2 * u]; }
    a.ntaps = nt / 2;
    a.pairx = 1;
    TT = nt / 2;
  }
  const int groups = a.ncg * ((a.ntaps + TT - 1) / TT);
  const int KC = BM / 32;
  for (int i = n - 1; i >= 0; --i) {
    av[i] = a;
    av[i].A = wv[i].A; av[i].dY = wv[i].dY; av[i].dW = wv[i].dW; av[i].dbias = wv[i].dbias;
    av[i].ws = allow_slab ? wv[i].ws : nullptr; av[i].ws_bytes = allow_slab ? wv[i].ws_bytes : 0;
    av[i].defer = wv[i].defer; av[i].n_defer = wv[i].n_defer;
  }
  // <TPW, CIF, COF, KC>
  switch (id) {
    case 0: if (KC == 8) return launch_wt<9, 2, 1, 8>(av, n, groups, st, wv[0].ev_mid); break;
    case 1: if (KC == 8 && narrow) return launch_wt_ng<9, 1, 2, 8, 1, 3>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 8) return launch_wt<9, 2, 2, 8>(av, n, groups, st, wv[0].ev_mid); break;
    case 2: if (KC == 8 && narrow) return launch_wt_ng<4, 1, 4, 8, 1, 3>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 8) return launch_wt<4, 2, 4, 8>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 2) return launch_wt<4, 2, 4, 2>(av, n, groups, st, wv[0].ev_mid); break;
    case 3: if (KC == 4) return launch_wt<4, 1, 8, 4>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 2) return launch_wt<4, 1, 8, 2>(av, n, groups, st, wv[0].ev_mid); break;
    case 4: if (KC == 4) return launch_wt<4, 1, 8, 4>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 2) return launch_wt<4, 1, 8, 2>(av, n, groups, st, wv[0].ev_mid); break;
    case 5: if (KC == 4) return launch_wt<9, 1, 4, 4>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 2) return launch_wt<9, 1, 4, 2>(av, n, groups, st, wv[0].ev_mid); break;
    case 6: if (KC == 8 && pairx && hi) return launch_wt_ng<5, 1, 2, 8, 1, 4>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 8 && pairx) return launch_wt<5, 1, 2, 8>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 8) return launch_wt<9, 1, 2, 8>(av, n, groups, st, wv[0].ev_mid); break;
    case 8: if (KC == 8) return launch_wt_ng<7, 1, 2, 8, 1, 3>(av, n, groups, st, wv[0].ev_mid); break;
    case 7: if (KC == 8 && narrow) return launch_wt_ng<11, 1, 1, 8, 1, 3>(av, n, groups, st, wv[0].ev_mid);
            if (KC == 8) return launch_wt<11, 2, 1, 8>(av, n, groups, st, wv[0].ev_mid); break;
  }
  return SV_E_UNSUPPORTED;
}

int svk_wgrad_tile(const WgradArgs& w, hipStream_t st) { return svk_wgrad_tile_multi(&w, 1, st); }

int svk_wgrad_dispatch_multi(const WgradArgs* w, int n, int dtype, int cfg, hipStream_t st) {
  static const bool no_multi = getenv("SV_NO_MULTI") != nullptr;
  if (dtype == SV_BF16 && !no_multi && n <= SV_WGRAD_MAX_MULTI) {
    const int rc = svk_wgrad_tile_multi(w, n, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  if (dtype == SV_F32 && !no_multi && n <= SV_WGRAD_MAX_MULTI) {
    const int rc = svk_wgrad_tile_f32_multi(w, n, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  // im2col kernel (no tile instantiation for this shape, or fp32): all of them in one launch
  for (int i = 0; i < n; ++i)
    if (w[i].s2d3 || w[i].clampin || w[i].dy_os || w[i].dy_s2d) return SV_E_UNSUPPORTED;      // views only the tile kernels form
  const bool any_tile = no_multi || n > SV_WGRAD_IM2COL_MAX_MULTI;
  int rc = SV_OK;
  if (!any_tile) {
    WgradArgs wi[SV_WGRAD_IM2COL_MAX_MULTI];
    for (int i = 0; i < n; ++i) {
      wi[i] = w[i];
      wi[i].ev_mid[0] = wi[i].ev_mid[1] = nullptr;
      if (n > 1) svg_wgrad_set_msplit(&wi[i], cfg, dtype, 512 / n);   // the problems share the chip
    }
    rc = svk_wgrad_multi(wi, n, dtype, cfg, st);
  } else {
    for (int i = 0; i < n && !rc; ++i) {
      WgradArgs wi = w[i];
      wi.ev_mid[0] = wi.ev_mid[1] = nullptr;
      rc = svk_wgrad_dispatch(wi, dtype, cfg, st);
    }
  }
  if (rc) return rc;
  if (w[0].ev_mid[0]) { (void)hipEventRecord(w[0].ev_mid[0], st); (void)hipEventRecord(w[0].ev_mid[1], st); }   // no second stage on this path
  return SV_OK;
}

int svk_wgrad_dispatch(const WgradArgs& w, int dtype, int cfg, hipStream_t st) {
  if (dtype == SV_BF16) {
    const int rc = svk_wgrad_tile(w, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  if (dtype == SV_F32) {
    const int rc = svk_wgrad_tile_f32_multi(&w, 1, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  if (w.ups || w.fold_kw || w.s2d3 || w.clampin || w.dy_os || w.dy_s2d) return SV_E_UNSUPPORTED;    // the im2col kernel needs the materialised hi-res tensor / cannot fold / reads no views
  return svk_wgrad(w, dtype, cfg, st);
}
