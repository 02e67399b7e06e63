// Weight-stationary stride-1 NHWC convolution, gfx950: the decoder stack (d3 / d4 / d5 forward and input gradients, d2 forward) and -- as
// stride-1 problems in disguise -- the encoder's stride-2 layers e1 / e2 forward (space-to-depth view, RowCfg::S2D) and e2's input gradient
// (merged parity classes, RowCfg::CLS); 8-pixel-wide images go two to a strip (RowCfg::PAIR).  DESIGN.md 4h.
//
// tile_conv.hip re-streams the weight tile of every K step through LDS for every 256-pixel tile and runs its phases
// (stage -> K loop -> store) one after the other inside a workgroup.  Here the roles are turned around:
//
//   * WEIGHTS LIVE IN REGISTERS.  A wave owns a 16-channel output block and (all of, or half of) K: its share of the
//     prepared weight image [Cout][tap][Cin] is loaded ONCE per workgroup into <= 144 VGPRs as MFMA A-operands
//     (v_mfma_f32_16x16x32_bf16, D = W x pixels) and stays there for every image the workgroup processes.
//   * THE IMAGE ROLLS THROUGH AN LDS ROW RING.  A workgroup (8 waves, one per CU) walks an image top to bottom in steps
//     of STEP output rows; the ring holds the STEP + KH - 1 input rows of the current step and the STEP rows of the
//     next one, so every input row is staged exactly once (no vertical halo re-staging; with the fused 2x bilinear
//     upsample, vae/model.py:163-167, the hi-res rows are blended on the fly from the LOW-RES tensor, bitwise as
//     tile_stage.hip.h does it).  Rows are planar (32-B planes, as tile_stage.hip.h: conflict-free ds_read_b128).
//   * ROW-WINDOW REUSE from registers: for one 16-pixel strip, filter column kx and 32-channel chunk, a wave reads the
//     MF + KH - 1 input-row fragments once and issues MF * KH MFMAs from them (9 LDS reads per 24 MFMAs at KH = 6).
//   * NO K-LOOP BARRIERS, NO WEIGHT TRAFFIC: the only workgroup barrier is one per step (a raw s_barrier behind
//     lgkmcnt(0): __syncthreads() would also drain the step's global stores).  d3 (its weights need 8 waves' registers):
//     one 8-wave workgroup per CU whose two 4-wave halves run STAGGERED (half 0 stages its share of the next step's rows,
//     then computes; half 1 computes, then stages).  d4: 4-wave workgroups, two per CU, independent of each other.
//   * Layers whose K does not fit 144 registers (K = 2304: d4 forward; 2048: d3 forward) split K over wave PAIRS by
//     channel chunk; the odd wave hands its partial sums to the even one through a double-buffered LDS slot guarded by
//     two workgroup-scope counters (no barrier: the pair stays decoupled from the other waves).
//   * STAGING one step ahead.  Upsampled layers: the blend (global loads of the low-res pieces, packed-fp32 lerps, LDS
//     stores).  Plain layers: LDS-DMA (global_load_lds_dwordx4) of whole in-image row planes, no registers, no VALU; the
//     halo columns are zeroed once per launch.
//
// Measured (MI355X, one network pair = 1024 images per launch; profiles/r02_*): d3 forward 85 -> 82 us, d3 input
// gradient 116 -> 70, d4 forward 174 -> 158, d4 input gradient 163 -> 129; at 128 images d3 33 -> 18 / 23 -> 18 us.
// In-kernel stamps (SV_DEBUG_KNOBS builds, SV_RC_STAMP=1): the MFMA loop runs at 18-21 cycles per MFMA (16 = pipe
// rate); what keeps the launch at ~55 % matrix-pipe utilisation is the per-wave serial tail around it -- stores
// (21-26 k cycles per wave and launch), staging (20-28 k; the blend of the upsampled layers 60-120 k: its global-load
// latency and ~300 VALU instructions per 2x2x8-channel item), weight load / unit prologue (25 k) -- which two waves per
// SIMD that synchronise every step overlap only partly.
//
// Planned from the same TapGemmArgs as the other conv kernels (svk_row_conv_try), reading the same prepared weight
// images (conv_api.hip), so it is a drop-in below svk_conv_dispatch_multi.
#include "common.hip.h"
#include "kernels.h"
#include "tile_stage.hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace {

struct RowConvArgs {
  const void* A; const void* Wt; const float* bias; void* out;
  int B, H, W;          // logical (hi-res when `ups`) input extent == output extent (stride 1, SAME)
  int lda, ldo, Ktot, act;
  int y_lo, x_lo;       // tap (ky, kx) reads input pixel (y + ky + y_lo, x + kx + x_lo)
  int bands, band_rows; // an image is cut into `bands` row bands of band_rows rows (a unit of work = one band)
  int lead;             // ADJ configs with bands > 1: every band but the first starts `lead` (= one step) rows early and keeps those rows' low-res
                        // output to itself -- the warm-up step rebuilds what the adjoint carries across the band edge (low-res row i needs the hi-res
                        // rows 2i - 1 .. 2i + 2) -- so a band is independent of its neighbour.  0 otherwise
  int os, ooy, oox;     // output pixel of grid pixel (y, x): (os * y + ooy, os * x + oox) of a [B, os * H, os * W, ldo] tensor (os = 2: one parity
                        // class of a stride-2 layer's input gradient; 1, 0, 0 otherwise)
  const void* mask;     // ADJ configs: the low-res activation whose ReLU mask (> 0) gates the low-res gradient (or null);
                        // `out` is then the LOW-RES gradient [B, H/2, W/2, ldo].
                        // CLS configs: the layer input [B, 2H, 2W, ldo] whose ReLU mask gates the gradient `out` of the same shape (or null)
};
__device__ __forceinline__ int band_row0(const RowConvArgs& g, int band) { return band * g.band_rows - (band ? g.lead : 0); }   // first row a band computes
struct RowConvMulti { RowConvArgs a[8]; int units_per_prob, units, dbg; unsigned long long* stamps; };   // dbg: timing ablations (SV_DEBUG_KNOBS builds only)

template <int KH_, int KW_, int CIN_, int N_, int WIDTH_, int MF_, int NBW_, int KS_, int XG_, int RG_, bool UPS_, int WAVES_ = 8, bool ADJ_ = false, bool CLS_ = false, bool S2D_ = false, bool PAIR_ = false, bool REV_ = false, bool MB_ = false, bool MA_ = false>
struct RowCfg {
  static constexpr int KH = KH_, KW = KW_, CIN = CIN_, N = N_, WIDTH = WIDTH_, MF = MF_, NBW = NBW_, KS = KS_, XG = XG_, RG = RG_;
  static constexpr bool UPS = UPS_;
  static constexpr int WAVES = WAVES_, NT = 64 * WAVES_;      // 8 waves: one workgroup per CU, its two halves staggered;
                                                              // 4 waves: two independent workgroups per CU (<= 80 KB of LDS each)
  // ADJ: the layer's logical input is a 2x bilinear upsample and this is its INPUT GRADIENT: the hi-res gradient rows
  // stay in an LDS ring and leave as the LOW-RES gradient through the adjoint of the resize (+ ReLU mask) -- the hi-res
  // tensor and the stand-alone upsample2x_bwd pass never exist (vae/model.py:163-167 backwards).
  static constexpr bool ADJ = ADJ_;
  // CLS: the MERGED PARITY CLASSES of a stride-2 layer's input gradient (TapGemmArgs::cls_n; conv_api.hip: svg_dgrad_merged_args): a
  // stride-1 KH x KW conv over the dY grid whose N = 4 * (N / 4) columns are (class, channel); class (ph, pw) lands on pixel
  // (2y + ph, 2x + pw) of the [B, 2H, 2W, ldo] gradient, gated by the ReLU mask of the layer input.  Its weight image keeps the
  // class order of the taps (y-major, offsets descending): only the one-time weight load indexes differently.
  static constexpr bool CLS = CLS_;
  // S2D: a STRIDE-2 forward layer (k = 2 KH, pad KH - 1) as the stride-1 KH x KW conv over the space-to-depth view of its input: a
  // "pixel" of the class grid is the 2 x 2 block of input pixels, CIN = 4 x the layer's channels, the 32-channel K chunk cc = (py, px)
  // of s2d tap (ky, kx) is tap (2 ky + py, 2 kx + px) of the layer's own (y-major) weight image -- nothing is re-prepared, only the DMA
  // source address and the one-time weight load index differently.  `A` is the layer's input [B, 2H, 2W, lda], lda = CIN / 4.
  static constexpr bool S2D = S2D_;
  // PAIR: 8-pixel-wide images (d2's 8 x 8 grid): the 16-pixel strip of an MFMA is the same row of TWO images, laid side by side in the ring
  // with a zero gap of KW - 1 pixels (each half sees its own SAME padding); a unit of work is a pair of images.  Only the lane's ring
  // offset, the DMA (one instruction per image) and the store address know.  g.W = 8, g.B = images (the last pair may be half).
  static constexpr bool PAIR = PAIR_;
  // MB (round 4): an upsampled forward layer whose 2x bilinear resize runs ON THE MATRIX PIPE.  The raw LOW-RES rows come in by LDS-DMA
  // (no registers, no VALU) into a small raw ring two steps ahead of their readers; one step ahead, every wave blends its 16-channel
  // fragment of the next step's hi-res rows with MFMAs against per-lane constant weight operands (mb_blend below; the scheme of
  // wgrad_roll.hip's blend stage) and stores them into the planar row ring.  The VALU form (stage_rows: ~350 instructions per thread and
  // step plus the exposed latency of its global loads) was 27 % of the d4 forward launch -- VALU issue is step time one to one beside
  // MFMAs (profiles/r04_valu_mfma_overlap.txt: every v_fma per MFMA costs ~5 % of the MFMA rate).  Needs the low-res width to be
  // the 16 K columns of one MFMA operand row and one 16-channel fragment per wave; bands == 1 (the first window needs <= NRAW raw rows).
  static constexpr bool MB = MB_;
  // MA (round 4): an ADJ layer whose resize ADJOINT runs on the matrix pipe, chained from the accumulators with no LDS round trip.  The conv is
  // computed with SWAPPED operands, X = pixels x W^T: a lane holds 4 consecutive PIXELS of one channel, which is exactly the A-operand layout of
  // a second MFMA that contracts over the pixels (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"):
  //   lo[i][j][c] = sum_Y wy(i, Y) sum_X wx(j, X) hi[Y][X][c]      Z[16 channels x 16 low-res pixels] += A = bf16(X rows Y)[channels x 32 pixels] . B = wy wx
  // with B a per-lane constant (the 32 hi-res pixels of a row are the wave's two 16-pixel strips).  Per step and wave 8 adjoint MFMAs beside 288
  // conv MFMAs; the hi-res out ring in LDS (45 KB), its stores, adjoint_rows' 16 LDS reads and ~190 VALU instructions per item are gone.
  // The hi-res gradient is rounded to bf16 before the adjoint, as it would go to HBM (the same values the two-launch form sees); the fp32 sums run
  // in another order than adj2x_row_bf16's, so the result equals the two-launch form to a bf16 ulp, not bitwise.
  static constexpr bool MA = MA_;
  // (WIDTH 16 -- d3: one strip, the upper half of the A operand is zero and only the 8 low-res pixel columns of Z are stored)
  static_assert(!MA_ || (ADJ_ && (WIDTH_ == 32 || WIDTH_ == 16) && NBW_ == 1 && XG_ == 1 && RG_ == 1 && KS_ == 1 && MF_ == 4 && CIN_ != 8), "matrix-pipe adjoint");
  // (WIDTH 16 -- d3: low-res width 8, the upper half of the blend's K is zero weights over zero-initialised LDS; 8 waves = 8 fragments)
  static_assert(!MB_ || (UPS_ && (WAVES_ == 8 || WAVES_ == 4 || WAVES_ == 2) && (CIN_ / 16) % WAVES_ == 0 && (WIDTH_ == 32 || WIDTH_ == 16) && !ADJ_ && !CLS_ && !S2D_ && !PAIR_ && CIN_ != 8), "matrix-pipe blend");
  static constexpr int FPW = MB_ ? CIN_ / 16 / WAVES_ : 1;     // 16-channel fragments a wave blends
  // WAVES == 2: ONE wave per SIMD with the whole register file (512 registers: the weights of a 16-channel block over the WHOLE K -- 288 for
  // d4 -- stay in one wave, so there is no K-half exchange and no partner to lose the matrix pipe to); two such workgroups per CU
  static constexpr int WPE = WAVES_ == 2 ? 1 : 2;              // waves per SIMD the kernel is compiled for
  static constexpr int NRAW = 6;                               // raw ring depth in low-res rows: blend(s) reads 3, the DMAs of step s write the next 2; a
                                                               // unit's first window + first step need rows 0..5 at once
  static constexpr int RAWB = (WIDTH_ / 2) * CIN_ * 2;         // bytes per raw low-res row, [pixel][channel] as the DMA writes it
  static constexpr int RAWR = MB_ ? (NRAW + 1) * RAWB : 0;     // (+ one zero slot behind the ring: WIDTH 16 reads 16 K columns of 8-pixel rows)
  static constexpr int NSG = WIDTH_ / 16;                      // 16-pixel segments of a hi-res row
  static constexpr int HL = (KW_ - 1) / 2, HR = KW_ - 1 - HL;  // SAME padding: halo pixels left / right (k 6: 2 / 3, k 4: 1 / 2)
  static constexpr bool REV = REV_ || CLS_;                   // the weight image holds the taps y-major with DESCENDING offsets (the parity classes' order)
  static_assert(!PAIR_ || (WIDTH_ == 16 && XG_ == 1 && !UPS_ && !ADJ_ && !CLS_ && CIN_ != 8 && (!S2D_ || CIN_ == 256)), "image pairs");
  // Sub-pixels of 32 channels (e2: a K chunk = one sub-pixel) or of 8 (e1's padded RGB: a K chunk = all four, lane quarter kq = (py, px)).
  // CIN 256 (e3: k 4, pad 1, 64-channel sub-pixels, with PAIR): the 2 x 2 blocks are aligned one pixel EARLIER -- block (Y, X) = input rows
  // 2Y - 1, 2Y x columns 2X - 1, 2X -- so that the four taps of a row are exactly two blocks (KH = KW = 2, y_lo = x_lo = 0): block row / column 8
  // of the 9 x 9 block grid holds real data in its first sub-row / sub-column (the kernel stages it as its halo row / pixel).
  static_assert(!S2D_ || (!UPS_ && !ADJ_ && !CLS_ && (CIN_ == 128 || CIN_ == 32 || (CIN_ == 256 && PAIR_))), "space-to-depth form");
  static_assert(!CLS_ || (!UPS_ && !ADJ_ && KS_ == 1 && (N_ / 4) % 16 == 0), "merged parity classes");
  // TP: 8-channel pixels (the 6-channel head's gradient): one 16-B piece per pixel, so an MFMA K step (32) packs FOUR
  // taps -- the lane quarter kq reads pixel (row + (kq >> 1), column + (kq & 1)): a fragment is a 2 x 2 block of taps, the 6 x 6 filter is
  // exactly 3 x 3 such blocks (K = 288 = 9 steps, none padded).  (Until round 3's end the quarters read four consecutive ROWS of one
  // filter column: 6 x 2 row groups with 2 of 8 rows dummies = 12 steps; 25 % fewer MFMAs and half the fragment reads now.)
  static constexpr bool TP = CIN == 8;
  static_assert(!TP || (KH == 6 && KW == 6 && !UPS_ && KS == 1), "tap-packed form");
  static_assert(!ADJ_ || (!UPS_ && KS_ == 1), "the adjoint epilogue belongs to plain input-gradient layers");
  static constexpr int NCH = TP ? 1 : CIN / 32, CPW = NCH / KS, NBG = N / 16 / NBW;
  static constexpr int KHG = TP ? KH / 2 : KH, KWG = TP ? KW / 2 : KW;   // weight fragments per chunk: KWG x KHG (TP: 2 x 2 tap blocks)
  static_assert(NBG * KS * XG * RG == WAVES, "one role per wave");
  static_assert((TP || CIN % 32 == 0) && NCH % KS == 0 && N % (16 * NBW) == 0 && WIDTH % (16 * XG) == 0, "shape");
  static constexpr int TIW = PAIR ? 2 * (8 + KW - 1) : WIDTH + KW - 1, STEP = RG * MF, WIN = TP ? MF + 4 : MF + KH - 1;   // (TP: fragment w = rows w, w + 1)
  // ring rows.  8 waves: two windows (the current one + the next step's STEP new rows, or the whole first window of the
  // workgroup's NEXT unit, staged during the last step).  4 waves: one window + one step (the next unit's first window
  // is staged between units; the other workgroup of the CU computes meanwhile)
  static constexpr bool PRE = WAVES == 8 && !MB_;             // (MB: a unit's first window is blended between units -- its raw rows arrive during the last step)
  static constexpr int R = PRE ? 2 * (STEP + KH - 1) : 2 * STEP + KH - 1;
  static constexpr int PIXB = TP ? 16 : 32;                   // bytes per pixel in a plane
  // TP: a fragment's lane quarters read two consecutive rows (16 pixels each, the odd quarters one pixel to the right: every 16-lane
  // group of the ds_read_b128 covers 256 contiguous bytes = each bank once); the row pitch is a multiple of 256 B and the ring's first
  // row is kept twice, behind its end, so that rows slot, slot + 1 are always linear
  // MB: the row pitch is WIDTH + 3 pixels, not TIW = WIDTH + 5 -- the last two (zero) halo pixels of a row ARE the first two (zero) halo pixels
  // of the next one (nothing ever writes a halo pixel after the launch's zero fill): 3.3 KB of LDS, which pays for the sixth raw row
  static constexpr int ROWB = TP ? (TIW * 16 + 255) / 256 * 256 : MB_ ? (WIDTH_ + HR) * 32 : TIW * 32;
  static constexpr int RDUP = TP ? 1 : 0;
  static constexpr int NPL = TP ? 1 : CIN / 16, PLB = (R + RDUP) * ROWB;    // planes (two 16-B pieces each; TP: one piece), bytes per plane
  static constexpr int RING = NPL * PLB + (MB_ ? 64 : 0);     // (MB: the aliased halo pixels of the very last row)
  static_assert(!MB_ || (HL <= HR && KW_ == KH_), "MB: the right halo of a row aliases the left halo of the next");
  static constexpr int EXS = (MB_ && WAVES_ <= 4) ? 1 : 2;    // exchange slots per wave pair (4-wave MB: one -- the raw ring takes the second slot's LDS)
  static constexpr int EXF = NBW * MF * 1024;                 // bytes per slot: NBW*MF accumulator fragments of 1 KB
  static constexpr int EXB = KS == 2 ? (WAVES / 2) * EXS * EXF : 0;
  // ADJ: ring of hi-res gradient rows [slot][pixel][N] bf16: STEP rows being written + STEP + 3 being read by the adjoint
  static constexpr int ORR = (ADJ && !MA_) ? 2 * STEP + 3 : 0, OROWB = WIDTH * N * 2, OUTB = ORR * OROWB;
  // MA: per thread 96 B of LDS scratch -- the adjoint's two constant B operands, its two carried accumulators between steps and the X rows of
  // strip 0 while strip 1 computes (24 VGPRs that would otherwise live through the MFMA loops: the kernel sits at the 256-register edge)
  static constexpr int MAWB = MA_ ? NT * 96 : 0;
  static constexpr int LDS = RING + EXB + OUTB + RAWR + MAWB + 64;
  static_assert((WAVES_ != 4 && WAVES_ != 2) || LDS <= 81920, "two workgroups per CU");
  static constexpr int SPW = WIDTH / 16 / XG;                 // strips per wave and row group
  static constexpr int CPP = CIN / 8;                         // 16-B pieces per pixel
};

// rows [Ya, Ya + nrows) x columns [x_lo, x_lo + TIW) of the conv's logical input into ring slots qa.. (mod R); zero
// outside the image (SAME padding lives in hi-res space).  The nt threads t take share `part` of `nparts` of the tasks.
template <typename C>
__device__ __forceinline__ void stage_rows(const RowConvArgs& g, int b, int Ya, int nrows, int qa, char* sRing, int t, int nt,
                                           int part, int nparts) {
  const bf16_t* Ab = (const bf16_t*)g.A;
  if constexpr (C::UPS) {
    const int LH = g.H >> 1, LW = g.W >> 1;
    const int i_lo = (Ya - 1) >> 1, nbi = ((Ya + nrows - 2) >> 1) - i_lo + 1;
    const int j_lo = (g.x_lo - 1) >> 1, nbj = ((g.x_lo + C::TIW - 2) >> 1) - j_lo + 1;
    const int total = nbi * nbj * C::CPP;
    const int t0 = (int)((int64_t)total * part / nparts), t1 = (int)((int64_t)total * (part + 1) / nparts);
    const float inv_nbj = 1.0f / (float)nbj;
    const bf16_t* img = Ab + (int64_t)b * LH * LW * g.lda;
    // two tasks per thread per round: all eight global loads are issued before the first blend, so a thread pays the
    // L2 / HBM latency once per round (a staging share is 1-2 tasks per thread)
    for (int q = t0 + t; q < t1; q += 2 * nt) {
      uint4 ld[2][4];
      int ti[2], tj[2], tc[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int qq = q + k * nt;
        const bool on = qq < t1;
        const int c = qq & (C::CPP - 1), r2 = qq / C::CPP;
        const int bi = (int)(((float)r2 + 0.5f) * inv_nbj), bj = r2 - bi * nbj;
        const int i = i_lo + bi, j = j_lo + bj;
        ti[k] = on ? i : -0x40000000; tj[k] = j; tc[k] = c;
        const int y0 = min(max(i, 0), LH - 1), y1 = min(max(i + 1, 0), LH - 1);
        const int x0 = min(max(j, 0), LW - 1), x1 = min(max(j + 1, 0), LW - 1);
        const bf16_t* p = img + c * 8;
        ld[k][0] = ld[k][1] = ld[k][2] = ld[k][3] = make_uint4(0, 0, 0, 0);
        if (on) {
          ld[k][0] = *(const uint4*)(p + ((int64_t)y0 * LW + x0) * g.lda);
          ld[k][1] = *(const uint4*)(p + ((int64_t)y0 * LW + x1) * g.lda);
          ld[k][2] = *(const uint4*)(p + ((int64_t)y1 * LW + x0) * g.lda);
          ld[k][3] = *(const uint4*)(p + ((int64_t)y1 * LW + x1) * g.lda);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = ti[k], j = tj[k], c = tc[k];
        uint4 blk[2][2];
        blend2x2<bf16_t>(ld[k][0], ld[k][1], ld[k][2], ld[k][3], blk);
#pragma unroll
        for (int dyb = 0; dyb < 2; ++dyb) {
          const int Y = 2 * i + 1 + dyb, d = Y - Ya;
          if ((unsigned)d >= (unsigned)nrows) continue;
          int slot = qa + d;
          if (slot >= C::R) slot -= C::R;
#pragma unroll
          for (int dxb = 0; dxb < 2; ++dxb) {
            const int X = 2 * j + 1 + dxb, xi = X - g.x_lo;
            if ((unsigned)xi >= (unsigned)C::TIW) continue;
            const bool in = (unsigned)Y < (unsigned)g.H && (unsigned)X < (unsigned)g.W;
            *(uint4*)(sRing + (c >> 1) * C::PLB + slot * C::ROWB + xi * 32 + (c & 1) * 16) = in ? blk[dyb][dxb] : make_uint4(0, 0, 0, 0);
          }
        }
      }
    }
  } else {
    const int total = nrows * C::TIW * C::CPP;
    const int t0 = (int)((int64_t)total * part / nparts), t1 = (int)((int64_t)total * (part + 1) / nparts);
    const bf16_t* img = Ab + (int64_t)b * g.H * g.W * g.lda;
    constexpr float inv_row = 1.0f / (float)(C::TIW * C::CPP);
    for (int q = t0 + t; q < t1; q += 4 * nt) {             // four 16-B loads in flight per thread
      uint4 v[4];
      int off[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int qq = q + k * nt;
        const int d = (int)(((float)qq + 0.5f) * inv_row), r2 = qq - d * (C::TIW * C::CPP);
        const int c = r2 & (C::CPP - 1), xi = r2 / C::CPP;
        const int Y = Ya + d, X = g.x_lo + xi;
        int slot = qa + d;
        if (slot >= C::R) slot -= C::R;
        off[k] = qq < t1 ? (c >> 1) * C::PLB + slot * C::ROWB + xi * 32 + (c & 1) * 16 : -1;
        v[k] = make_uint4(0, 0, 0, 0);
        if (qq < t1 && (unsigned)Y < (unsigned)g.H && (unsigned)X < (unsigned)g.W) v[k] = *(const uint4*)(img + ((int64_t)Y * g.W + X) * g.lda + c * 8);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (off[k] >= 0) *(uint4*)(sRing + off[k]) = v[k];
    }
  }
}

// Non-upsampled layers: the rows go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU, the wave does
// not wait), issued a whole step ahead.  One wave-instruction moves one (row, plane): WIDTH pixels x 32 B = the in-image
// part of the row in that plane, lane-linear (lane l = pixel l>>1, piece l&1) exactly as the plane stores it.  The halo
// columns are zeroed once per launch and never written again; rows outside the image are zero-filled by plain stores.
template <typename C>
__device__ __forceinline__ void stage_rows_dma(const RowConvArgs& g, int b, int Ya, int nrows, int qa, char* sRing, int w, int nw, int lane) {
  constexpr int LPR = C::TP ? C::WIDTH : C::WIDTH * 2;      // lanes per (row, plane): one 16-B piece each
  static_assert(LPR <= 64, "one DMA instruction per (row, plane)");
  const bf16_t* img = (const bf16_t*)g.A + (int64_t)b * g.H * g.W * g.lda * (C::S2D ? 4 : 1);
  const int PL = -g.x_lo;
  const bool on = lane < LPR;
  if constexpr (C::PAIR) {                                   // b = pair index: images 2b, 2b + 1, each row half its own 16-lane DMA
    for (int idx = w; idx < nrows * C::NPL; idx += nw) {
      const int d = idx / C::NPL, p = idx - d * C::NPL;
      const int Y = Ya + d;
      int slot = qa + d;
      if (slot >= C::R) slot -= C::R;
      const bool inside = (unsigned)Y < (unsigned)g.H;
#pragma unroll
      for (int im = 0; im < 2; ++im) {
        const int bi = 2 * b + im;
        char* dst = sRing + p * C::PLB + slot * C::ROWB + (PL + im * (8 + C::KW - 1)) * 32;
        if constexpr (C::S2D) {
          // plane p = sub-pixel (py, px) = (p >> 3, (p >> 2) & 1), 16-channel quarter p & 3 of input pixel (2Y + py - 1, 2X + px - 1), X = 0..8;
          // a sub-row / sub-column outside the 16 x 16 input is never written by a DMA: zero-filled row, or the ring's initial zeros
          const int py = p >> 3, px = (p >> 2) & 1, iy = 2 * Y + py - 1, ix = 2 * (lane >> 1) + px - 1;
          const bf16_t* src = (const bf16_t*)g.A + (((int64_t)bi * 16 + iy) * 16 + ix) * g.lda + (p & 3) * 16 + (lane & 1) * 8;
          if (lane < 18) {
            if ((unsigned)iy < 16u && bi < g.B) {
              if ((unsigned)ix < 16u) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
          }
          continue;
        }
        const bf16_t* src = (const bf16_t*)g.A + (((int64_t)bi * g.H + Y) * 8 + (lane >> 1)) * g.lda + (2 * p + (lane & 1)) * 8;
        if (lane < 16) {
          if (inside && bi < g.B) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
          else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
        }
      }
    }
    return;
  }
  for (int idx = w; idx < nrows * C::NPL; idx += nw) {
    const int d = idx / C::NPL, p = idx - d * C::NPL;
    const int Y = Ya + d;
    int slot = qa + d;
    if (slot >= C::R) slot -= C::R;
    // S2D, 32-channel sub-pixels: plane p = sub-pixel (py, px) = (p >> 2, (p >> 1) & 1), channel half p & 1 of the input pixel (2Y + py, 2X + px)
    const bf16_t* src = C::TP ? img + ((int64_t)Y * g.W + lane) * g.lda
                      : (C::S2D && C::CIN == 128) ? img + ((int64_t)(2 * Y + (p >> 2)) * (2 * g.W) + 2 * (lane >> 1) + ((p >> 1) & 1)) * g.lda + (2 * (p & 1) + (lane & 1)) * 8
                      : C::S2D ? img + ((int64_t)(2 * Y + p) * (2 * g.W) + lane) * g.lda          // 8-channel sub-pixels: plane p = input row parity, the row is linear
                              : img + ((int64_t)Y * g.W + (lane >> 1)) * g.lda + (2 * p + (lane & 1)) * 8;
    const bool inside = (unsigned)Y < (unsigned)g.H;
#pragma unroll
    for (int dup = 0; dup < (C::RDUP ? 2 : 1); ++dup) {
      if (dup && slot >= C::RDUP) break;                     // TP: the ring's first rows live a second time behind its end
      char* dst = sRing + p * C::PLB + (slot + dup * C::R) * C::ROWB + PL * C::PIXB;
      if (inside) {
        if (on) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else if (on) {
        *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- MB (RowCfg::MB)
// 64 lanes x 16 B, global -> LDS (lane l lands at lds + 16 l), as inline assembly: behind the DMA builtin the compiler orders every later
// LDS read of an object the DMA may alias behind s_waitcnt vmcnt(0) -- the whole ring here -- which would park the wave on its newest
// transfer at the first fragment read (wgrad_roll.hip has the same helper).  The only waits are the counted ones at the end of a step.
__device__ __forceinline__ void rc_dma16(const void* base, uint32_t off, const char* lds) {    // base: wave-uniform; off: this lane's byte offset
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ short4_t rc_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

// The problem's geometry BY VALUE, read once per unit: a RowConvArgs reference is a pointer into the kernel-argument segment, and every
// use inside the step loop became a scalar load + wait (with one wave per SIMD nothing hides them: ~2 000 cycles per step)
struct MbGeom { const bf16_t* A; int H, LH, LW, lda, x_lo; };
__device__ __forceinline__ MbGeom mb_geom(const RowConvArgs& g) { return MbGeom{(const bf16_t*)g.A, g.H, g.H >> 1, g.W >> 1, g.lda, g.x_lo}; }
// raw low-res row r (in the image) of image b -> raw ring slot r % NRAW, half hf (pixels 8 hf .. 8 hf + 7): ONE wave-instruction
template <typename C>
__device__ __forceinline__ void mb_dma_half(const MbGeom& g, int b, int r, int hf, char* sRaw, int lane) {
  constexpr int PPI = 1024 / (C::CIN * 2);                   // pixels per wave-instruction (64 channels: 8)
  static_assert(PPI * 2 == C::WIDTH / 2, "two instructions per raw row");
  const bf16_t* rowp = g.A + (((int64_t)b * g.LH + r) * g.LW + hf * PPI) * g.lda;    // wave-uniform
  const uint32_t off = (uint32_t)(((lane / (64 / PPI)) * g.lda + (lane % (64 / PPI)) * 8) * 2);
  rc_dma16(rowp, off, sRaw + (r % C::NRAW) * C::RAWB + hf * 1024);
}
// rows [r0, r1] (clamped to the image by the caller), two instructions each, dealt round-robin to the waves
template <typename C>
__device__ __forceinline__ void mb_dma_rows(const MbGeom& g, int b, int r0, int r1, char* sRaw, int wave, int lane) {
  for (int idx = wave; idx < 2 * (r1 - r0 + 1); idx += C::WAVES) mb_dma_half<C>(g, b, r0 + (idx >> 1), idx & 1, sRaw, lane);
}

// The per-lane constant B operands of the blend: for output row parity dyb and 16-pixel segment sg,
//   hi[Y = 2 bb + 1 + dyb][X = 16 sg + (lane & 15)] = sum over h in {0, 1}, jj in 0..15 of  wy(h, dyb) wx(jj, X) raw[clamp(bb + h)][jj]
// with tf.image.resize's half-pixel weights (vae/model.py:163-167; blend2x2 in tile_stage.hip.h): odd Y = .75 lo + .25 hi, even Y = .25 lo + .75 hi,
// odd X = 2i + 1: .75 raw[i] + .25 raw[min(i + 1, LW - 1)], even X = 2i: .25 raw[max(i - 1, 0)] + .75 raw[i].  K index 8 (lane >> 4) + j of the
// MFMA <-> (h = j >> 2, jj = 4 (lane >> 4) + (j & 3)) (the transposed read's order).  All products are exact in bf16 (1/16, 3/16, 9/16, 1/4, 3/4).
template <typename C>
__device__ __forceinline__ void mb_weights(int lane, short8_t (&bw)[2][C::NSG]) {
  constexpr int LW = C::WIDTH / 2;
  const int gq = (lane >> 4) * 4;
#pragma unroll
  for (int dyb = 0; dyb < 2; ++dyb)
#pragma unroll
    for (int sg = 0; sg < C::NSG; ++sg) {
      const int X = 16 * sg + (lane & 15), i = X >> 1;
      const int ja = (X & 1) ? i : max(i - 1, 0), jb = (X & 1) ? min(i + 1, LW - 1) : i;      // columns carrying .25 / .75 for even X, .75 / .25 for odd X
      const float wa = (X & 1) ? 0.75f : 0.25f, wb = 1.f - wa;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int h = j >> 2, jj = gq + (j & 3);
        const float wx = (jj == ja ? wa : 0.f) + (jj == jb ? wb : 0.f);
        const float wy = (h == 0) == (dyb == 0) ? 0.75f : 0.25f;
        bw[dyb][sg][j] = (short)(__float_as_uint(wx * wy) >> 16);
      }
    }
}

// hi-res rows [Ya, Ya + n) of the conv's logical input -> ring slots qa .. (mod R), this wave's 16-channel fragment (plane `wave`), from the
// raw ring (its rows must have landed and be visible: counted vmcnt + barrier).  Rows outside the image are zeros (SAME padding lives in
// hi-res space); the halo COLUMNS were zeroed once per launch and are never written.  Per block row: two transposed reads (the A operand,
// K = 2 raw rows x 16 columns), and per hi-res row and segment one MFMA + one 8-B store (4 consecutive channels of one pixel per lane).
// NB block rows per call, all reads first, then all MFMAs, then the stores (a block row alone is a chain of LDS latency -> MFMA latency ->
// convert -> store: ~400 cycles of a wave that has nothing else to issue).
struct MbLane { int rd; char* wr; };                         // of the wave's FIRST fragment; fragment f adds 32 B / one plane
template <typename C>
__device__ __forceinline__ MbLane mb_lane(const MbGeom& g, char* sRing, int wave, int lane) {
  const int g4 = lane >> 4, cf = wave * C::FPW;
  return MbLane{(4 * g4 + ((lane & 15) >> 2)) * (C::CIN * 2) + cf * 32 + (lane & 3) * 8, sRing + cf * C::PLB + ((lane & 15) - g.x_lo) * 32 + g4 * 8};
}
template <typename C, int NB>
__device__ __forceinline__ void mb_blend_rows(const MbGeom& g, int bb0, int Ya, int n, int qa, const char* sRaw, const MbLane& ml,
                                              const short8_t (&bw)[2][C::NSG]) {
  const int LH = g.LH;
#pragma unroll
  for (int f = 0; f < C::FPW; ++f) {
    short4_t a_lo[NB], a_hi[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int rlo = min(max(bb0 + k, 0), LH - 1), rhi = min(max(bb0 + k + 1, 0), LH - 1);
      a_lo[k] = rc_tr16(sRaw + (rlo % C::NRAW) * C::RAWB + ml.rd + f * 32);
      a_hi[k] = rc_tr16(sRaw + (rhi % C::NRAW) * C::RAWB + ml.rd + f * 32);
    }
    f32x4 dd[NB][2][C::NSG];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const short8_t af = (short8_t){a_lo[k][0], a_lo[k][1], a_lo[k][2], a_lo[k][3], a_hi[k][0], a_hi[k][1], a_hi[k][2], a_hi[k][3]};
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dyb = 0; dyb < 2; ++dyb)
#pragma unroll
        for (int sg = 0; sg < C::NSG; ++sg)
          dd[k][dyb][sg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bw[dyb][sg]), z, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int dyb = 0; dyb < 2; ++dyb) {
        const int Y = 2 * (bb0 + k) + 1 + dyb, d = Y - Ya;
        if ((unsigned)d >= (unsigned)n) continue;            // wave-uniform
        int slot = qa + d;
        if (slot >= C::R) slot -= C::R;
        const bool rowin = (unsigned)Y < (unsigned)g.H;      // wave-uniform
        if (rowin) {
#pragma unroll
          for (int sg = 0; sg < C::NSG; ++sg) {
            const f32x4 v4 = dd[k][dyb][sg];
            const bf16x2 lo2 = __builtin_convertvector((f32x2){v4[0], v4[1]}, bf16x2), hi2 = __builtin_convertvector((f32x2){v4[2], v4[3]}, bf16x2);
            *(uint2*)(ml.wr + f * C::PLB + slot * C::ROWB + sg * 512) = make_uint2(__builtin_bit_cast(uint32_t, lo2), __builtin_bit_cast(uint32_t, hi2));
          }
        } else {
#pragma unroll
          for (int sg = 0; sg < C::NSG; ++sg) *(uint2*)(ml.wr + f * C::PLB + slot * C::ROWB + sg * 512) = make_uint2(0u, 0u);
        }
      }
  }
}
// any row range: block rows (Ya - 1) >> 1 .. (Ya + n - 2) >> 1, two per call
template <typename C>
__device__ __forceinline__ void mb_blend(const MbGeom& g, int Ya, int n, int qa, const char* sRaw, const MbLane& ml, const short8_t (&bw)[2][C::NSG]) {
  const int b_lo = (Ya - 1) >> 1, b_hi = (Ya + n - 2) >> 1;
  int bb = b_lo;
  for (; bb + 1 <= b_hi; bb += 2) mb_blend_rows<C, 2>(g, bb, Ya, n, qa, sRaw, ml, bw);
  if (bb <= b_hi) mb_blend_rows<C, 1>(g, bb, Ya, n, qa, sRaw, ml, bw);
}

// The raw rows of a unit's first window (rows [Ya, Ya + STEP + KH - 1)) AND of its first step's blend: low-res rows 0 .. 5 of a whole image
// (bands == 1), all at once -- issued during the previous unit's last step (the raw ring is idle then) or, for a workgroup's first unit,
// in front of the launch's first wait.
template <typename C>
__device__ __forceinline__ void mb_dma_first(const MbGeom& g, int b, int Ya, char* sRaw, int wave, int lane) {
  constexpr int NW = C::STEP + C::KH - 1;
  const int LH = g.LH;
  // the window's block rows (Ya - 1) >> 1 .. and the first step's (rows up to Ya + NW + STEP - 1): raw rows up to ((Ya + NW + STEP - 2) >> 1) + 1
  const int r0 = min(max((Ya - 1) >> 1, 0), LH - 1), r1 = min(((Ya + NW + C::STEP - 2) >> 1) + 1, LH - 1);
  mb_dma_rows<C>(g, b, r0, r1, sRaw, wave, lane);
}

// ADJ: low-res gradient rows [e_lo, e_hi] of image b from the hi-res gradient rows in the out ring:
//   g_lo[i, j] = sum_{a, d in -1..2} wy[a] wx[d] g_hi[clamp(2i + a), clamp(2j + d)],  w = (.25, .75, .75, .25)
// then the ReLU mask of the low-res activation -- the arithmetic (and summation order) of upsample2x_bwd_kernel
// (pointwise.hip: both call adj2x_row_bf16, common.hip.h), on the same bf16-rounded hi-res values: bitwise the unfused result.
template <typename C>
__device__ __forceinline__ void adjoint_rows(const RowConvArgs& g, int b, int e_lo, int e_hi, int obase, const char* sOut, int t, int nt) {
  constexpr int NP8 = C::N / 8, LW = C::WIDTH / 2, IPR = LW * NP8;
  // A low-res row is IPR = 128 items (pixel j, 8-channel piece c) in every ADJ configuration: a thread keeps ITS (j, c) for the whole
  // launch (column offsets, output offset: computed once, hoisted out of the step loop) and a wave's 64 items share the row i, so
  // the row clamps, the out-ring slots (a modulo by ORR each) and the row bases are scalar work.  VALU instructions are step time
  // one to one in this kernel: the index arithmetic was ~45 of ~190 per item.
  static_assert(IPR % 64 == 0 && C::NT % IPR == 0, "a wave's items share their low-res row");
  const int LH = g.H >> 1;
  const int r = t % IPR, j = r / NP8, c = r - j * NP8;
  const int xo[4] = {max(2 * j - 1, 0) * (C::N * 2) + c * 16, 2 * j * (C::N * 2) + c * 16, (2 * j + 1) * (C::N * 2) + c * 16,
                     min(2 * j + 2, C::WIDTH - 1) * (C::N * 2) + c * 16};
  const int64_t o_jc = (int64_t)j * g.ldo + c * 8;
  for (int i = e_lo + __builtin_amdgcn_readfirstlane(t / IPR); i <= e_hi; i += C::NT / IPR) {
    const int64_t o = ((int64_t)b * LH + i) * LW * g.ldo + o_jc;
    uint4 mv = make_uint4(0, 0, 0, 0);
    if (g.mask) mv = *(const uint4*)((const bf16_t*)g.mask + o);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int a = -1; a <= 2; ++a) {
      const int oy = min(max(2 * i + a, 0), g.H - 1);
      const char* row = sOut + ((obase + oy) % C::ORR) * C::OROWB;
      adj2x_row_bf16(acc, *(const uint4*)(row + xo[0]), *(const uint4*)(row + xo[1]), *(const uint4*)(row + xo[2]), *(const uint4*)(row + xo[3]), a == -1 || a == 2);
    }
    bf16_t res[8], m8[8];
    *(uint4*)m8 = mv;
#pragma unroll
    for (int e = 0; e < 8; ++e) res[e] = (bf16_t)((!g.mask || (float)m8[e] > 0.f) ? acc[e] : 0.f);
    *(uint4*)((bf16_t*)g.out + o) = *(const uint4*)res;
  }
}

#ifdef SV_DEBUG_KNOBS
#define SV_STAMP(acc_) do { if (mg.stamps) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); acc_ += t__ - tlast; tlast = t__; } } while (0)
#else
#define SV_STAMP(acc_) do {} while (0)
#endif

template <typename C>
__global__ __launch_bounds__(C::NT, C::WPE) void row_conv_kernel(const RowConvMulti mg) {
  constexpr int KH = C::KH, KW = C::KW, MF = C::MF, NBW = C::NBW, CPW = C::CPW, WIN = C::WIN, R = C::R, TIW = C::TIW, STEP = C::STEP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sRing = smem;
  char* sEx = smem + C::RING;
  char* sOut = smem + C::RING + C::EXB;                      // ADJ: hi-res gradient rows
  char* sRaw = smem + C::RING + C::EXB + C::OUTB;            // MB: raw low-res rows (DMA)
  char* sMaw = smem + C::RING + C::EXB + C::OUTB + C::RAWR; // MA: [thread][2][16 B]
  int* sFlag = (int*)(smem + C::RING + C::EXB + C::OUTB + C::RAWR + C::MAWB);    // [pairs][2]: strips produced (odd wave), consumed (even wave)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int idx = wave;
  const int ks = idx % C::KS; idx /= C::KS;                  // K half (fastest: partners are waves 2p, 2p+1 -- different SIMDs)
  const int nbg = idx % C::NBG; idx /= C::NBG;
  const int xg = idx % C::XG; idx /= C::XG;
  const int rg = idx;                                        // row group
  const int pair = wave >> 1, half = wave >> 2;
  constexpr bool STAG = C::WAVES == 8 && C::UPS && !C::MB;  // staggered halves (only the VALU blend staging needs them)
  constexpr int NT = C::NT;
  const int dbg0 = SV_DBG(mg.dbg);
  const int m = lane & 15, kq = lane >> 4;
  const int lane_off = C::TP ? (m + (kq & 1)) * 16 + (kq >> 1) * C::ROWB : (m + (C::PAIR ? (m >> 3) * (KW - 1) : 0)) * 32 + (kq & 1) * 16 + ((kq >> 1) + ks * CPW * 2) * C::PLB;
  if (tid < 16) sFlag[tid] = 0;
  const int dbg = SV_DBG(mg.dbg);                           // 1 skip staging, 2 skip the MFMA loop, 4 skip the stores, 8 skip the K-half exchange

  unsigned long long tlast = 0, t_stage = 0, t_mfma = 0, t_exch = 0, t_epi = 0, t_bar = 0, t_other = 0;
#ifdef SV_DEBUG_KNOBS
  if (mg.stamps) tlast = __builtin_amdgcn_s_memtime();
#endif
  bf16x8 Wr[NBW][CPW][C::KWG][C::KHG];
  float bv[NBW][4];
  // MA: the adjoint's constant B operands (.25 wx and .75 wx), its two carried accumulators (low-res rows i0 - 1 and i0 at the top of a step),
  // the bf16 X rows of the step's strips and the prefetched ReLU-mask words
  uint2 ma_m[3];
  if constexpr (C::MA) {
    // B[k = 8 (lane >> 4) + j][col = lane & 15 = low-res pixel jl]: hi-res pixel X = (j < 4 ? 4 (lane >> 4) + j : 16 + 4 (lane >> 4) + j - 4);
    // wx(jl, X): even X = 2i: .25 [jl == max(i - 1, 0)] + .75 [jl == i]; odd X = 2i + 1: .75 [jl == i] + .25 [jl == min(i + 1, LW - 1)]
    short8_t q8, t8;
    const int jl = lane & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int X = (j < 4 ? 0 : 16) + 4 * (lane >> 4) + (j & 3), i = X >> 1;
      const int ja = (X & 1) ? i : max(i - 1, 0), jb = (X & 1) ? min(i + 1, C::WIDTH / 2 - 1) : i;
      const float wa = (X & 1) ? 0.75f : 0.25f;
      const float wx = X >= C::WIDTH ? 0.f : (jl == ja ? wa : 0.f) + (jl == jb ? 1.f - wa : 0.f);
      q8[j] = (short)(__float_as_uint(0.25f * wx) >> 16);
      t8[j] = (short)(__float_as_uint(0.75f * wx) >> 16);
    }
    *(short8_t*)(sMaw + tid * 96) = q8;                      // (every thread reads back only what it wrote itself: no barrier needed)
    *(short8_t*)(sMaw + tid * 96 + 16) = t8;
  }
  short8_t mbw[2][C::MB ? C::NSG : 1];                                        // MB: the blend's constant weight operands
  MbLane mbl = MbLane{0, nullptr};
  if constexpr (C::MB) mb_weights<C>(lane, mbw);
  int obase = 0;                                             // ADJ: out-ring slot of hi-res row 0 of the current unit
  int cur_prob = -1;
  int strips = 0;                                            // strips this wave has finished (the pair counts in lockstep)

  // Every staging job is issued one step ahead: the next step's STEP new rows or -- in a unit's last step -- the whole
  // first window of the workgroup's NEXT unit, which lands right behind the current window (R = two windows).
  constexpr int NST = C::SPW * NBW * MF;                      // global stores a storing wave issues per step (behind its DMAs)
  if (!(dbg & 1)) {
    // PAIR (four problems: twins x output-channel halves): unit u = (r0, prob) with prob fastest, so a workgroup (grid % problems == 0)
    // stays on one problem and loads its weights once
    const int np = mg.units / mg.units_per_prob;
    const int prob = C::PAIR ? (int)blockIdx.x % np : (int)blockIdx.x / mg.units_per_prob, r0 = C::PAIR ? (int)blockIdx.x / np : (int)blockIdx.x - prob * mg.units_per_prob;
    const RowConvArgs& g = mg.a[prob];
    if constexpr (C::MB) {
      // the halo columns are zeros for the whole launch (the blend only ever writes in-image columns); the raw ring too: at WIDTH 16 the
      // blend's K columns 8..15 read the NEXT raw slot (zero weights, but 0 x non-finite garbage would be NaN)
      for (int q = tid; q < C::RING / 16; q += NT) *(uint4*)(sRing + q * 16) = make_uint4(0, 0, 0, 0);
      for (int q = tid; q < C::RAWR / 16; q += NT) *(uint4*)(sRaw + q * 16) = make_uint4(0, 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const MbGeom mg0 = mb_geom(g);
      mbl = mb_lane<C>(mg0, sRing, wave, lane);
      mb_dma_first<C>(mg0, r0 / g.bands, band_row0(g, r0 % g.bands) + g.y_lo, sRaw, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      mb_blend<C>(mg0, band_row0(g, r0 % g.bands) + g.y_lo, STEP + KH - 1, 0, sRaw, mbl, mbw);
      // (the kernel's first __syncthreads() below publishes the window; the first step's raw rows are in the ring already)
    } else
    if constexpr (C::UPS) stage_rows<C>(g, r0 / g.bands, band_row0(g, r0 % g.bands) + g.y_lo, STEP + KH - 1, 0, sRing, tid, NT, 0, 1);
    else {
      // The whole ring starts as zeros, once per launch: the halo columns (the DMAs only ever write in-image pixels).  (The tap-packed
      // form used to read two DUMMY rows behind the window with zero weights -- 0 x stale-LDS-garbage is NaN when the garbage is not
      // finite, hence zeros everywhere; its 2 x 2 tap blocks have no dummies any more, the halo columns still need them.)
      for (int q = tid; q < C::RING / 16; q += NT) *(uint4*)(sRing + q * 16) = make_uint4(0, 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage_rows_dma<C>(g, r0 / g.bands, band_row0(g, r0 % g.bands) + g.y_lo, STEP + KH - 1, 0, sRing, wave, C::WAVES, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  int q0 = 0;                                                // ring slot of input row y0 + y_lo of the current step
  for (int u = blockIdx.x; u < mg.units; u += gridDim.x) {
    const int np = mg.units / mg.units_per_prob;
    const int prob = C::PAIR ? u % np : u / mg.units_per_prob;
    // MB (one wave per SIMD has nobody to hide a scalar load behind): the problem BY VALUE -- as a reference every g.field in the step loop and
    // the epilogue is a load from the kernel-argument segment + a wait
#ifdef SV_RC_ARGS_BY_VALUE_ALL
    using GRef = const RowConvArgs;
#else
    using GRef = typename std::conditional<C::MB, const RowConvArgs, const RowConvArgs&>::type;
#endif
    GRef g = mg.a[prob];
    const int r0 = C::PAIR ? u / np : u - prob * mg.units_per_prob;
    const int band = r0 % g.bands, b = r0 / g.bands;
    const int yb = band_row0(g, band);
    const int un = u + gridDim.x;                            // the next unit of this workgroup
    const bool has_next = un < mg.units;
    const int nprob = has_next ? (C::PAIR ? un % np : un / mg.units_per_prob) : prob;
    const RowConvArgs& gn = mg.a[nprob];
    const int rn = C::PAIR ? un / np : un - nprob * mg.units_per_prob;
    if (prob != cur_prob) {
      cur_prob = prob;
      const bf16_t* Wt = (const bf16_t*)g.Wt;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const int n = (nbg * NBW + nb) * 16 + m;
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
          for (int kx = 0; kx < C::KWG; ++kx)
#pragma unroll
            for (int ky = 0; ky < C::KHG; ++ky) {
              if constexpr (C::TP) {                         // fragment (kx, ky) = tap block: lane quarter kq carries tap (2 ky + (kq >> 1), 2 kx + (kq & 1))
                Wr[nb][cc][kx][ky] = *(const bf16x8*)(Wt + (int64_t)n * g.Ktot + ((2 * kx + (kq & 1)) * KH + 2 * ky + (kq >> 1)) * 8);
              } else
                if constexpr (C::S2D && C::CIN == 128) {
                  const int c4 = ks * CPW + cc, py = c4 >> 1, px = c4 & 1;
                  Wr[nb][cc][kx][ky] = *(const bf16x8*)(Wt + (int64_t)n * g.Ktot + ((2 * ky + py) * (2 * KW) + 2 * kx + px) * 32 + kq * 8);
                } else if constexpr (C::S2D && C::CIN == 256) {
                  const int c8 = ks * CPW + cc, py = c8 >> 2, px = (c8 >> 1) & 1;
                  Wr[nb][cc][kx][ky] = *(const bf16x8*)(Wt + (int64_t)n * g.Ktot + ((2 * ky + py) * (2 * KW) + 2 * kx + px) * 64 + (c8 & 1) * 32 + kq * 8);
                } else if constexpr (C::S2D) {
                  Wr[nb][cc][kx][ky] = *(const bf16x8*)(Wt + (int64_t)n * g.Ktot + ((2 * ky + (kq >> 1)) * (2 * KW) + 2 * kx + (kq & 1)) * 8);
                } else
                Wr[nb][cc][kx][ky] = *(const bf16x8*)(Wt + (int64_t)n * g.Ktot + (C::REV ? (KH - 1 - ky) * KW + (KW - 1 - kx) : kx * KH + ky) * C::CIN + (ks * CPW + cc) * 32 + kq * 8);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[nb][e] = g.bias ? g.bias[(nbg * NBW + nb) * 16 + kq * 4 + e] : 0.f;
      }
    }
    if (u == (int)blockIdx.x) __syncthreads();               // the first window (and the flag words) are in place
    SV_STAMP(t_other);
    const int nsteps = (g.band_rows + (band ? g.lead : 0)) / STEP;
    MbGeom mgu = MbGeom{nullptr, 0, 0, 0, 0, 0}, mgn = mgu;  // MB: this unit's and the next unit's geometry, in registers
    int mb_nb = 0, mb_nY = 0;
    if constexpr (C::MB) { mgu = mb_geom(g); mgn = mb_geom(gn); mb_nb = rn / gn.bands; mb_nY = band_row0(gn, rn % gn.bands) + gn.y_lo; }
    // ADJ: next low-res row to emit.  A later band starts at the row its predecessor cannot finish (low-res row i needs hi-res rows up to 2i + 2)
    int emitted = band ? ((band * g.band_rows) >> 1) - 1 : 0;
    const bool mute0 = C::ADJ && band > 0 && g.lead > 0;        // the band's first step is the warm-up: nothing of it is emitted
    for (int s = 0; s < nsteps; ++s) {
      const int y0 = yb + s * STEP;
      const bool more = s + 1 < nsteps;
      int qn = q0 + STEP + KH - 1;                           // ring slot right behind the current window
      if (qn >= R) qn -= R;
      // the staging job of this step
      const bool job = (more || (has_next && C::PRE)) && !(dbg & 1);
      const RowConvArgs& gj = more ? g : gn;
      const int jb = more ? b : rn / gn.bands;
      const int jY = more ? y0 + g.y_lo + STEP + KH - 1 : band_row0(gn, rn % gn.bands) + gn.y_lo;
      const int jrows = more ? STEP : STEP + KH - 1;
      if constexpr (C::ADJ && !C::MA) {
        // hi-res rows < y0 are complete: low-res row i needs hi-res rows 2i-1 .. 2i+2
        const int e_hi = (y0 - 3) >> 1;
        if (e_hi >= emitted && !(dbg & 4)) adjoint_rows<C>(g, b, emitted, e_hi, obase, sOut, tid, NT);
        if (e_hi >= emitted) emitted = e_hi + 1;
      }
      if constexpr (C::MB) {
        // the next step's STEP rows from the raw ring (landed before the barrier that ended the previous step), then the two raw rows the
        // blend of the step after will add to them: they land while this step computes
        // (priority: the SIMD's other wave -- the CU's second workgroup -- is usually inside its 144-MFMA strip loop and, being older half of
        //  the time, wins every arbitration of the matrix pipe: without it the blend's eight MFMAs waited ~2 000 cycles per step)
        __builtin_amdgcn_s_setprio(3);
        if (job) {
          if (jY & 1) mb_blend_rows<C, 2>(mgu, (jY - 1) >> 1, jY, STEP, qn, sRaw, mbl, mbw);   // k 6: the step's rows are exactly two block rows
          else mb_blend<C>(mgu, jY, STEP, qn, sRaw, mbl, mbw);                                   // k 4: they straddle three
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef SV_MB_STAMP_SPLIT
        SV_STAMP(t_exch);                                    // (diagnostic: the blend alone; the DMA issue stays in t_stage)
#endif
        if (s + 2 < nsteps && !(dbg & 1)) {
          // the blend of the NEXT step reads raw rows up to ((jY + 2 STEP - 2) >> 1) + 1; this step's reached ((jY + STEP - 2) >> 1) + 1
          const int LHm = mgu.LH, have = min(((jY + STEP - 2) >> 1) + 1, LHm - 1), want = min(((jY + 2 * STEP - 2) >> 1) + 1, LHm - 1);
          if (want > have) mb_dma_rows<C>(mgu, b, have + 1, want, sRaw, wave, lane);
        } else if (!more && has_next && !(dbg & 1)) {
          // the unit's last step: nothing reads the raw ring any more -- the next unit's first rows land while this step computes
          mb_dma_first<C>(mgn, mb_nb, mb_nY, sRaw, wave, lane);
        }
      } else
      if constexpr (STAG) {
        if (job && half == 0) stage_rows<C>(gj, jb, jY, jrows, qn, sRing, tid, 256, 0, 2);
      } else if constexpr (C::UPS) {
        if (job) stage_rows<C>(gj, jb, jY, jrows, qn, sRing, tid, NT, 0, 1);
      } else {
        if (job) stage_rows_dma<C>(gj, jb, jY, jrows, qn, sRing, wave, C::WAVES, lane);      // lands while this step computes
      }
      SV_STAMP(t_stage);
      // ---------------- compute: this wave's MF rows x SPW strips x NBW channel blocks
      const int yw = y0 + rg * MF;
      int qw = q0 + rg * MF;
      if (qw >= R) qw -= R;
      for (int si = 0; si < C::SPW; ++si) {
        const int x0 = (xg + si * C::XG) * 16;
        int abase[WIN];
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
          int slot = qw + j;
          if (slot >= R) slot -= R;
          abase[j] = lane_off + slot * C::ROWB + x0 * C::PIXB;
        }
        if constexpr (C::MA) {
          if (si == C::SPW - 1) {
            // the ReLU-mask words of the low-res rows this step completes (i0 - 1, i0, and i0 + 1 at the bottom): in flight under the last strip's MFMAs
            const int LHa = g.H >> 1, LWa = g.W >> 1, i0 = yw >> 1;
            const int64_t orow = ((int64_t)b * LHa + i0) * LWa * g.ldo + (int64_t)m * g.ldo + nbg * 16 + kq * 4;
            ma_m[0] = ma_m[1] = ma_m[2] = make_uint2(0x3f803f80u, 0x3f803f80u);
            if (g.mask && (C::WIDTH == 32 || m < C::WIDTH / 2)) {
              if (yw > 0) ma_m[0] = *(const uint2*)((const bf16_t*)g.mask + orow - (int64_t)LWa * g.ldo);
              ma_m[1] = *(const uint2*)((const bf16_t*)g.mask + orow);
              if (yw + MF == g.H) ma_m[2] = *(const uint2*)((const bf16_t*)g.mask + orow + (int64_t)LWa * g.ldo);
            }
          }
        }
        f32x4 acc[NBW][MF];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
          for (int j = 0; j < MF; ++j) acc[nb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2)) {
          // NIT = CPW * KW window iterations (one 32-channel chunk x one filter column each).  The window ROLLS: row w of
          // the next iteration is fetched as soon as the last MFMA that reads row w of this one has been issued, so the
          // LDS latency of every fragment sits behind >= MF * (KH - 1) MFMAs and a wave that is alone on its SIMD (its
          // partner staging) still keeps the matrix pipe fed
          constexpr int NIT = CPW * KW;
          bf16x8 win[WIN];
#pragma unroll
          for (int j = 0; j < WIN; ++j) win[j] = *(const bf16x8*)(sRing + abase[j]);
          if constexpr (C::TP) {
            // fragment w = input rows qw+w, qw+w+1 x filter columns 2kx, 2kx+1 (one tap per lane quarter): it is tap-row pair gq of output row w - 2 gq
#pragma unroll
            for (int kx = 0; kx < C::KWG; ++kx) {
#pragma unroll
              for (int w = 0; w < WIN; ++w) {
#pragma unroll
                for (int gq = 0; gq < C::KHG; ++gq) {
                  const int j = w - 2 * gq;
                  if (j < 0 || j >= MF) continue;
#pragma unroll
                  for (int nb = 0; nb < NBW; ++nb)
                    acc[nb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wr[nb][0][kx][gq], win[w], acc[nb][j], 0, 0, 0);
                }
                if (kx + 1 < C::KWG) win[w] = *(const bf16x8*)(sRing + abase[w] + (kx + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          } else
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int cc = it / KW, kx = it % KW;
            const int ncc = (it + 1) / KW, nkx = (it + 1) % KW;
#pragma unroll
            for (int w = 0; w < WIN; ++w) {
#pragma unroll
              for (int ky = 0; ky < KH; ++ky) {
                const int j = w - ky;
                if (j < 0 || j >= MF) continue;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                  acc[nb][j] = C::MA ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(win[w], Wr[nb][cc][kx][ky], acc[nb][j], 0, 0, 0)      // X = pixels x W^T
                                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wr[nb][cc][kx][ky], win[w], acc[nb][j], 0, 0, 0);
              }
              if (it + 1 < NIT) win[w] = *(const bf16x8*)(sRing + abase[w] + ncc * 2 * C::PLB + nkx * 32);
              // pin the order: left alone, hipcc sinks every fetch to just in front of its first use (one register
              // for all of them) and the wave stalls on the LDS latency of each fragment
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        SV_STAMP(t_mfma);
        if (C::KS == 2 && !(dbg & 8)) {
          // K halves meet: the odd wave of the pair publishes its partial sums, the even one adds them
          char* slot = sEx + (pair * C::EXS + (strips & (C::EXS - 1))) * C::EXF + lane * 16;
          if (ks == 1) {
            while (strips - __hip_atomic_load(&sFlag[pair * 2 + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= C::EXS)
              __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
              for (int j = 0; j < MF; ++j) *(f32x4*)(slot + (nb * MF + j) * 1024) = acc[nb][j];
            __hip_atomic_store(&sFlag[pair * 2], strips + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            while (__hip_atomic_load(&sFlag[pair * 2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= strips)
              __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
              for (int j = 0; j < MF; ++j) acc[nb][j] += *(const f32x4*)(slot + (nb * MF + j) * 1024);
            __hip_atomic_store(&sFlag[pair * 2 + 1], strips + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        ++strips;
        SV_STAMP(t_exch);
        if constexpr (C::MA) {
          // X rows (bf16, as the hi-res gradient would go to HBM): lane = (channel lane & 15 of the wave's block, pixel group lane >> 4), 4 pixels
          uint2 ma_x[MF];
#pragma unroll
          for (int j = 0; j < MF; ++j) {
            const bf16x2 lo2 = __builtin_convertvector((f32x2){acc[0][j][0], acc[0][j][1]}, bf16x2), hi2 = __builtin_convertvector((f32x2){acc[0][j][2], acc[0][j][3]}, bf16x2);
            ma_x[j] = make_uint2(__builtin_bit_cast(uint32_t, lo2), __builtin_bit_cast(uint32_t, hi2));
          }
          if (si == 0 && C::SPW > 1) {
            *(uint4*)(sMaw + tid * 96 + 64) = make_uint4(ma_x[0].x, ma_x[0].y, ma_x[1].x, ma_x[1].y);
            *(uint4*)(sMaw + tid * 96 + 80) = make_uint4(ma_x[2].x, ma_x[2].y, ma_x[3].x, ma_x[3].y);
          }
          if (si == C::SPW - 1 && !(dbg & 4)) {
            const bool top = yw == 0, bot = yw + MF == g.H;   // wave-uniform (ADJ units are whole images)
            bf16x8 ax[MF];
            if constexpr (C::SPW > 1) {
              const uint4 x01 = *(const uint4*)(sMaw + tid * 96 + 64), x23 = *(const uint4*)(sMaw + tid * 96 + 80);     // strip 0's rows
              const uint2 x0[MF] = {make_uint2(x01.x, x01.y), make_uint2(x01.z, x01.w), make_uint2(x23.x, x23.y), make_uint2(x23.z, x23.w)};
#pragma unroll
              for (int j = 0; j < MF; ++j) ax[j] = __builtin_bit_cast(bf16x8, make_uint4(x0[j].x, x0[j].y, ma_x[j].x, ma_x[j].y));
            } else {
#pragma unroll
              for (int j = 0; j < MF; ++j) ax[j] = __builtin_bit_cast(bf16x8, make_uint4(ma_x[j].x, ma_x[j].y, 0u, 0u));
            }
            const bf16x8 bq = *(const bf16x8*)(sMaw + tid * 96), bt = *(const bf16x8*)(sMaw + tid * 96 + 16);
            f32x4 ma_p = *(const f32x4*)(sMaw + tid * 96 + 32), ma_q = *(const f32x4*)(sMaw + tid * 96 + 48);
            const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int LHa = g.H >> 1, LWa = g.W >> 1, i0 = yw >> 1;     // this step completes low-res rows i0 - 1 and i0 (and i0 + 1 at the bottom)
            const int64_t orow = ((int64_t)b * LHa + i0) * LWa * g.ldo + (int64_t)m * g.ldo + nbg * 16 + kq * 4;
            auto emit = [&](const f32x4& zv, int di, const uint2& mv) {
              bf16_t res[4];
              const uint32_t mw[2] = {mv.x, mv.y};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t h = (mw[e >> 1] >> ((e & 1) * 16)) & 0xffffu;      // bf16 bits of the low-res activation: > 0 <=> sign clear and not zero
                res[e] = (bf16_t)((!g.mask || (!(h & 0x8000u) && (h & 0x7fffu))) ? zv[e] : 0.f);
              }
              if (!(mute0 && s == 0) && (C::WIDTH == 32 || m < C::WIDTH / 2)) *(uint2*)((bf16_t*)g.out + orow + (int64_t)di * LWa * g.ldo) = *(const uint2*)res;
            };
            // hi-res row Y = 2r feeds low-res rows max(r - 1, 0) (.25) and r (.75); Y = 2r + 1 feeds r (.75) and min(r + 1, LH - 1) (.25)
            if (!top) {
              ma_p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], bq, ma_p, 0, 0, 0);
              emit(ma_p, -1, ma_m[0]);
            } else ma_q = z4;
            ma_q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], bt, ma_q, 0, 0, 0);
            if (top) ma_q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], bq, ma_q, 0, 0, 0);
            ma_q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[1], bt, ma_q, 0, 0, 0);
            ma_q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[2], bq, ma_q, 0, 0, 0);
            emit(ma_q, 0, ma_m[1]);
            ma_p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[1], bq, z4, 0, 0, 0);
            ma_p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[2], bt, ma_p, 0, 0, 0);
            ma_p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[3], bt, ma_p, 0, 0, 0);
            if (bot) {
              ma_p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[3], bq, ma_p, 0, 0, 0);
              emit(ma_p, 1, ma_m[2]);
            } else ma_q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[3], bq, z4, 0, 0, 0);
            *(f32x4*)(sMaw + tid * 96 + 32) = ma_p;
            *(f32x4*)(sMaw + tid * 96 + 48) = ma_q;
          }
        } else if constexpr (C::ADJ) {
          // the hi-res gradient row goes to the out ring (bf16, as it would go to HBM); adjoint_rows picks it up
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
              bf16_t pk[4] = {(bf16_t)acc[nb][j][0], (bf16_t)acc[nb][j][1], (bf16_t)acc[nb][j][2], (bf16_t)acc[nb][j][3]};
              *(uint2*)(sOut + ((obase + yw + j) % C::ORR) * C::OROWB + (x0 + m) * (C::N * 2) + ((nbg * NBW + nb) * 16 + kq * 4) * 2) = *(const uint2*)pk;
            }
        } else if constexpr (C::CLS) {
          if (!(dbg & 4)) {
            // a 16-column block = 16 channels of ONE parity class: pixel (2 (yw + j) + ph, 2 (x0 + m) + pw), channels chb + 4 kq ..
            // All NBW * MF mask loads go out before the first is used (one after the other they were 47 % of the launch: in-kernel stamps).
            constexpr int CN = C::N / 4;
            int64_t o[NBW][MF];
            uint2 mv[NBW][MF];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
              const int col = (nbg * NBW + nb) * 16, cls = col / CN, chb = col - cls * CN;
              const int64_t pix0 = ((int64_t)b * 2 * g.H + 2 * yw + (cls >> 1)) * (2 * g.W) + 2 * (x0 + m) + (cls & 1);
#pragma unroll
              for (int j = 0; j < MF; ++j) {
                o[nb][j] = (pix0 + (int64_t)j * 4 * g.W) * g.ldo + chb + kq * 4;
                mv[nb][j] = g.mask ? *(const uint2*)((const bf16_t*)g.mask + o[nb][j]) : make_uint2(0x3f803f80u, 0x3f803f80u);
              }
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
              for (int j = 0; j < MF; ++j) {
                bf16_t pk[4] = {(bf16_t)acc[nb][j][0], (bf16_t)acc[nb][j][1], (bf16_t)acc[nb][j][2], (bf16_t)acc[nb][j][3]};
                const uint32_t mw[2] = {mv[nb][j].x, mv[nb][j].y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const uint32_t h = (mw[e >> 1] >> ((e & 1) * 16)) & 0xffffu;      // bf16 bits: > 0 <=> sign clear and not zero
                  if ((h & 0x8000u) || !(h & 0x7fffu)) pk[e] = (bf16_t)0.f;
                }
                *(uint2*)((bf16_t*)g.out + o[nb][j]) = *(const uint2*)pk;
              }
          }
        } else
        if ((C::KS == 1 || ks == 0) && !(dbg & 4)) {
          // D rows = channels: a lane holds 4 consecutive channels of pixel x0 + m (PAIR: pixel m & 7 of image 2b + (m >> 3))
          const int bi = C::PAIR ? 2 * b + (m >> 3) : b;
          const int64_t o0 = (((int64_t)bi * (g.os * g.H) + g.os * yw + g.ooy) * (g.os * g.W) + g.os * (C::PAIR ? (m & 7) : x0 + m) + g.oox) * g.ldo + kq * 4;
          bf16_t* ob = (bf16_t*)g.out + o0;
          const bool st_on = !C::PAIR || bi < g.B;
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = acc[nb][j][e] + bv[nb][e];
                if (g.act == SV_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
              }
              bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
              const int64_t oo = (int64_t)j * g.os * (g.os * g.W) * g.ldo + (nbg * NBW + nb) * 16;
              if (C::PAIR && g.mask && st_on) {                 // a ReLU mask on the output itself (d2's input gradient)
                const uint2 mv = *(const uint2*)((const bf16_t*)g.mask + o0 + oo);
                const uint32_t mw[2] = {mv.x, mv.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const uint32_t h = (mw[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                  if ((h & 0x8000u) || !(h & 0x7fffu)) pk[e] = (bf16_t)0.f;
                }
              }
              if (st_on) *(uint2*)(ob + oo) = *(const uint2*)pk;
            }
        }
        SV_STAMP(t_epi);
      }
      SV_STAMP(t_other);
      if constexpr (STAG) {
        if (job && half == 1) stage_rows<C>(gj, jb, jY, jrows, qn, sRing, tid - 256, 256, 1, 2);
      } else if constexpr (!C::UPS || C::MB) {
        // the DMAs of this step are older than its stores: wait for everything but the stores (ADJ: this step's only
        // global stores are the adjoint's, issued before the DMAs)
        if (C::MA && !(dbg & 4)) {
          // the step's DMAs are older than its low-res stores: 1 store at the top of an image, 3 at the bottom, 2 otherwise
          if (mute0 && s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a warm-up step stores nothing)
          else if (y0 == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          else if (y0 + STEP == g.H) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else
        if (!C::ADJ && (C::KS == 1 || ks == 0) && !(dbg & 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#ifdef SV_MB_STAMP_SPLIT
      SV_STAMP(t_bar);                                       // (diagnostic: the end-of-step vmcnt wait booked with the barrier)
#else
      SV_STAMP(t_stage);
#endif
      // the staged rows are in place (LDS writes drained), this step's reads are done.  A raw barrier: __syncthreads()
      // would also wait for this step's global STORES (vmcnt(0)) and expose their HBM latency once per step
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      SV_STAMP(t_bar);
      q0 = more ? q0 + STEP : (C::PRE ? qn : 0);             // a new unit starts at its own first window
      if (q0 >= R) q0 -= R;
    }
    if constexpr (C::ADJ && !C::MA) {                        // the image's last low-res rows (the step loop ended behind a barrier)
      const int e_end = band + 1 == g.bands ? (g.H >> 1) - 1 : ((band + 1) * g.band_rows - 3) >> 1;     // (rows below it belong to the next band)
      if (!(dbg & 4) && e_end >= emitted) adjoint_rows<C>(g, b, emitted, e_end, obase, sOut, tid, NT);
      obase = (obase + g.H) % C::ORR;
    }
    if (!C::PRE && has_next && !(dbg & 1)) {                 // no room to prefetch: stage the next unit's first window now
      if constexpr (C::MB) {
        // its raw rows were fetched during the last step and waited for at its end
        mb_blend<C>(mgn, mb_nY, STEP + KH - 1, 0, sRaw, mbl, mbw);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        continue;
      } else
      if constexpr (C::UPS) stage_rows<C>(gn, rn / gn.bands, band_row0(gn, rn % gn.bands) + gn.y_lo, STEP + KH - 1, 0, sRing, tid, NT, 0, 1);
      else {
        stage_rows_dma<C>(gn, rn / gn.bands, band_row0(gn, rn % gn.bands) + gn.y_lo, STEP + KH - 1, 0, sRing, wave, C::WAVES, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#ifdef SV_DEBUG_KNOBS
  if (mg.stamps && lane == 0) {
    unsigned long long* o = mg.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = t_stage; o[1] = t_mfma; o[2] = t_exch; o[3] = t_epi; o[4] = t_bar; o[5] = t_other;
  }
#endif
}

template <typename C>
static int launch_row(const RowConvArgs* a, int n, hipStream_t st) {
  RowConvMulti m;
  for (int i = 0; i < 8; ++i) m.a[i] = a[i < n ? i : 0];
  m.units_per_prob = (C::PAIR ? (a[0].B + 1) / 2 : a[0].B) * a[0].bands;
  m.units = n * m.units_per_prob;
  m.dbg = 0;
  m.stamps = nullptr;
#ifdef SV_DEBUG_KNOBS
  static const int dbg = getenv("SV_RC_DBG") ? atoi(getenv("SV_RC_DBG")) : 0;
  m.dbg = dbg;
  static unsigned long long* stamp_buf = nullptr;
  static const bool stamp = getenv("SV_RC_STAMP") != nullptr;
  if (stamp && !stamp_buf) (void)hipMalloc(&stamp_buf, 512 * 8 * 8 * 8);
  if (stamp) { m.stamps = stamp_buf; (void)hipMemsetAsync(stamp_buf, 0, 512 * 8 * 8 * 8, st); }
#endif
  static const int wgs_env = getenv("SV_RC_WGS") ? atoi(getenv("SV_RC_WGS")) : 0;
  const int wgs_max = wgs_env ? wgs_env : 256 * (C::WAVES == 8 ? 1 : 2);                 // one 8-wave or two 4-wave (or two 2-wave, 512-register) workgroups per CU
  int grid = m.units < wgs_max ? m.units : wgs_max;
  if (C::PAIR) grid = grid / n * n;                          // a workgroup stays on one problem (see the kernel's unit decode)
  sv_ensure_dynamic_lds((const void*)row_conv_kernel<C>, C::LDS);
  hipLaunchKernelGGL((row_conv_kernel<C>), dim3(grid), dim3(C::NT), C::LDS, st, m);
  SV_LAUNCH_CHECK();
#ifdef SV_DEBUG_KNOBS
  if (m.stamps) {                                            // diagnostic build only: per-phase cycle shares of the launch
    static int shown = 0;
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h(512 * 8 * 8);
    (void)hipMemcpy(h.data(), m.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    if (shown++ % 16 == 15) {
      const char* nm[6] = {"stage", "mfma", "exch", "epi", "barrier", "other"};
      for (int hf = 0; hf < 2; ++hf) {
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < grid; ++w)
          for (int wv = hf * (C::WAVES / 2); wv < (hf + 1) * (C::WAVES / 2); ++wv)
            for (int k = 0; k < 6; ++k) s[k] += (double)h[((size_t)w * 8 + wv) * 8 + k];
        fprintf(stderr, "row_conv stamps half %d (cycles per wave):", hf);
        for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.0f", nm[k], s[k] / (grid * (C::WAVES / 2)));
        fprintf(stderr, "\n");
      }
    }
  }
#endif
  return SV_OK;
}

//                  KH KW CIN   N   W MF NBW KS XG RG UPS  WAVES ADJ
using RC_d4f  = RowCfg<6, 6, 64, 32, 32, 4, 1, 2, 1, 1, true, 4>;          // d4 forward   (K 2304: pairs split the two 32-channel chunks)
using RC_d4fm = RowCfg<6, 6, 64, 32, 32, 4, 1, 2, 1, 1, true, 4, false, false, false, false, false, true>;   //   ... with the resize on the matrix pipe (MB; whole images: bands == 1)
using RC_d4fw = RowCfg<6, 6, 64, 32, 32, 4, 1, 1, 1, 1, true, 2, false, false, false, false, false, true>;   //   ... one wave per SIMD, whole K per wave (no exchange)
using RC_d4g  = RowCfg<6, 6, 32, 64, 32, 4, 1, 1, 1, 1, false, 4>;         // d4 input gradient (K 1152)
using RC_d4ga = RowCfg<6, 6, 32, 64, 32, 4, 1, 1, 1, 1, false, 4, true>;   //   ... fused with the resize adjoint
using RC_d4gm = RowCfg<6, 6, 32, 64, 32, 4, 1, 1, 1, 1, false, 4, true, false, false, false, false, false, true>;   //   ... the adjoint on the matrix pipe (MA)
using RC_d3f  = RowCfg<4, 4, 128, 64, 16, 4, 1, 2, 1, 1, true>;            // d3 forward   (K 2048)
using RC_d3fm = RowCfg<4, 4, 128, 64, 16, 4, 1, 2, 1, 1, true, 8, false, false, false, false, false, true>;   //   ... with the resize on the matrix pipe (MB)
using RC_d3g  = RowCfg<4, 4, 64, 128, 16, 4, 1, 1, 1, 1, false>;           // d3 input gradient (K 1024)
using RC_d3ga = RowCfg<4, 4, 64, 128, 16, 4, 1, 1, 1, 1, false, 8, true>;
using RC_d3gm = RowCfg<4, 4, 64, 128, 16, 4, 1, 1, 1, 1, false, 8, true, false, false, false, false, false, true>;   //   ... the adjoint on the matrix pipe (MA)
using RC_d5g  = RowCfg<6, 6, 8, 32, 64, 4, 2, 1, 4, 1, false, 4>;          // d5 input gradient (8-channel pixels: four taps per K step)
using RC_d5ga = RowCfg<6, 6, 8, 32, 64, 4, 2, 1, 4, 1, false, 4, true>;
// e2 input gradient, merged parity classes (K 576, 4 x 32 columns): 4-wave workgroups, a wave = the two column blocks of one class.
// Measured in the step (2 x 512 images): tile kernel 0.103 ms; 8 waves x one block 0.082; this 0.075; this with 8 rows per step 0.098
using RC_e2g  = RowCfg<3, 3, 64, 128, 16, 4, 2, 1, 1, 1, false, 4, false, true>;
// d2 forward / input gradient (k 4, 128 -> 128 channels on the 8 x 8 grid, K 2048): image PAIRS per strip; its 512 KB of weights do not fit one CU's
// registers, so every problem runs as two output-channel halves (N 64: the d3-forward register budget) -- four problems per launch
using RC_d2   = RowCfg<4, 4, 128, 64, 16, 4, 1, 2, 1, 1, false, 8, false, false, false, true>;
// e3 forward (k 4, stride 2, 64 -> 128 channels, 16 x 16 -> 8 x 8): space-to-depth with shifted blocks (K 1024) on image pairs
using RC_e3f  = RowCfg<2, 2, 256, 128, 16, 4, 1, 1, 1, 1, false, 8, false, false, true, true>;
// e3 input gradient (k 4, stride 2): its four parity classes are four 2 x 2 stride-1 problems over the 8 x 8 dY grid with their own windows
// (K 512, 64 columns each, class -> sub-pixel store with the ReLU mask): eight problems per launch (classes x networks) on image pairs
using RC_e3g  = RowCfg<2, 2, 128, 64, 16, 4, 1, 1, 1, 1, false, 4, false, false, false, true, true>;
// e2 forward (k 6, stride 2, 32 -> 64 channels) as a 3 x 3 stride-1 conv over the space-to-depth input (K 1152)
using RC_e2f  = RowCfg<3, 3, 128, 64, 16, 4, 1, 1, 1, 1, false, 4, false, false, true>;
// e1 forward (k 6, stride 2, 8-channel padded RGB -> 32 channels) the same way: 32 s2d channels = ONE K chunk per tap (K 288)
using RC_e1f  = RowCfg<3, 3, 32, 32, 32, 4, 1, 1, 2, 1, false, 4, false, false, true>;

}  // namespace

// n (1 or 2: the x / x-hat twins) tap-GEMM problems of identical geometry on the row-ring kernel; SV_E_UNSUPPORTED when
// the shape has no instantiation (the caller falls back to the tile kernel)
static int row_plan(const TapGemmArgs* t, int n, int dtype, RowConvArgs* a, int* nprob = nullptr) {      // a[4]; -> instantiation id (and the number of kernel problems), or SV_E_UNSUPPORTED
  static const bool off = getenv("SV_NO_ROWCONV") != nullptr;          // A/B: the tile kernel for every layer
  if (off || dtype != SV_BF16 || n < 1 || n > 8) return SV_E_UNSUPPORTED;
  int cfg = -1;
  if (nprob) *nprob = n;
  for (int k = 0; k < 8; ++k) { a[k].os = 1; a[k].ooy = 0; a[k].oox = 0; }
  for (int i = 0; i < n; ++i) {
    const TapGemmArgs& p = t[i];
    if (p.lOY < 0 || p.lOX < 0) return SV_E_UNSUPPORTED;
    if (n > 2 && !(getenv("SV_RC_E3G") && p.S == 1 && p.OS == 2 && p.ntaps == 4 && !p.cls_n)) return SV_E_UNSUPPORTED;   // more than the twins: only e3's class problems
    static const bool no_cls = getenv("SV_RC_NO_CLS") != nullptr;     // A/B: merged parity classes on the tile kernel
    static const bool no_s2d = getenv("SV_RC_NO_S2D") != nullptr;     // A/B: the stride-2 forward on the tile kernel
    static const bool no_pair_s2d = getenv("SV_RC_NO_E3") != nullptr; // A/B: e3's forward on the tile kernel
    static const int e3_min = getenv("SV_RC_E3_MIN") ? atoi(getenv("SV_RC_E3_MIN")) : 256;
    const bool cls = p.cls_n > 0;
    if (p.S == 2 && p.SX == 2 && !cls) {                               // stride-2 forward: the space-to-depth form
      if (no_s2d || p.OS != 1 || p.splitk != 1 || p.d2s || p.out_f32 || p.ooy || p.oox || p.mask || p.adj || p.ups) return SV_E_UNSUPPORTED;
      const int OY = 1 << p.lOY, OX = 1 << p.lOX, cin = (1 << p.cl2) * 8;
      if (p.IH != 2 * OY || p.IW != 2 * OX || p.OHF != OY || p.OWF != OX || OY % 4) return SV_E_UNSUPPORTED;
      int c = -1;
      if (cin == 32 && p.N == 64 && OX == 16) c = 9;                   // e2
      else if (cin == 8 && p.N == 32 && OX == 32) c = 10;              // e1
      // e3 (k 4): image pairs per strip -- from 256 images per launch (64 images per network are 64 pairs = 64 workgroups: the tile kernel is
      // 0.6 % of the step faster there, 0.2 % / 0.8 % slower at 128 / 256: profiles/r04_b64_sweep3.txt)
      else if (cin == 64 && p.N == 128 && OX == 8 && OY == 8 && !no_pair_s2d && n * (p.M >> (p.lOY + p.lOX)) >= e3_min) c = 12;
      const int kk = c == 12 ? 4 : 6, pad = c == 12 ? 1 : 2;
      if (c < 0 || p.lda != cin || p.ntaps != kk * kk || p.Ktot != kk * kk * cin || p.ldo < p.N) return SV_E_UNSUPPORTED;
      for (int q = 0; q < kk * kk; ++q)                                // the layer's own order: y-major
        if (p.dy[q] != q / kk - pad || p.dx[q] != q % kk - pad) return SV_E_UNSUPPORTED;
      if (i && cfg != c) return SV_E_UNSUPPORTED;
      cfg = c;
      RowConvArgs& r = a[i];
      r.A = p.A; r.Wt = p.Wt; r.bias = p.bias; r.out = p.out; r.mask = nullptr;
      r.B = p.M >> (p.lOY + p.lOX); r.H = OY; r.W = OX;
      r.lda = p.lda; r.ldo = p.ldo; r.Ktot = p.Ktot; r.act = p.act;
      r.y_lo = c == 12 ? 0 : -1; r.x_lo = c == 12 ? 0 : -1;
      if (c == 12) { r.W = 8; r.bands = 1; r.band_rows = 8; if (i && r.B != a[0].B) return SV_E_UNSUPPORTED; continue; }
      int bands = 1;
      while (n * r.B * bands < 512 && OY / (bands * 2) >= 4 && (OY / (bands * 2)) % 4 == 0) bands *= 2;
      r.bands = bands; r.band_rows = OY / bands;
      if (i && (r.B != a[0].B || r.H != a[0].H || r.bands != a[0].bands)) return SV_E_UNSUPPORTED;
      continue;
    }
    // (measured, 2 x 512 images: 0.050 ms here against 0.044 on the tile kernel -- eight problems of two 4-row steps per image pair are all
    // prologue --, so only SV_RC_E3G=1 sends it here)
    static const bool e3g = getenv("SV_RC_E3G") != nullptr;
    if (e3g && !cls && p.S == 1 && p.SX == 1 && p.OS == 2 && p.ntaps == 4 && p.N == 64 && p.lOX == 3 && p.lOY == 3 && !p.ups && !p.adj) {
      const int cin = (1 << p.cl2) * 8;
      if (p.splitk != 1 || p.d2s || p.out_f32 || p.bias || cin != 128 || p.lda != 128 || p.Ktot != 4 * 128 || p.ldo < 64) return SV_E_UNSUPPORTED;
      if (p.IH != 8 || p.IW != 8 || p.OHF != 16 || p.OWF != 16 || (unsigned)p.ooy > 1u || (unsigned)p.oox > 1u) return SV_E_UNSUPPORTED;
      for (int q = 0; q < 4; ++q)                                      // the class's own order: y-major, offsets descending
        if (p.dy[q] != p.dy[0] - q / 2 || p.dx[q] != p.dx[0] - q % 2) return SV_E_UNSUPPORTED;
      if (i && cfg != 13) return SV_E_UNSUPPORTED;
      cfg = 13;
      RowConvArgs& r = a[i];
      r.A = p.A; r.Wt = p.Wt; r.bias = nullptr; r.out = p.out; r.mask = p.mask;
      r.B = p.M >> 6; r.H = 8; r.W = 8;
      r.lda = p.lda; r.ldo = p.ldo; r.Ktot = p.Ktot; r.act = p.act;
      r.y_lo = p.dy[0] - 1; r.x_lo = p.dx[0] - 1;
      r.bands = 1; r.band_rows = 8;
      r.os = 2; r.ooy = p.ooy; r.oox = p.oox;
      if (i && r.B != a[0].B) return SV_E_UNSUPPORTED;
      continue;
    }
    if (n > 2) return SV_E_UNSUPPORTED;                               // every other form: one problem or the x / x-hat twins
    static const bool no_pair = getenv("SV_RC_NO_PAIR") != nullptr;   // A/B: the 8 x 8-grid layer d2 on the tile kernel
    // (measured, 2 x 512 images: forward 0.053 -> 0.048 ms; the input gradient 0.056 -> 0.058 -- but up to 2 x 256 images it is the faster
    //  form: the step -1.1 % / -0.3 % / -0.6 % at 64 / 128 / 256 images per network (profiles/r04_b64_sweep3.txt).  SV_RC_PAIR_DGRAD=1 / 0 forces it)
    static const int pair_dgrad_env = getenv("SV_RC_PAIR_DGRAD") ? atoi(getenv("SV_RC_PAIR_DGRAD")) : -1;
    const bool pair_dgrad = pair_dgrad_env >= 0 ? pair_dgrad_env != 0 : n * (p.M >> (p.lOY + p.lOX)) <= 512;
    if (!no_pair && !cls && p.S == 1 && p.SX == 1 && p.OS == 1 && p.lOX == 3 && p.lOY == 3 && p.ntaps == 16 && p.N == 128 && !p.ups && !p.adj && (!p.mask || pair_dgrad)) {
      // d2 forward / input gradient: image pairs per strip, two output-channel halves per problem (RC_d2)
      const int cin = (1 << p.cl2) * 8;
      if (p.splitk != 1 || p.d2s || p.out_f32 || p.ooy || p.oox || cin != 128 || p.lda != 128 || p.Ktot != 16 * 128 || p.ldo < 128) return SV_E_UNSUPPORTED;
      if (p.IH != 8 || p.IW != 8 || p.OHF != 8 || p.OWF != 8) return SV_E_UNSUPPORTED;
      for (int q = 0; q < 16; ++q)                                     // x-major, y-minor full grid
        if (p.dy[q] != p.dy[0] + q % 4 || p.dx[q] != p.dx[0] + q / 4) return SV_E_UNSUPPORTED;
      if (i && cfg != 11) return SV_E_UNSUPPORTED;
      cfg = 11;
      for (int h = 0; h < 2; ++h) {
        RowConvArgs& r = a[2 * i + h];
        r.A = p.A; r.Wt = (const bf16_t*)p.Wt + (int64_t)h * 64 * p.Ktot; r.bias = p.bias ? p.bias + 64 * h : nullptr;
        r.out = (bf16_t*)p.out + 64 * h; r.mask = p.mask ? (const bf16_t*)p.mask + 64 * h : nullptr;
        r.B = p.M >> 6; r.H = 8; r.W = 8;
        r.lda = p.lda; r.ldo = p.ldo; r.Ktot = p.Ktot; r.act = p.act;
        r.y_lo = p.dy[0]; r.x_lo = p.dx[0];
        r.bands = 1; r.band_rows = 8;
      }
      if (i && a[2 * i].B != a[0].B) return SV_E_UNSUPPORTED;
      if (nprob) *nprob = 2 * n;
      continue;
    }
    if (cls && (no_cls || p.adj)) return SV_E_UNSUPPORTED;
    if (p.S != 1 || p.SX != 1 || p.OS != (cls ? 2 : 1) || p.splitk != 1 || p.d2s || p.out_f32 || p.ooy || p.oox) return SV_E_UNSUPPORTED;
    if (p.mask && !p.adj && !cls) return SV_E_UNSUPPORTED;             // a ReLU mask on the output itself: tile kernel
    const int OY = 1 << p.lOY, OX = 1 << p.lOX;
    if (OY != p.IH || OX != p.IW || p.OHF != OY * p.OS || p.OWF != OX * p.OS) return SV_E_UNSUPPORTED;
    const int cin = (1 << p.cl2) * 8;
    if (p.lda != cin || p.Ktot != p.ntaps * cin) return SV_E_UNSUPPORTED;
    int kh = 1, kw = 1;
    if (cls) {                                                         // the classes' own order: y-major, offsets descending from +1
      kh = kw = 3;
      if (p.ntaps != 9 || p.bias) return SV_E_UNSUPPORTED;
      for (int q = 0; q < 9; ++q)
        if (p.dy[q] != 1 - q / 3 || p.dx[q] != 1 - q % 3) return SV_E_UNSUPPORTED;
    } else {
      while (kh < p.ntaps && p.dx[kh] == p.dx[0]) ++kh;
      if (p.ntaps % kh) return SV_E_UNSUPPORTED;
      kw = p.ntaps / kh;
      for (int q = 0; q < p.ntaps; ++q)                                // x-major, y-minor full grid
        if (p.dy[q] != p.dy[0] + q % kh || p.dx[q] != p.dx[0] + q / kh) return SV_E_UNSUPPORTED;
    }
    int c = -1;
    if (cls) { if (cin == 64 && p.N == 128 && p.cls_n == 32 && OX == 16) c = 8; }
    else
    if (kh == 6 && kw == 6 && cin == 64 && p.N == 32 && OX == 32 && p.ups && !p.adj) c = 0;
    else if (kh == 6 && kw == 6 && cin == 32 && p.N == 64 && OX == 32 && !p.ups) c = p.adj ? 4 : 1;
    else if (kh == 4 && kw == 4 && cin == 128 && p.N == 64 && OX == 16 && p.ups && !p.adj) c = 2;
    else if (kh == 4 && kw == 4 && cin == 64 && p.N == 128 && OX == 16 && !p.ups) c = p.adj ? 5 : 3;
    else if (kh == 6 && kw == 6 && cin == 8 && p.N == 32 && OX == 64 && !p.ups) c = p.adj ? 7 : 6;
    if (c < 0 || (i && c != cfg)) return SV_E_UNSUPPORTED;
    cfg = c;
    const int step = 4;
    if (OY % step || p.ldo < (cls ? p.cls_n : p.N)) return SV_E_UNSUPPORTED;
    RowConvArgs& r = a[i];
    r.A = p.A; r.Wt = p.Wt; r.bias = p.bias; r.out = p.out; r.mask = (p.adj || cls) ? p.mask : nullptr;
    r.B = p.M >> (p.lOY + p.lOX); r.H = OY; r.W = OX;
    r.lda = p.lda; r.ldo = p.ldo; r.Ktot = p.Ktot; r.act = p.act;
    r.y_lo = cls ? -1 : p.dy[0]; r.x_lo = cls ? -1 : p.dx[0];
    // small batches: cut the images into row bands until there is a unit of work for every workgroup slot (not with the
    // fused adjoint: its low-res rows straddle band edges)
    const bool four = c == 0 || c == 1 || c == 4 || c == 6 || c == 7 || c == 8;
    // (fused adjoint, round 4: bands of >= two steps, every band but the first preceded by a warm-up step -- RowConvArgs::lead; 64 images per
    //  network used to leave half of the CUs without a workgroup in the three decoder input gradients: serial dgrad.d5 47 -> 26 us, d4 40 -> 32,
    //  step 0.640 -> 0.630 ms; a second workgroup per CU at 128 images per network does not pay for its warm-up steps: +0.3 %.
    //  SV_RC_ADJ_BANDS=0: whole images)
    static const bool adj_bands = getenv("SV_RC_ADJ_BANDS") == nullptr || atoi(getenv("SV_RC_ADJ_BANDS")) != 0;
    int bands = 1;
    while ((!p.adj || adj_bands) && n * r.B * bands < ((four && !p.adj) ? 512 : 256) && OY / (bands * 2) >= (p.adj ? 2 * step : step) && (OY / (bands * 2)) % step == 0) bands *= 2;
    r.bands = bands; r.band_rows = OY / bands; r.lead = p.adj && bands > 1 ? step : 0;
    if (i && (r.B != a[0].B || r.H != a[0].H || r.bands != a[0].bands)) return SV_E_UNSUPPORTED;
  }
  // The upsampled forward layers: in the training step at B = 512 the LDS-tile kernel is as fast (fwd.d4 0.140 vs 0.141 ms,
  // fwd.d3 0.077 vs 0.081: their blend staging costs this kernel what the weight streaming costs that one); at small
  // batches this kernel wins clearly (128 images: d3 18 vs 33 us).  SV_RC_FWD=1 / 0 forces it on / off.
  static const int fwd_mode = getenv("SV_RC_FWD") ? atoi(getenv("SV_RC_FWD")) : -1;
  // (Round 3, re-measured under the two-side-stream schedule: d4 forward 0.137 ms here against 0.146 on the tile kernel, d3 still equal: d4 always here.)
  // (Round 4: with the resize on the matrix pipe -- RowCfg::MB, whole images per unit -- d3 stays here at every size.)
  static const bool no_mb3 = getenv("SV_RC_NO_MB") != nullptr || getenv("SV_RC_NO_MB3") != nullptr;
  const bool d3_mb = cfg == 2 && !no_mb3 && a[0].bands == 1 && a[0].H == 16 && a[0].W == 16 && a[0].lda == 128;
  if ((cfg == 0 && fwd_mode == 0) || (cfg == 2 && (fwd_mode == 0 || (fwd_mode < 0 && n * a[0].B > 512 && !d3_mb)))) return SV_E_UNSUPPORTED;
  return cfg;
}

bool svk_row_conv_supported(const TapGemmArgs* t, int n, int dtype) {
  RowConvArgs a[8] = {};
  return row_plan(t, n, dtype, a) >= 0;
}

int svk_row_conv_try(const TapGemmArgs* t, int n, int dtype, hipStream_t st) {
  RowConvArgs a[8] = {};
  const int cfg = row_plan(t, n, dtype, a, &n);
  switch (cfg) {
    case 0: {
      static const bool no_mb = getenv("SV_RC_NO_MB") != nullptr;       // A/B: the VALU blend staging (stage_rows)
      // two waves per SIMD with the K-half exchange (4-wave workgroups, the default) or ONE wave per SIMD holding the whole K in the 512-register
      // file (2-wave workgroups, no exchange: SV_RC_MB_WAVES=2) -- measured equal alone on the chip (136-138 us at 1024 images), the 4-wave
      // form 7 % faster inside the step (0.119 against 0.128 ms)
      static const int mbw = getenv("SV_RC_MB_WAVES") ? atoi(getenv("SV_RC_MB_WAVES")) : 4;
      if (!no_mb && a[0].bands == 1 && a[0].H == 32 && a[0].W == 32 && a[0].lda == 64) return mbw == 2 ? launch_row<RC_d4fw>(a, n, st) : launch_row<RC_d4fm>(a, n, st);
      return launch_row<RC_d4f>(a, n, st);
    }
    case 1: return launch_row<RC_d4g>(a, n, st);
    case 2: {
      static const bool no_mb3 = getenv("SV_RC_NO_MB") != nullptr || getenv("SV_RC_NO_MB3") != nullptr;
      if (!no_mb3 && a[0].bands == 1 && a[0].H == 16 && a[0].W == 16 && a[0].lda == 128) return launch_row<RC_d3fm>(a, n, st);
      return launch_row<RC_d3f>(a, n, st);
    }
    case 3: return launch_row<RC_d3g>(a, n, st);
    case 4: {
      static const bool no_ma = getenv("SV_RC_NO_MA") != nullptr;       // A/B: the adjoint through the LDS out ring (adjoint_rows)
      return no_ma ? launch_row<RC_d4ga>(a, n, st) : launch_row<RC_d4gm>(a, n, st);
    }
    case 5: {
      static const bool no_ma3 = getenv("SV_RC_NO_MA") != nullptr || getenv("SV_RC_NO_MA3") != nullptr;
      return no_ma3 ? launch_row<RC_d3ga>(a, n, st) : launch_row<RC_d3gm>(a, n, st);
    }
    case 6: return launch_row<RC_d5g>(a, n, st);
    case 7: return launch_row<RC_d5ga>(a, n, st);
    case 8: return launch_row<RC_e2g>(a, n, st);
    case 9: return launch_row<RC_e2f>(a, n, st);
    case 10: return launch_row<RC_e1f>(a, n, st);
    case 11: return launch_row<RC_d2>(a, n, st);
    case 12: return launch_row<RC_e3f>(a, n, st);
    case 13: return launch_row<RC_e3g>(a, n, st);
  }
  return SV_E_UNSUPPORTED;
}
