// Weight gradient with LDS-resident tiles (bf16):  dW[t][ci][co] += sum_pixels X[pix + tap t][ci] * dY[pix][co]
//
// A workgroup owns one TAP GROUP (all input channels, all output channels of those taps) and walks
// a strided subset of the spatial tiles.  Per tile it stages, once, the input patch with its halo
// and the matching dY patch in their natural NHWC layout (coalesced 16-B loads); every tap's
// k-major MFMA operand is then a SHIFTED WINDOW of the same LDS patch, read transposed with
// ds_read_b64_tr_b16 (pixels are the MFMA K dimension).  The im2col form (wgrad.hip) re-gathers
// the input once per tap through L2; here the re-use factor is the tap-group size, which is made
// as large as the accumulator registers allow (up to 36 fragments = 144 VGPRs per wave).
// Accumulators stay in registers across all of the workgroup's tiles and are flushed once with
// fp32 atomics.  Waves split the tap group (WT) and/or the input-channel fragments (WC).
//
// LDS layout: pixel records of PS = Cin*2 + 32 bytes (dY: Cout*2 + 32).  With PS/32 odd, eight
// consecutive pixels start on eight different 32-B bank groups, and the MFMA K index is mapped to
// pixels as k = 8g + 4h + q  <->  pixel 16h + 4g + q so that the two lane-halves of each
// transposed read touch 8 consecutive pixels: conflict-free.
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"

#ifndef SV_WT_PF
#define SV_WT_PF 4      // LDS prefetch depth (fragments) of the transposed A-operand reads
#endif

__device__ __forceinline__ short4_t tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

// TPW taps per wave, CIF ci-fragments (16 channels) per wave, COF co-fragments (all of Cout_pad16),
// WT x WC = 4 waves over (taps, ci-fragments); KC = 32-pixel K chunks per tile (BM = 32*KC pixels)
template <int TPW, int CIF, int COF, int WT, int WC, int KC>
__global__ __launch_bounds__(256, 2) void wgrad_tile_kernel(const WgradTileArgs g) {
  static_assert(WT * WC == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sIn = smem;                       // [NB][TIH][TIW] pixels of PS bytes (+ slack)
  char* sDy = smem + g.in_bytes;          // [BM] pixels of YS bytes (+ slack)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wt = wave / WC, wc = wave % WC;
  const int tap0 = blockIdx.y * (TPW * WT) + wt * TPW;      // this wave's first tap
  const int TW = 1 << g.lTW, TH = 1 << g.lTH, NB = 1 << g.lNB;
  const int cpp = 1 << g.cl2;
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3, lr = lane & 15;

  // per-lane LDS byte offsets for the transposed reads: read h of chunk kc -> tile pixel
  // r = 32*kc + 16*h + 4*lg + lq, channel block 4*lp (see the K <-> pixel map above)
  int inb[KC][2], dyb[KC][2];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = kc * 32 + 16 * h + 4 * lg + lq;
      const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
      inb[kc][h] = ((bl * g.TIH + ty * g.S) * g.TIW + tx * g.S) * g.PS + (wc * CIF * 16 + 4 * lp) * 2;
      dyb[kc][h] = r * g.YS + 4 * lp * 2;
    }
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
  float bsum = 0.f;
  const bool do_bias = g.dbias != nullptr && blockIdx.y == 0;
  const int ycols = g.ldy;                          // dY channels per pixel (power of two >= 8)
  const int bcol = tid & (ycols - 1), bgrp = tid / ycols, nbg = 256 / ycols;

  const bf16_t* __restrict__ Ab = (const bf16_t*)g.A;
  const bf16_t* __restrict__ Yb = (const bf16_t*)g.dY;
  // staging geometry: LPR lanes sweep one tile row (no integer division anywhere)
  const int ppr = g.TIW * cpp;                      // 16-B pieces per input-tile row
  const int LPR = ppr > 160 ? 64 : 32, lLPR = ppr > 160 ? 6 : 5;
  const int srow = tid >> lLPR, slane = tid & (LPR - 1), rows_pp = 256 >> lLPR;
  const int nrows = NB * g.TIH;
  const int lycp = g.lycp;                          // log2(16-B pieces per dY pixel)
  const int dy_total = (32 * KC) << lycp;

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx0 = (t % g.tilesX) << g.lTW; t /= g.tilesX;
    const int ty0 = (t % g.tilesY) << g.lTH; t /= g.tilesY;
    const int b0 = t << g.lNB;
    __syncthreads();                                // previous tile fully consumed
    // ---- stage input patch (+halo, zero outside the image)
    {
      const int iy_base = ty0 * g.S + g.y_lo, ix_base = tx0 * g.S + g.x_lo;
      for (int row = srow; row < nrows; row += rows_pp) {
        int bl = 0, iyl = row;
        while (iyl >= g.TIH) { iyl -= g.TIH; ++bl; }
        const int iy = iy_base + iyl, b = b0 + bl;
        const bool rok = b < g.B && (unsigned)iy < (unsigned)g.IH;
        const bf16_t* src = Ab + ((int64_t)(b * g.IH + iy) * g.IW) * g.lda;
        char* drow = sIn + row * g.TIW * g.PS;
        for (int pc0 = slane; pc0 < ppr; pc0 += LPR * 4) {
          uint4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int pc = pc0 + u * LPR;
            const int ixl = pc >> g.cl2, c = pc & (cpp - 1), ix = ix_base + ixl;
            v[u] = make_uint4(0, 0, 0, 0);
            if (pc < ppr && rok && (unsigned)ix < (unsigned)g.IW) v[u] = *(const uint4*)(src + (int64_t)ix * g.lda + c * 8);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int pc = pc0 + u * LPR;
            if (pc < ppr) *(uint4*)(drow + (pc >> g.cl2) * g.PS + (pc & (cpp - 1)) * 16) = v[u];
          }
        }
      }
    }
    // ---- stage dY patch
    for (int q = tid; q < dy_total; q += 256) {
      const int r = q >> lycp, c = q & ((1 << lycp) - 1);
      const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
      const int b = b0 + bl;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b < g.B) v = *(const uint4*)(Yb + ((int64_t)(b * g.OY + ty0 + ty) * g.OX + tx0 + tx) * g.ldy + c * 8);
      *(uint4*)(sDy + r * g.YS + c * 16) = v;
    }
    __syncthreads();
    // ---- MFMA: K = pixels
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      short8_t bfr[COF];
#pragma unroll
      for (int j = 0; j < COF; ++j) {
        const short4_t lo = tr16(sDy + dyb[kc][0] + j * 32), hi = tr16(sDy + dyb[kc][1] + j * 32);
        bfr[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      // A fragments of the (tap, ci-fragment) sequence u = t2*CIF + i, prefetched PF deep so the
      // LDS round trip (~100+ cycles) hides under the MFMAs of earlier fragments
      constexpr int U = TPW * CIF, PF = U < SV_WT_PF ? U : SV_WT_PF;
      short4_t alo[PF], ahi[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        alo[u] = tr16(sIn + inb[kc][0] + tapoff[u / CIF] + (u % CIF) * 32);
        ahi[u] = tr16(sIn + inb[kc][1] + tapoff[u / CIF] + (u % CIF) * 32);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const short4_t lo = alo[u % PF], hi = ahi[u % PF];
        const short8_t af = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (u + PF < U) {
          alo[u % PF] = tr16(sIn + inb[kc][0] + tapoff[(u + PF) / CIF] + ((u + PF) % CIF) * 32);
          ahi[u % PF] = tr16(sIn + inb[kc][1] + tapoff[(u + PF) / CIF] + ((u + PF) % CIF) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch PF fragments ahead of its MFMAs (hipcc sinks it otherwise)
#pragma unroll
        for (int j = 0; j < COF; ++j)
          acc[u / CIF][u % CIF][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[j]), acc[u / CIF][u % CIF][j], 0, 0, 0);
      }
    }
    if (do_bias) {
      for (int r = bgrp; r < 32 * KC; r += nbg) bsum += (float)*(const bf16_t*)(sDy + r * g.YS + bcol * 2);
    }
  }

  // ---- flush: D row = ci (lane>>4)*4+reg within the fragment, col = co lane&15
#pragma unroll
  for (int t2 = 0; t2 < TPW; ++t2) {
    const int tap = tap0 + t2;
    if (tap >= g.ntaps) continue;
#pragma unroll
    for (int i = 0; i < CIF; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int ci = (wc * CIF + i) * 16 + lg * 4 + r4;
        if (ci >= g.Cin_real) continue;
#pragma unroll
        for (int j = 0; j < COF; ++j) {
          const int co = j * 16 + lr;
          if (co < g.N) atomicAdd(g.dW + ((int64_t)(tap * g.Cin_real + ci)) * g.N + co, acc[t2][i][j][r4]);
        }
      }
  }
  if (do_bias) {
    __syncthreads();
    float* red = (float*)smem;
    red[bgrp * ycols + bcol] = bsum;
    __syncthreads();
    if (tid < ycols && tid < g.N) {
      float s = 0.f;
      for (int k = 0; k < nbg; ++k) s += red[k * ycols + tid];
      atomicAdd(g.dbias + tid, s);
    }
  }
}

template <int TPW, int CIF, int COF, int WT, int WC, int KC>
static int launch_wt(const WgradTileArgs& a, int groups, hipStream_t st) {
  const size_t lds = (size_t)a.in_bytes + a.dy_bytes;
  static size_t attr_set = 0;
  if (lds > attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_tile_kernel<TPW, CIF, COF, WT, WC, KC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = lds;
  }
  // resident workgroups per CU by LDS, capped at 3 (accumulator-heavy waves)
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 3) per_cu = 3;
  if (per_cu < 1) per_cu = 1;
  int msplit = (256 * per_cu + groups - 1) / groups;
  if (msplit > a.ntiles) msplit = a.ntiles;
  dim3 grid(msplit, groups), block(256);
  hipLaunchKernelGGL((wgrad_tile_kernel<TPW, CIF, COF, WT, WC, KC>), grid, block, lds, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// Returns SV_E_UNSUPPORTED when the layer shape has no tile instantiation (caller falls back to
// the im2col wgrad).  Shapes: conv layers of the SPLIT-VAE encoder/decoder with few channels and
// many taps, where tap re-use from LDS pays; the wide (Cin = 128) layers stay on the im2col GEMM.
int svk_wgrad_tile(const WgradArgs& w, hipStream_t st) {
  static const bool force_old = getenv("SV_FORCE_IM2COL") != nullptr;
  static const char* only = getenv("SV_WGRAD_TILE_IDS");    // e.g. "0145": restrict to these ids (profiling A/B)
  if (force_old) return SV_E_UNSUPPORTED;
  const int OY = 1 << w.lOY, OX = 1 << w.lOX;
  if (OY * OX < 16 || w.ycols != w.ldy) return SV_E_UNSUPPORTED;
  const int cin = w.Cin_pad, cout = w.ldy, nt = w.ntaps;
  int id = -1, BM = 0, groups = 0;
  if (nt == 36 && cin == 32 && cout == 8) { id = 0; BM = 256; groups = 1; }         // d5
  else if (nt == 36 && cin == 64 && cout == 32) { id = 1; BM = 128; groups = 2; }   // d4
  else if (nt == 16 && cin == 128 && cout == 64) { id = 2; BM = 128; groups = 4; }  // d3
  else if (nt == 16 && cin == 128 && cout == 128) { id = 3; BM = 64; groups = 8; }  // d2
  else if (nt == 16 && cin == 64 && cout == 128) { id = 4; BM = 64; groups = 4; }   // e3
  else if (nt == 36 && cin == 32 && cout == 64) { id = 5; BM = 64; groups = 2; }    // e2
  else if (nt == 36 && cin == 8 && cout == 32) { id = 6; BM = 256; groups = 1; }    // e1
  else return SV_E_UNSUPPORTED;
  if (only && !strchr(only, '0' + id)) return SV_E_UNSUPPORTED;
  if (!only && (id == 2 || id == 3 || id == 4 || id == 5)) return SV_E_UNSUPPORTED;   // measured: im2col GEMM is faster there
  if (OY * OX < BM && (BM % (OY * OX))) return SV_E_UNSUPPORTED;
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
  WgradTileArgs a;
  memset(&a, 0, sizeof(a));
  a.A = w.A; a.dY = w.dY; a.dW = w.dW; a.dbias = w.dbias;
  a.B = B; a.IH = w.IH; a.IW = w.IW; a.lda = w.lda; a.cl2 = w.cl2; a.S = w.S;
  a.lTW = lTW; a.lTH = lTH; a.lNB = lNB; a.OY = OY; a.OX = OX;
  a.tilesX = OX / TW; a.tilesY = OY / TH;
  a.ntiles = a.tilesX * a.tilesY * ((B + NB - 1) / NB);
  a.TIW = (TW - 1) * w.S + (x_hi - x_lo) + 1; a.TIH = (TH - 1) * w.S + (y_hi - y_lo) + 1;
  a.y_lo = y_lo; a.x_lo = x_lo;
  a.PS = cin * 2 + (cin >= 32 ? 32 : 0);            // +32 B: odd multiple of 32 -> conflict-free transposed reads
  a.ldy = cout; a.YS = cout * 2 + (cout >= 32 ? 32 : 0);
  a.lycp = ilog2_exact(cout / 8);
  a.in_bytes = (NB * a.TIH * a.TIW * a.PS + 64 + 15) / 16 * 16;   // slack: 16-column transposed reads of narrow pixels
  a.dy_bytes = (BM * a.YS + 64 + 15) / 16 * 16;
  if (a.in_bytes + a.dy_bytes > 150 * 1024) return SV_E_UNSUPPORTED;
  a.Cin_real = w.Cin_real; a.N = w.N; a.ntaps = nt;
  memcpy(a.dy, w.dy, sizeof(a.dy));
  memcpy(a.dx, w.dx, sizeof(a.dx));
  switch (id) {
    case 0: return launch_wt<9, 2, 1, 4, 1, 8>(a, groups, st);
    case 1: return launch_wt<18, 1, 2, 1, 4, 4>(a, groups, st);
    case 2: return launch_wt<4, 2, 4, 1, 4, 4>(a, groups, st);
    case 3: return launch_wt<2, 2, 8, 1, 4, 2>(a, groups, st);
    case 4: return launch_wt<4, 1, 8, 1, 4, 2>(a, groups, st);
    case 5: return launch_wt<9, 1, 4, 2, 2, 2>(a, groups, st);
    case 6: return launch_wt<9, 1, 2, 4, 1, 8>(a, groups, st);
  }
  return SV_E_UNSUPPORTED;
}

int svk_wgrad_dispatch(const WgradArgs& w, int dtype, int cfg, hipStream_t st) {
  if (dtype == SV_BF16) {
    const int rc = svk_wgrad_tile(w, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  return svk_wgrad(w, dtype, cfg, st);
}
