// Shared LDS staging of an input spatial tile (+halo) for tile_conv.hip and wgrad_tile.hip.
//
//   stage_tile_plain      : tile pixels copied from the NHWC tensor, zero outside the image (SAME pad)
//   stage_tile_upsampled  : the conv's logical input is the 2x bilinear upsample (tf.image.resize,
//                           half-pixel centres, edge clamp: vae/model.py:163-167) of a LOW-RES tensor;
//                           the hi-res tile is produced on the fly and the hi-res tensor never exists.
//                           Hi-res rows {2i+1, 2i+2} both interpolate low-res rows i and i+1, so one
//                           thread turns 4 low-res 16-B pieces into a 2x2 block of hi-res pieces: the
//                           load count equals the plain path's, the bytes fetched are 4x fewer.
//                           Same blend order as upsample2x_fwd_kernel -> bitwise the same values.
#pragma once
#include "common.hip.h"

struct TileStageGeom {
  int B, IH, IW, lda;       // logical (hi-res when upsampled) input extent; lda = channels per pixel in memory
  int cl2;                  // log2(16-B chunks per pixel staged)
  int TIW, TIH, PS;         // LDS tile extent in pixels, bytes per pixel record
  int NB;                   // images per tile
  int plane_bytes;          // > 0: PLANAR tile -- the 16-B chunk c of a pixel lives in plane c>>1 (planes of 32-B
                            // pixel records, PS = 32): the same conflict-free ds_read_b128 pattern as a 32-B pixel for
                            // any channel count, without the +32 B padding of the linear layout
};
__device__ __forceinline__ int tile_piece_off(const TileStageGeom& s, int pixel, int c) {
  return s.plane_bytes ? (c >> 1) * s.plane_bytes + pixel * 32 + (c & 1) * 16 : pixel * s.PS + c * 16;
}

template <typename T, int NT = 256, bool CLAMP = false>   // NT = threads of the workgroup; CLAMP: replicate the edge instead of zero
__device__ __forceinline__ void stage_tile_plain(const T* __restrict__ Ab, const TileStageGeom& s, int b0, int iy_base,
                                                 int ix_base, char* sIn, int tid) {
  constexpr int EPP = ElemTraits<T>::EPP;
  const int cpp = 1 << s.cl2;
  // LPR lanes sweep one tile row (no integer division); 4 independent 16-B loads in flight per lane
  const int ppr = s.TIW * cpp;                      // pieces per tile row
  const int LPR = ppr > 160 ? 64 : 32, lLPR = ppr > 160 ? 6 : 5;
  const int srow = tid >> lLPR, slane = tid & (LPR - 1), rows_pp = NT >> lLPR;
  const int nrows = s.NB * s.TIH;
  for (int row = srow; row < nrows; row += rows_pp) {
    int bl = 0, iyl = row;
    while (iyl >= s.TIH) { iyl -= s.TIH; ++bl; }
    const int iy = CLAMP ? min(max(iy_base + iyl, 0), s.IH - 1) : iy_base + iyl, b = b0 + bl;
    const bool rok = b < s.B && (unsigned)iy < (unsigned)s.IH;
    const T* src = Ab + ((int64_t)(b * s.IH + iy) * s.IW) * s.lda;
    const int prow = row * s.TIW;                      // first tile pixel of this row
    for (int pc0 = slane; pc0 < ppr; pc0 += LPR * 4) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = pc0 + u * LPR;
        const int ixl = pc >> s.cl2, c = pc & (cpp - 1), ix = CLAMP ? min(max(ix_base + ixl, 0), s.IW - 1) : ix_base + ixl;
        v[u] = make_uint4(0, 0, 0, 0);
        if (pc < ppr && rok && (unsigned)ix < (unsigned)s.IW) v[u] = *(const uint4*)(src + (int64_t)ix * s.lda + c * EPP);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = pc0 + u * LPR;
        if (pc < ppr) *(uint4*)(sIn + tile_piece_off(s, prow + (pc >> s.cl2), pc & (cpp - 1))) = v[u];
      }
    }
  }
}

// The same tile by LDS-DMA (global_load_lds_dwordx4: no staging registers, every transfer of the tile in flight at once; the register form above keeps four
// 16-B loads per lane in flight and waits -- a 20 x 20 x 64-B tile is three dependent L2 round trips per workgroup, which two workgroups per CU do not hide).
// Tiles without padding bytes only: LINEAR (PS == 16 << cl2: piece L = pixel * cpp + c lives at byte 16 L) or PLANAR; one wave-instruction fills 64
// consecutive 16-B slots and each lane fetches the piece of its slot.  Lanes whose pixel is outside the image (zero padding) or past the batch store zeros instead
// (masked-off lanes of the DMA instruction write nothing).  `wave` must be wave-uniform.
__device__ __forceinline__ void stage_dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <typename T, int NW = 4, bool CLAMP = false>
__device__ __forceinline__ void stage_tile_plain_dma(const T* __restrict__ Ab, const TileStageGeom& s, int b0, int iy_base, int ix_base, char* sIn, int lane, int wave) {
  constexpr int EPP = ElemTraits<T>::EPP;
  const float inv_h = 1.0f / (float)s.TIH;
  auto piece = [&](int row, int ixl, int c, char* lds_wave_base, char* lds_lane) {       // row = image * TIH + tile row
    const int bl = (int)(((float)row + 0.5f) * inv_h), iyl = row - bl * s.TIH, b = b0 + bl;
    int iy = iy_base + iyl, ix = ix_base + ixl;
    bool ok = b < s.B;
    if (CLAMP) { iy = min(max(iy, 0), s.IH - 1); ix = min(max(ix, 0), s.IW - 1); }
    else ok = ok && (unsigned)iy < (unsigned)s.IH && (unsigned)ix < (unsigned)s.IW;
    if (ok) stage_dma16(Ab + ((int64_t)(b * s.IH + iy) * s.IW + ix) * s.lda + c * EPP, lds_wave_base);
    else *(uint4*)lds_lane = make_uint4(0, 0, 0, 0);
  };
  if (s.plane_bytes) {
    // PLANAR tiles: plane k holds chunks 2k, 2k + 1 of every pixel as 32-B records -- slot = pixel * 2 + (c & 1) is linear inside a plane
    const int slots = s.NB * s.TIH * s.TIW * 2, ipp = (slots + 63) >> 6, nplanes = 1 << (s.cl2 - 1);
    const float inv_w = 1.0f / (float)s.TIW;
    int plane = 0, k = wave;
    while (k >= ipp) { k -= ipp; ++plane; }
    while (plane < nplanes) {
      const int slot = k * 64 + lane;
      if (slot < slots) {
        const int p = slot >> 1, row = (int)(((float)p + 0.5f) * inv_w);
        piece(row, p - row * s.TIW, plane * 2 + (slot & 1), sIn + plane * s.plane_bytes + k * 1024, sIn + plane * s.plane_bytes + slot * 16);
      }
      k += NW;
      while (k >= ipp) { k -= ipp; ++plane; }
    }
    return;
  }
  const int cpp = 1 << s.cl2, ppr = s.TIW << s.cl2, total = s.NB * s.TIH * ppr;
  const float inv_row = 1.0f / (float)ppr;
  for (int base = wave * 64; base < total; base += NW * 64) {
    const int L = base + lane;
    if (L >= total) continue;
    const int row = (int)(((float)L + 0.5f) * inv_row), r = L - row * ppr;            // (image, tile row); (tile column, chunk)
    piece(row, r >> s.cl2, r & (cpp - 1), sIn + base * 16, sIn + L * 16);
  }
}

// Space-to-depth view of the padded RGB tensor (conv_geom.h: svg_s2d3): Ab = [B, 2 IH, 2 IW, 8] fp32 (channels 0..2 real), the tile pixel (i, j) of the
// [B, IH, IW, 16] view holds channel (py*2+px)*3 + c = pixel (2i+py, 2j+px) channel c; zero outside the image (SAME padding) and in channels 12..15.
// One item = one source pixel: a 16-B load (its first four floats), three 4-B LDS stores.
template <int NT = 256>
__device__ __forceinline__ void stage_tile_s2d3(const float* __restrict__ Ab, const TileStageGeom& s, int b0, int iy_base, int ix_base, char* sIn, int tid) {
  const int total = s.NB * s.TIH * s.TIW * 4;
  const float inv_row = 1.0f / (float)(s.TIW * 4), inv_h = 1.0f / (float)s.TIH;
  constexpr int U = 4;                                   // loads in flight per lane (one per pass left every item's L2 round trip in line: five passes for an 18 x 18 tile)
  for (int it0 = tid; it0 < total; it0 += NT * U) {
    float4 v[U];
    int pix[U], par[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = it0 + u * NT;
      const int row = (int)(((float)it + 0.5f) * inv_row), r = it - row * (s.TIW * 4);      // (image, tile row); (tile column, parity)
      const int bl = (int)(((float)row + 0.5f) * inv_h), iyl = row - bl * s.TIH;
      const int ixl = r >> 2, p = r & 3, iy = iy_base + iyl, ix = ix_base + ixl, b = b0 + bl;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < total && b < s.B && (unsigned)iy < (unsigned)s.IH && (unsigned)ix < (unsigned)s.IW)
        v[u] = *(const float4*)(Ab + (((int64_t)b * (2 * s.IH) + 2 * iy + (p >> 1)) * (2 * s.IW) + 2 * ix + (p & 1)) * 8);
      pix[u] = it < total ? row * s.TIW + ixl : -1;
      par[u] = p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (pix[u] < 0) continue;
      const float f[3] = {v[u].x, v[u].y, v[u].z};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int ch = par[u] * 3 + c;
        *(float*)(sIn + tile_piece_off(s, pix[u], ch >> 2) + (ch & 3) * 4) = f[c];
      }
      if (par[u] == 0) *(float4*)(sIn + tile_piece_off(s, pix[u], 3)) = make_float4(0.f, 0.f, 0.f, 0.f);     // channels 12..15
    }
  }
}

// the 2x2 hi-res block {2i+1, 2i+2} x {2j+1, 2j+2} from the low-res 2x2 neighbourhood: horizontal
// interpolation first (shared by the two rows), packed fp32 math (56 v_pk ops per 16-B piece quad
// instead of ~200 scalar ones: the blend, not the MFMAs, was the busiest user of the issue slots)
template <typename T>
__device__ __forceinline__ void blend2x2(const uint4& a00, const uint4& a01, const uint4& a10, const uint4& a11, uint4 (&out)[2][2]) {
  constexpr int NP = Piece<T>::NP;
  f32x2 v00[NP], v01[NP], v10[NP], v11[NP];
  Piece<T>::unpack(a00, v00); Piece<T>::unpack(a01, v01); Piece<T>::unpack(a10, v10); Piece<T>::unpack(a11, v11);
#pragma unroll
  for (int dxb = 0; dxb < 2; ++dxb) {
    const float fx = dxb ? 0.75f : 0.25f;               // weight of the second (x1) column: odd X .25, even X .75
    f32x2 top[NP], bot[NP], r0[NP], r1[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      top[e] = lerp2(v00[e], v01[e], fx);
      bot[e] = lerp2(v10[e], v11[e], fx);
      r0[e] = lerp2(top[e], bot[e], 0.25f);             // odd Y
      r1[e] = lerp2(top[e], bot[e], 0.75f);             // even Y
    }
    out[0][dxb] = Piece<T>::pack(r0);
    out[1][dxb] = Piece<T>::pack(r1);
  }
}

// Ab: LOW-RES tensor [B, IH/2, IW/2, lda]; geometry (IH, IW, iy_base, ix_base, tile) in HI-RES pixels.
//
// Column-walking form: a thread owns one (image, block column j, 16-B channel piece) and walks a run of block rows,
// carrying the HORIZONTALLY interpolated low-res row from one block row to the next -- half the loads and horizontal
// lerps of the per-block form below, and the per-block index arithmetic, clamps and column tests happen once per run.
// The arithmetic per value is that of blend2x2 (horizontal lerp first, then the vertical one): bitwise the same tile.
template <typename T, int NT = 256>
__device__ __forceinline__ void stage_tile_upsampled_cols(const T* __restrict__ Ab, const TileStageGeom& s, int b0, int iy_base,
                                                          int ix_base, char* sIn, int tid) {
  constexpr int EPP = ElemTraits<T>::EPP, NP = Piece<T>::NP;
  const int cpp = 1 << s.cl2;
  const int LH = s.IH >> 1, LW = s.IW >> 1;
  const int i_lo = (iy_base - 1) >> 1, nbi = ((iy_base + s.TIH - 2) >> 1) - i_lo + 1;
  const int j_lo = (ix_base - 1) >> 1, nbj = ((ix_base + s.TIW - 2) >> 1) - j_lo + 1;
  const int ncols = s.NB * nbj * cpp;
  int nrg = NT / ncols;                                   // row groups: as many as fill the workgroup
  nrg = nrg < 1 ? 1 : nrg > nbi ? nbi : nrg;
  const int rpg = (nbi + nrg - 1) / nrg;
  const int items = ncols * nrg;
  const float inv_cols = 1.0f / (float)ncols, inv_nbj = 1.0f / (float)nbj;
  for (int it = tid; it < items; it += NT) {
    const int rg = (int)(((float)it + 0.5f) * inv_cols), r = it - rg * ncols;
    const int c = r & (cpp - 1), r2 = r >> s.cl2;
    const int bl = (int)(((float)r2 + 0.5f) * inv_nbj), bj = r2 - bl * nbj;
    const int j = j_lo + bj, b = b0 + bl;
    const bool bok = b < s.B;
    const int x0 = min(max(j, 0), LW - 1), x1 = min(max(j + 1, 0), LW - 1);
    const T* img = Ab + (int64_t)b * LH * LW * s.lda + c * EPP;
    int txs[2];
    bool colok[2], colin[2];
#pragma unroll
    for (int dxb = 0; dxb < 2; ++dxb) {
      const int X = 2 * j + 1 + dxb;
      txs[dxb] = X - ix_base;
      colok[dxb] = (unsigned)txs[dxb] < (unsigned)s.TIW;
      colin[dxb] = bok && (unsigned)X < (unsigned)s.IW;    // SAME padding lives in hi-res space
    }
    const int i_beg = i_lo + rg * rpg, i_end = min(i_beg + rpg, i_lo + nbi);
    auto hrow = [&](int i, f32x2 (&h)[2][NP]) {            // low-res row clamp(i), interpolated to the two hi-res columns
      const int y = min(max(i, 0), LH - 1);
      uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
      if (bok) {
        a0 = *(const uint4*)(img + ((int64_t)y * LW + x0) * s.lda);
        a1 = *(const uint4*)(img + ((int64_t)y * LW + x1) * s.lda);
      }
      f32x2 v0[NP], v1[NP];
      Piece<T>::unpack(a0, v0); Piece<T>::unpack(a1, v1);
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        h[0][e] = lerp2(v0[e], v1[e], 0.25f);              // odd X
        h[1][e] = lerp2(v0[e], v1[e], 0.75f);              // even X
      }
    };
    f32x2 hp[2][NP], hc[2][NP];
    hrow(i_beg, hp);
    for (int i = i_beg; i < i_end; ++i) {
      hrow(i + 1, hc);
#pragma unroll
      for (int dyb = 0; dyb < 2; ++dyb) {
        const int Y = 2 * i + 1 + dyb, ty = Y - iy_base;
        if ((unsigned)ty >= (unsigned)s.TIH) continue;
        const bool rowin = (unsigned)Y < (unsigned)s.IH;
        const float fy = dyb ? 0.75f : 0.25f;              // odd Y .25, even Y .75
#pragma unroll
        for (int dxb = 0; dxb < 2; ++dxb) {
          if (!colok[dxb]) continue;
          f32x2 o[NP];
#pragma unroll
          for (int e = 0; e < NP; ++e) o[e] = lerp2(hp[dxb][e], hc[dxb][e], fy);
          const uint4 v = (rowin && colin[dxb]) ? Piece<T>::pack(o) : make_uint4(0, 0, 0, 0);
          *(uint4*)(sIn + tile_piece_off(s, (bl * s.TIH + ty) * s.TIW + txs[dxb], c)) = v;
        }
      }
#pragma unroll
      for (int dxb = 0; dxb < 2; ++dxb)
#pragma unroll
        for (int e = 0; e < NP; ++e) hp[dxb][e] = hc[dxb][e];
    }
  }
}

template <typename T, int NT = 256>
__device__ __forceinline__ void stage_tile_upsampled_blocks(const T* __restrict__ Ab, const TileStageGeom& s, int b0, int iy_base,
                                                     int ix_base, char* sIn, int tid) {
  constexpr int EPP = ElemTraits<T>::EPP;
  const int cpp = 1 << s.cl2;
  const int LH = s.IH >> 1, LW = s.IW >> 1;
  // block row i covers hi-res rows 2i+1, 2i+2 (i = (Y-1)>>1): low-res rows i, i+1 (clamped)
  const int i_lo = (iy_base - 1) >> 1, nbi = ((iy_base + s.TIH - 2) >> 1) - i_lo + 1;
  const int j_lo = (ix_base - 1) >> 1, nbj = ((ix_base + s.TIW - 2) >> 1) - j_lo + 1;
  const int per_img = nbi * nbj * cpp, total = s.NB * per_img;
  // q / per_img and r2 / nbj by reciprocal multiplication: exact for these small operands
  // ((n + 0.5) / d is never within 2^-20 of an integer), 3 VALU ops instead of a ~25-op integer division
  const float inv_img = 1.0f / (float)per_img, inv_nbj = 1.0f / (float)nbj;
  for (int q = tid; q < total; q += NT) {
    const int bl = (int)(((float)q + 0.5f) * inv_img), r1 = q - bl * per_img;
    const int c = r1 & (cpp - 1), r2 = r1 >> s.cl2;
    const int bi = (int)(((float)r2 + 0.5f) * inv_nbj), bj = r2 - bi * nbj;
    const int i = i_lo + bi, j = j_lo + bj, b = b0 + bl;
    const int y0 = min(max(i, 0), LH - 1), y1 = min(max(i + 1, 0), LH - 1);
    const int x0 = min(max(j, 0), LW - 1), x1 = min(max(j + 1, 0), LW - 1);
    uint4 a00 = make_uint4(0, 0, 0, 0), a01 = a00, a10 = a00, a11 = a00;
    if (b < s.B) {
      const T* img = Ab + (int64_t)b * LH * LW * s.lda + c * EPP;
      a00 = *(const uint4*)(img + ((int64_t)y0 * LW + x0) * s.lda);
      a01 = *(const uint4*)(img + ((int64_t)y0 * LW + x1) * s.lda);
      a10 = *(const uint4*)(img + ((int64_t)y1 * LW + x0) * s.lda);
      a11 = *(const uint4*)(img + ((int64_t)y1 * LW + x1) * s.lda);
    }
    uint4 blk[2][2];
    blend2x2<T>(a00, a01, a10, a11, blk);
#pragma unroll
    for (int dyb = 0; dyb < 2; ++dyb) {
      const int Y = 2 * i + 1 + dyb, ty = Y - iy_base;
      if ((unsigned)ty >= (unsigned)s.TIH) continue;
#pragma unroll
      for (int dxb = 0; dxb < 2; ++dxb) {
        const int X = 2 * j + 1 + dxb, tx = X - ix_base;
        if ((unsigned)tx >= (unsigned)s.TIW) continue;
        // SAME padding lives in hi-res space
        const bool in = b < s.B && (unsigned)Y < (unsigned)s.IH && (unsigned)X < (unsigned)s.IW;
        *(uint4*)(sIn + tile_piece_off(s, (bl * s.TIH + ty) * s.TIW + tx, c)) = in ? blk[dyb][dxb] : make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// the staging every kernel calls: the per-block form.  -DSV_STAGE_COLS: the column-walking form (measured A/B, round 2:
// bitwise the same tiles, 2.5x fewer VALU instructions per tile, and 2.5 % SLOWER on the whole step (2.433 vs 2.374 ms) --
// its threads carry a dependent chain of row loads where the per-block form keeps four independent loads per thread in
// flight: the staging is bound by load latency, not by instruction issue)
template <typename T, int NT = 256>
__device__ __forceinline__ void stage_tile_upsampled(const T* __restrict__ Ab, const TileStageGeom& s, int b0, int iy_base,
                                                     int ix_base, char* sIn, int tid) {
#ifdef SV_STAGE_COLS
  stage_tile_upsampled_cols<T, NT>(Ab, s, b0, iy_base, ix_base, sIn, tid);
#else
  stage_tile_upsampled_blocks<T, NT>(Ab, s, b0, iy_base, ix_base, sIn, tid);
#endif
}
