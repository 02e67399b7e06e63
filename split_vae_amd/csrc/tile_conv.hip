// Direct NHWC convolution on the gfx950 matrix cores with the input tile resident in LDS.
//
// The im2col "tap GEMM" (tap_gemm.hip) re-gathers every input pixel once per tap (16-36x) through
// L2.  Here a workgroup stages its input SPATIAL tile (+ halo, SAME padding zero-filled) into LDS
// ONCE with coalesced 16-B loads and every tap's MFMA A-fragment is a shifted window of that tile:
//     A-frag(lane) = lds[ lane_pixel_base + piece_offset[p] ],   p = (tap, 16-B channel chunk)
// so the only per-K-step global traffic is the (L2-resident) weight tile.  Rows of the implicit
// GEMM are the pixels of a TH x TW patch of NB images (TW = min(W,16) so a 16-row MFMA fragment is
// 16 neighbouring pixels); input stride 1 (decoder convs, all dgrads) or 2 (encoder forwards);
// outputs may be scattered with stride/offset (parity classes of a stride-2 dgrad).
//
// Weight pipeline: K advances in 128-byte (8-piece) steps, weight tile of step ks+1 prefetched
// global -> registers during step ks and written to the other LDS buffer.  (A 3-slot ring with a
// two-step prefetch was measured SLOWER: its extra 4-8 KB of LDS drops the big-tile layers from
// two resident workgroups per CU to one, and that overlap is worth more than the latency cover.)
#include "common.hip.h"
#include "kernels.h"
#include "tile_stage.hip.h"
#include "dlogistic.hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef SV_TC_PPS16
#define SV_TC_PPS16 16    // K-step pieces of the 16-column kernel (build-time A/B knob)
#endif
#ifndef SV_TC_PPS32
#define SV_TC_PPS32 8     // K-step pieces of the 32-column kernel (build-time A/B knob)
#endif
#ifndef SV_TC_PPS64
#define SV_TC_PPS64 8     // ... of the 64-column kernel
#endif
// with row-window reuse (YR = KH) the 16-column kernel steps one whole filter column (KH taps x 4 pieces), the wider
// ones two taps (their weight tiles are BN x PPS x 16 B x 2 buffers of LDS)
static __host__ __device__ constexpr int tile_pps(int BN, int YR = 0) {
  return YR ? (BN == 16 || YR == 5 ? YR * 4 : 8) : BN == 16 ? SV_TC_PPS16 : BN == 32 ? SV_TC_PPS32 : BN == 64 ? SV_TC_PPS64 : 8;   // (KH = 5: a step = one filter column, 20 pieces)
}

// position of weight piece q of row n inside its LDS row: XOR swizzle within each 8-piece group (a trailing group of 4 pieces,
// PPS = 20, swizzles within 4) -- the rows of a fragment then start on different banks
template <int PPS>
__device__ __forceinline__ int wpiece(int q, int n) {
  if constexpr (PPS % 8 == 0) return q ^ (n & 7);
  else return (q & ~7) == (PPS & ~7) ? (q & ~3) | ((q & 3) ^ (n & 3)) : q ^ (n & 7);
}

template <typename T> struct MmaOpT;
template <> struct MmaOpT<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaOpT<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    const float4 af = __builtin_bit_cast(float4, a), bf = __builtin_bit_cast(float4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, c, 0, 0, 0);
  }
};

// NW waves per workgroup: 4, or 8 (same tile, half the row fragments per wave) where the LDS tile
// limits a CU to two workgroups -- the kernel is latency-bound and wants waves
// YR > 0 (= filter height KH; stride-1 layers, taps x-major / y-minor, 32 bf16 channels per tap and phase, 16-pixel tile
// rows): ROW-WINDOW REUSE.  The A fragment of tap (ky, kx) for output row y is the A fragment of tap (ky+1, kx) for row
// y-1, and a wave's MF fragments are MF consecutive tile rows: per filter column a wave reads the MF+KH-1 input-row
// fragments it touches ONCE into registers and issues all MF*KH MFMAs from them -- 9 LDS reads instead of 24 for
// KH = 6 (the narrow-N layers are LDS-read-bound: every A fragment used to feed NF MFMAs only), back to back, so their
// latency overlaps the MFMAs of the rows that arrived first.
// waves per SIMD the row-window kernels are compiled for (left alone, hipcc spreads the window over AGPR copies and
// drops a wave: 92 + 120 registers for <64,4,4,YR=4>)
static constexpr int tile_min_waves(int BN, int NW, int YR, bool WR = false) { return (YR || WR) && NW == 4 ? (BN == 64 ? 3 : 4) : 1; }
// FX: the epilogue's per-class border-term path (TileConvArgs::fix_nc) is compiled in: always at fp32; at bf16 only in the instantiations the polyphase input
// gradient uses (the extra address registers cost the bf16 head's fused-loss epilogue 24 us per launch when every bf16 kernel carried them)
template <typename T, int BN, int MF, int NW, int YR, bool WR, bool FX = false>
__global__ __launch_bounds__(64 * NW, tile_min_waves(BN, NW, YR, WR)) void tile_conv_kernel(const TileConvMulti mg) {
  // blockIdx.z picks one of up to 8 problems of identical geometry (the x / x_hat twin networks and
  // the four parity classes of a stride-2 dgrad) so that they share one launch and one wave of
  // workgroups instead of paying the ~10 us fixed latency of a launch each
  const TileConvArgs& g = mg.a[blockIdx.z];
  if (SV_DBG(g.dbg) & 32) return;                     // ablation build only: the cost of dispatching the grid
  constexpr int NT = 64 * NW, WM = 16 * MF, BM = NW * WM, NF = BN / 16;
  // 16-B pieces of K per step.  128 B per weight row normally; the 16-column kernel takes 512-B steps:
  // its steps are otherwise 8 MFMAs per wave between two barriers (21 -> 6 steps for the packed d5)
  constexpr int PPS = tile_pps(BN, YR), RB = PPS * 16;   // RB = bytes per weight row per step
  constexpr int RPP = NT / PPS;                      // weight rows loaded per pass (PPS threads per row)
  constexpr int BRN = (BN + RPP - 1) / RPP;          // weight pieces per thread per step
  constexpr int EPP = ElemTraits<T>::EPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sB = smem;                                   // [2][BN][128 B], XOR-swizzled
  int* sOff = (int*)(smem + g.wslots * BN * RB);     // piece offsets, padded to a multiple of PPS
  char* sIn = smem + g.wslots * BN * RB + g.off_bytes;   // input tile [NB][TIH][TIW] pixels of PS bytes

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- which tile
  // workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give XCD k a CONTIGUOUS range of
  // tiles so that neighbouring tiles (shared halos, same image) meet in one L2 (g.xcd_chunk = ceil(ntiles/8), 0 = off)
  int t = blockIdx.x;
  if (g.xcd_chunk) {
    const int remapped = (t & 7) * g.xcd_chunk + (t >> 3);
    t = remapped < g.ntiles && (g.ntiles & 7) == 0 ? remapped : t;
  }
  const int tx0 = (t % g.tilesX) << g.lTW; t /= g.tilesX;
  const int ty0 = (t % g.tilesY) << g.lTH; t /= g.tilesY;
  const int b0 = t << g.lNB;
  const int n0 = blockIdx.y * BN;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH, NB = 1 << g.lNB;
  const int cpp = 1 << g.cl2;                         // 16-B chunks per pixel

  // ---- weight-tile pipeline: global -> one of two register sets -> LDS ring
  // (kernarg fields used in the K loop are copied to locals: through the reference hipcc re-issued
  // their scalar loads, with an lgkmcnt(0) wait each, in every K step)
  const T* __restrict__ Wb = (const T*)g.Wt;
  // CHANNEL PHASES (g.nph > 1, planar tiles): the tile is staged nph times with 1/nph of the channels each, the
  // K loop of a phase runs over (all taps) x (its channels), and the accumulators carry over.  The LDS tile shrinks
  // nph-fold, so more workgroups fit a CU (the kernel is latency-bound: occupancy is what it lacks).
  const int nph = g.nph, lcH = g.cl2 - g.lnph;        // log2(16-B chunks per pixel per phase)
  const int gP = g.P >> g.lnph, gKtot = g.Ktot, gdbg = SV_DBG(g.dbg);   // pieces of K per phase
  const int pp = tid % PPS, r0 = tid / PPS;
  int ph = 0;                                         // current phase (read by load_b)
  uint4 rbA[BRN];
  auto load_b = [&](int ks, uint4 (&rb)[BRN]) {
    const int pl = ks * PPS + pp;                     // piece within the phase: (tap, local chunk)
    const bool pv = pl < gP;
    const int p = ((pl >> lcH) << g.cl2) + (ph << lcH) + (pl & ((1 << lcH) - 1));   // piece in the weight row
#pragma unroll
    for (int i = 0; i < BRN; ++i) {
      const int n = r0 + RPP * i;
      rb[i] = (pv && n < BN && r0 < RPP) ? *(const uint4*)(Wb + (int64_t)(n0 + n) * gKtot + (int64_t)p * EPP) : make_uint4(0, 0, 0, 0);
    }
  };
  auto write_b = [&](int slot, const uint4 (&rb)[BRN]) {
#pragma unroll
    for (int i = 0; i < BRN; ++i) {
      const int n = r0 + RPP * i;
      if (n < BN && r0 < RPP) *(uint4*)(sB + slot * (BN * RB) + n * RB + (wpiece<PPS>(pp, n) << 4)) = rb[i];   // swizzled (wpiece)
    }
  };
  const int nk = (gdbg & 2) ? 0 : (gP + PPS - 1) / PPS;
  // WHOLE K RESIDENT (g.wslots > 2; small-K layers: the 8-channel d5 input gradient and e1): all K steps' weight tiles
  // (<= 24 KB) go to LDS once, next to the input tile, and the K loop runs without global loads or barriers.  Streamed,
  // each of its 5 steps is 16 MFMAs per wave behind a dependent L2 round trip and a barrier.
  constexpr int WRMAX = 6;
  constexpr bool wres = WR && YR == 0;
  uint4 rbw[wres ? WRMAX - 1 : 1][BRN];
  load_b(0, rbA);                                     // in flight while the input tile is staged
  if constexpr (wres) {
#pragma unroll
    for (int ks = 1; ks < WRMAX; ++ks) load_b(ks, rbw[ks - 1]);   // zero-filled past the last step
  }

  // ---- piece-offset table
  const int nkp = (gP + PPS - 1) / PPS * PPS;
  for (int p = tid; p < nkp; p += NT) {
    int off = 0;
    if (p < gP) {
      const int tap = p >> lcH, c = p & ((1 << lcH) - 1);
      const int tp = ((int)g.dy[tap] - g.y_lo) * g.TIW + ((int)g.dx[tap] - g.x_lo);   // tap offset in tile pixels
      off = g.plane_bytes ? tp * 32 + (c >> 1) * g.plane_bytes + (c & 1) * 16 : tp * g.PS + c * 16;
    }
    sOff[p] = off;
  }
  auto stage = [&](int phase) {                       // input tile of one phase (zero-filled outside the image)
    if (SV_DBG(g.dbg) & 1) return;
    const TileStageGeom sg = {g.B, g.IH, g.IW, g.lda, lcH, g.TIW, g.TIH, g.PS, NB, g.plane_bytes};
    const int iy_base = ty0 * g.S + g.y_lo, ix_base = tx0 * g.SX + g.x_lo;
    const T* Ap = (const T*)g.A + ((phase << lcH) * EPP);
    if constexpr (sizeof(T) == 4) {
      if (g.s2d3) { stage_tile_s2d3<NT>((const float*)g.A, sg, b0, iy_base, ix_base, sIn, tid); return; }     // the padded RGB tensor through its space-to-depth view
    }
    if (g.ups) stage_tile_upsampled<T, NT>(Ap, sg, b0, iy_base, ix_base, sIn, tid);
    else if (sizeof(T) == 4 && g.dma) {               // LDS-DMA: every transfer of the tile in flight at once, no staging registers
      // (fp32 instantiations only -- the constant folds the branch away at bf16: compiled into the bf16 kernels it cost the head's fused-loss forward 0.125 -> 0.150 ms)
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      if (g.clampin) stage_tile_plain_dma<T, NW, true>(Ap, sg, b0, iy_base, ix_base, sIn, lane, wv);
      else stage_tile_plain_dma<T, NW, false>(Ap, sg, b0, iy_base, ix_base, sIn, lane, wv);
    }
    else if (g.clampin) stage_tile_plain<T, NT, true>(Ap, sg, b0, iy_base, ix_base, sIn, tid);
    else stage_tile_plain<T, NT>(Ap, sg, b0, iy_base, ix_base, sIn, tid);
  };
  stage(0);
  // ---- per-lane pixel bases of this wave's MF row fragments
  const int lr = lane & 15, lg = lane >> 4;
  int lbase[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int r = wave * WM + i * 16 + lr;
    const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
    lbase[i] = ((bl * g.TIH + ty * g.S) * g.TIW + tx * g.SX) * g.PS;
  }

  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int ks, int slot) {
    const char* cB = sB + slot * (BN * RB);
#pragma unroll
    for (int kk = 0; kk < PPS / 4; ++kk) {
      const int off = sOff[ks * PPS + kk * 4 + lg];
      uint4 af[MF], bfr[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) af[i] = *(const uint4*)(sIn + lbase[i] + off);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int n = j * 16 + lr;
        bfr[j] = *(const uint4*)(cB + n * RB + (wpiece<PPS>(kk * 4 + lg, n) << 4));
      }
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) MmaOpT<T>::run(bfr[j], af[i], acc[i][j]);   // D = W x pixels: a lane ends up with 4 CHANNELS of one pixel
    }
  };
  // row-window variant: one filter column (YR taps = YR*4 pieces) per group, SPG K steps per group
  constexpr int SPG = YR ? YR * 4 / PPS : 1, TPS = PPS / 4;       // steps per group, taps per step
  const int rowpitch = g.TIW * g.PS;
  for (ph = 0; ph < nph; ++ph) {
    if (ph) {                                           // the last barrier of the previous K loop freed tile and weights
      load_b(0, rbA);
      stage(ph);
    }
    write_b(0, rbA);
    if constexpr (wres) {
#pragma unroll
      for (int ks = 1; ks < WRMAX; ++ks)
        if (ks < nk) write_b(ks, rbw[ks - 1]);
    }
    __syncthreads();                                    // input tile, offsets and weight tile 0 (or all of them) visible
    if constexpr (YR > 0) {
      for (int gq = 0; gq * SPG < nk; ++gq) {
        uint4 aw[MF + YR - 1];                          // input-row fragments ty .. ty+MF+YR-2 of this filter column
        const char* wp = sIn + lbase[0] + sOff[gq * (YR * 4) + lg];
#pragma unroll
        for (int iy = 0; iy < MF + YR - 1; ++iy) aw[iy] = *(const uint4*)(wp + iy * rowpitch);
#pragma unroll
        for (int sg = 0; sg < SPG; ++sg) {
          const int ks = gq * SPG + sg;
          const bool more = ks + 1 < nk;
          if (more) load_b(ks + 1, rbA);
          const char* cB = sB + (ks & 1) * (BN * RB);
#pragma unroll
          for (int kk = 0; kk < TPS; ++kk) {
            uint4 bfr[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
              const int n = j * 16 + lr;
              bfr[j] = *(const uint4*)(cB + n * RB + (wpiece<PPS>(kk * 4 + lg, n) << 4));
            }
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
              for (int j = 0; j < NF; ++j) MmaOpT<T>::run(bfr[j], aw[i + sg * TPS + kk], acc[i][j]);
          }
          if (more) write_b((ks & 1) ^ 1, rbA);
          __syncthreads();
        }
      }
    } else if constexpr (wres) {
      for (int ks = 0; ks < nk; ++ks) compute(ks, ks);
      __syncthreads();                                  // the epilogue reuses the LDS
    } else
    for (int ks = 0; ks < nk; ++ks) {
      const bool more = ks + 1 < nk && !(gdbg & 8);      // dbg 8: ablate the weight streaming (stale LDS weights)
      if (more) load_b(ks + 1, rbA);
      compute(ks, ks & 1);
      if (more) write_b((ks & 1) ^ 1, rbA);
      if (!(gdbg & 16)) __syncthreads();                 // dbg 16: ablate the per-step barrier
    }
    if (nk == 0) __syncthreads();
  }

  // ---- epilogue.  The operands are swapped (D rows = output channels, cols = pixels), so a lane
  // holds 4 consecutive channels (lane>>4)*4.. of pixel lane&15 per fragment: bias (one 16-B load per
  // column fragment) + activation, then ONE 8-byte (bf16) / 16-byte (fp32) LDS store per fragment into
  // a pixel-major tile that is written out with 16-byte row-contiguous stores (8-byte for the
  // 6-channel fp32 head), ReLU mask applied there.  (The first version stored 2-byte scalars and
  // re-loaded the bias per element behind a branch: 32 dependent global loads per tile, a 35 us floor.)
  const int oesz = g.out_f32 ? 4 : (int)sizeof(T);
  // channels stored by this column tile: the real ones and -- when their bytes are not a multiple of the 8-byte store piece (a
  // 3-channel input gradient: SPAIR's glimpse encoder) -- the zero pad channels up to the tensor's pitch (weight rows >= N are
  // zero, the bias is skipped); the plan rejects layers whose pitch leaves no such room
  const int Nst = (!g.d2s && !g.cls_n && ((g.N * oesz) & 7)) ? min(g.ldo, (g.N + 7) & ~7) : g.N;
  const int ncols = min(BN, Nst - n0);
  // x-packed conv (fp32 head): the 16 columns (px, co<8) become 2*C contiguous floats of output pixels 2*ox, 2*ox+1
  // (polyphase form, d2s_y: 32 columns (py, px, co<8) become two segments of 2*C floats: output rows 2*oy, 2*oy + 1)
  const int rowb = g.d2s ? (g.d2s_y ? 4 : 2) * g.d2s * 4 : ncols * oesz;   // output bytes per tile row
  const int srow = ((rowb + 15) & ~15) + 16;            // LDS row pitch (padded)
  char* sC = smem;                                      // reuse: every LDS read finished at the loop's last barrier
  float bv[NF][4];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int nl = j * 16 + lg * 4 + e;
      const int bi = g.d2s ? (nl & 7) : n0 + nl;
      bv[j][e] = (g.bias && nl < ncols && n0 + nl < g.N && (!g.d2s || bi < g.d2s)) ? g.bias[bi] : 0.f;
    }
  const bool relu = g.act == SV_ACT_RELU;
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int r = wave * WM + i * 16 + lr;
    // PER-CLASS POLYPHASE (conv_geom.h: svg_polyc): the out-of-image taps of the hi-res border rows / columns (poly_fix.hip: polyc_fix_kernel wrote them)
    // are added before the activation; only the lanes of border pixels load anything
    // (fp32 instantiations only: the extra address registers cost the bf16 head's fused-loss epilogue 24 us per launch -- 0.127 -> 0.150 ms -- when compiled in)
    constexpr bool FIXC = FX || sizeof(T) == 4;
    const float *frp = nullptr, *fcp = nullptr;
    if (FIXC && g.fix_nc) {
      const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
      const int b = b0 + bl, R = (ty0 + ty) * g.OS + g.ooy, Cc = (tx0 + tx) * g.OS + g.oox, nbot = g.fix_nc - g.fix_pad;
      const int rc = R < g.fix_pad ? R : R >= g.OHF - nbot ? g.fix_pad + R - (g.OHF - nbot) : -1;
      const int cc = Cc < g.fix_pad ? Cc : Cc >= g.OWF - nbot ? g.fix_pad + Cc - (g.OWF - nbot) : -1;
      if (b < g.B && rc >= 0) frp = g.fix + (((int64_t)b * g.fix_nc + rc) * g.OWF + Cc) * g.N + n0;
      if (b < g.B && cc >= 0) fcp = g.fix2 + (((int64_t)b * g.OHF + R) * g.fix_nc + cc) * g.N + n0;
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int nl = j * 16 + lg * 4;
      if (nl >= ncols) continue;
      float v[4];
      float fv[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (FIXC) {
        float4 fr = make_float4(0.f, 0.f, 0.f, 0.f), fc = fr;
        if (frp) fr = *(const float4*)(frp + nl);
        if (fcp) fc = *(const float4*)(fcp + nl);
        fv[0] = fr.x + fc.x; fv[1] = fr.y + fc.y; fv[2] = fr.z + fc.z; fv[3] = fr.w + fc.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[i][j][e] + bv[j][e];
        if (FIXC && g.fix_nc) v[e] += fv[e];
        if (relu) v[e] = fmaxf(v[e], 0.f);
      }
      if (g.d2s) {                                      // n = px*8 + co -> column px*C + co (co < C)
        const int px = nl >> 3, c0 = nl & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e < g.d2s) *(float*)(sC + r * srow + (px * g.d2s + c0 + e) * 4) = v[e];
      } else if (g.out_f32) *(float4*)(sC + r * srow + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
      else if constexpr (sizeof(T) == 2) {
        T pk[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
        *(uint2*)(sC + r * srow + nl * 2) = *(uint2*)pk;
      } else *(float4*)(sC + r * srow + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();
  if (SV_DBG(g.dbg) & 4) return;
  const int psz = (rowb & 15) ? 8 : 16;                 // piece size; rowb is a multiple of 8
  const int ppr_o = rowb / psz;
  if (g.d2s_y) {
    // POLYPHASE head (svg_poly): a tile row = the 2x2 output pixels of one low-res pixel: pieces 0..pps-1 go to output row
    // 2*oy, the rest to 2*oy + 1.  Border rows / columns add their out-of-image terms, which poly_fix.hip wrote in this
    // tensor's own layout (one aligned 16-B piece each); the loads of four pieces are issued before any is used.
    const int C = g.d2s, pps = ppr_o >> 1, total = BM * ppr_o;
    {
      if (g.nll_part) {
        // FUSED LOSS: one thread per hi-res pixel (4 per tile row): its 6 outputs + border terms -> out6, the NLL of its three
        // colour channels against the target image (vae/trainer.py:21-38, :127) and the bf16 gradient record; the tile's
        // NLL sum goes to nll_part[image][tile] (summed per image in a fixed order by svk_nll_rowsum: deterministic).
        const float* __restrict__ img = g.nll_img;
        float* __restrict__ outp = (float*)g.out;
        T* __restrict__ gp = (T*)g.nll_grad;                 // the gradient record in the step's dtype: [pixel][dm0 dm1 dm2 dl0 dl1 dl2 0 0]
        float nacc = 0.f;
#pragma unroll 2
        for (int hp = tid; hp < BM * 4; hp += NT) {
          const int r = hp >> 2, sub = hp & 3, py = sub >> 1, px = sub & 1;
          const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
          const int b = b0 + bl, oy = ty0 + ty, ox = tx0 + tx;
          if (b >= g.B || oy >= g.OY || ox >= g.OX) continue;
          const int R = oy * 2 + py, Cc = ox * 2 + px;
          const int64_t pix = ((int64_t)b * g.OHF + R) * g.OWF + Cc;
          const float x0 = img[pix * 6 + g.nll_ch], x1 = img[pix * 6 + g.nll_ch + 1], x2 = img[pix * 6 + g.nll_ch + 2];
          const float* lp = (const float*)(sC + r * srow) + sub * 6;
          f32x2 o01 = *(const f32x2*)lp, o23 = *(const f32x2*)(lp + 2), o45 = *(const f32x2*)(lp + 4);
          if (g.fix) {
            const int rc = R < 2 ? R : R >= g.OHF - 3 ? R - (g.OHF - 5) : -1;
            const int cg = ox == 0 ? 0 : ox == g.OX - 2 ? 1 : ox == g.OX - 1 ? 2 : -1;
            const float* fb = g.fix + (int64_t)b * (5 * g.OWF + 6 * g.OHF) * 6;
            if (rc >= 0) {
              const float* fp = fb + ((int64_t)rc * g.OWF + Cc) * 6;
              o01 += *(const f32x2*)fp; o23 += *(const f32x2*)(fp + 2); o45 += *(const f32x2*)(fp + 4);
            }
            if (cg >= 0) {
              const float* fp = fb + (int64_t)5 * g.OWF * 6 + (((int64_t)R * 3 + cg) * 2 + px) * 6;
              o01 += *(const f32x2*)fp; o23 += *(const f32x2*)(fp + 2); o45 += *(const f32x2*)(fp + 4);
            }
          }
          if (!g.nll_noout) {
            float* op = outp + pix * 6;
            *(f32x2*)op = o01; *(f32x2*)(op + 2) = o23; *(f32x2*)(op + 4) = o45;
          }
          float n0, n1, n2, dm0, dm1, dm2, dl0, dl1, dl2;
          dll_elem(x0, o01[0], o23[1], n0, dm0, dl0);        // channel k: mean o[k], log_scale o[3 + k] (vae/model.py:169)
          dll_elem(x1, o01[1], o45[0], n1, dm1, dl1);
          dll_elem(x2, o23[0], o45[1], n2, dm2, dl2);
          nacc += (n0 + n1) + n2;
          const float gs = g.nll_gscale;
          if constexpr (sizeof(T) == 2) {
            bf16x8 v;
            v[0] = (bf16_t)(dm0 * gs); v[1] = (bf16_t)(dm1 * gs); v[2] = (bf16_t)(dm2 * gs);
            v[3] = (bf16_t)(dl0 * gs); v[4] = (bf16_t)(dl1 * gs); v[5] = (bf16_t)(dl2 * gs);
            v[6] = (bf16_t)0.f; v[7] = (bf16_t)0.f;
            *(bf16x8*)(gp + pix * 8) = v;
          } else {                                            // fp32 step (round 6): the same record in floats, exactly dlogistic_kernel's values (dm * gscale, dl * gscale)
            float4* gq = (float4*)(gp + pix * 8);
            gq[0] = make_float4(dm0 * gs, dm1 * gs, dm2 * gs, dl0 * gs);
            gq[1] = make_float4(dl1 * gs, dl2 * gs, 0.f, 0.f);
          }
        }
        nacc = wave_sum(nacc);
        __syncthreads();                                    // every read of the transposed tile is done: reuse its head
        float* red = (float*)sC;
        if (lane == 0) red[wave] = nacc;
        __syncthreads();
        if (tid == 0 && b0 < g.B) {
          float s = 0.f;
          for (int k = 0; k < NW; ++k) s += red[k];
          const int tiles = g.tilesX * g.tilesY;
          g.nll_part[(int64_t)b0 * tiles + (ty0 >> g.lTH) * g.tilesX + (tx0 >> g.lTW)] = s;
        }
        return;
      }
    }
    for (int q0 = tid; q0 < total; q0 += 4 * NT) {
      float4 fr[4], fc[4];
      int64_t ob[4];
      int lo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * NT;
        const int r = q / ppr_o, c = q - r * ppr_o;
        const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
        const int b = b0 + bl, oy = ty0 + ty, ox = tx0 + tx;
        lo[u] = (q < total && b < g.B && oy < g.OY && ox < g.OX) ? r * srow + c * 16 : -1;
        const int py = c >= pps ? 1 : 0, R = oy * 2 + py, cs = c - py * pps;
        ob[u] = (((int64_t)b * g.OHF + R) * g.OWF + ox * 2) * C * 4 + cs * 16;
        fr[u] = fc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.fix && lo[u] >= 0) {
          const int rc = R < 2 ? R : R >= g.OHF - 3 ? R - (g.OHF - 5) : -1;                    // rows 0, 1, H-3, H-2, H-1 -> 0..4
          const int cg = ox == 0 ? 0 : ox == g.OX - 2 ? 1 : ox == g.OX - 1 ? 2 : -1;            // pixel pairs holding a border column
          const float* fb = g.fix + (int64_t)b * (5 * g.OWF + 6 * g.OHF) * C;
          if (rc >= 0) fr[u] = *(const float4*)(fb + ((int64_t)rc * g.OWF + ox * 2) * C + cs * 4);
          if (cg >= 0) fc[u] = *(const float4*)(fb + (int64_t)5 * g.OWF * C + ((int64_t)R * 3 + cg) * 2 * C + cs * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (lo[u] < 0) continue;
        float4 v = *(const float4*)(sC + lo[u]);
        v.x = (v.x + fr[u].x) + fc[u].x; v.y = (v.y + fr[u].y) + fc[u].y;       // row term first, then the column term: the
        v.z = (v.z + fr[u].z) + fc[u].z; v.w = (v.w + fr[u].w) + fc[u].w;       // order of the fused-loss loop above (bitwise equal out6)
        *(float4*)((char*)g.out + ob[u]) = v;
      }
    }
    return;
  }
  for (int q = tid; q < BM * ppr_o; q += NT) {
    const int r = q / ppr_o, c = q - r * ppr_o;
    const int tx = r & (TW - 1), ty = (r >> g.lTW) & (TH - 1), bl = r >> (g.lTW + g.lTH);
    const int b = b0 + bl, oy = ty0 + ty, ox = tx0 + tx;
    if (b >= g.B || oy >= g.OY || ox >= g.OX) continue;
    int64_t ob;                                          // byte offset in the output tensor
    if (g.cls_n) {                                       // merged parity classes: this piece's class picks the sub-pixel
      const int n = n0 + c * (psz / oesz), cls = n / g.cls_n, ch = n - cls * g.cls_n;
      const int64_t pix = ((int64_t)b * g.OHF + oy * g.OS + (cls >> 1)) * g.OWF + ox * g.OS + (cls & 1);
      ob = (pix * g.ldo + ch) * oesz;
    } else {
      const int64_t pix = ((int64_t)b * g.OHF + oy * g.OS + g.ooy) * g.OWF + ox * (g.d2s ? 2 : g.OS) + g.oox;
      ob = (pix * g.ldo + n0) * oesz + c * psz;
    }
    if (psz == 16) {
      uint4 v = *(const uint4*)(sC + r * srow + c * 16);
      if (g.mask) {                                     // mask tensor has the output's type and indexing (never fp32)
        const uint4 mv = *(const uint4*)((const char*)g.mask + ob);
        T ve[EPP], me[EPP];
        *(uint4*)ve = v; *(uint4*)me = mv;
#pragma unroll
        for (int e = 0; e < EPP; ++e) ve[e] = to_f32(me[e]) > 0.f ? ve[e] : from_f32<T>(0.f);
        v = *(uint4*)ve;
      }
      *(uint4*)((char*)g.out + ob) = v;
    } else {
      *(uint2*)((char*)g.out + ob) = *(const uint2*)(sC + r * srow + c * 8);
    }
  }
}

static inline size_t tile_lds_bytes(int BN, int BM, const TileConvArgs& a, size_t esz, int YR = 0) {
  size_t lds = (size_t)a.wslots * BN * tile_pps(BN, YR) * 16 + a.off_bytes + a.in_bytes;
  const size_t epi = (size_t)BM * (((BN * (a.out_f32 ? 4 : esz) + 15) & ~(size_t)15) + 16);   // epilogue transpose tile
  return lds < epi ? epi : lds;
}

template <typename T, int BN, int MF, int NW = 4, int YR = 0, bool WR = false, bool FX = false>
static int launch_tile(const TileConvArgs* a, int n, hipStream_t st) {
  const int Npad = round_up(a[0].N, BN);
  dim3 grid(a[0].ntiles, Npad / BN, n), block(64 * NW);
  size_t lds = 0;
  TileConvMulti m;
  for (int i = 0; i < n; ++i) {
    m.a[i] = a[i];
    const size_t l = tile_lds_bytes(BN, NW * 16 * MF, a[i], sizeof(T), YR);
    if (l > lds) lds = l;
  }
  sv_ensure_dynamic_lds((const void*)tile_conv_kernel<T, BN, MF, NW, YR, WR, FX>, lds);
  hipLaunchKernelGGL((tile_conv_kernel<T, BN, MF, NW, YR, WR, FX>), grid, block, lds, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// Plans the tiling for a tap-GEMM problem; returns false when the problem does not fit the
// direct kernel (dense layers, huge channel counts, K-split needed) and the caller should use
// the im2col kernel instead.
static bool tile_conv_plan_impl(const TapGemmArgs& t, int dtype, int B, TileConvArgs* a, int* cfg_out, bool narrow, int64_t* lds_out);
bool svk_tile_conv_plan(const TapGemmArgs& t, int dtype, int B, TileConvArgs* a, int* cfg_out) {
  int64_t lds = 0;
  if (!tile_conv_plan_impl(t, dtype, B, a, cfg_out, false, &lds)) return false;
  // fp32: a 128-column tile that leaves ONE workgroup per CU (> 78 KB of LDS: e3's stride-2 forward, 2 x 18 x 18 padded pixels + 32 KB of weight slots) runs on
  // 64-column tiles if those fit twice -- nothing overlaps a lone workgroup's staging and epilogue (fwd.e3 0.201 -> 0.181 ms at 2 x 512 images; SV_TC_NO_LDS_NARROW=1: off)
  static const bool no_narrow = getenv("SV_TC_NO_LDS_NARROW") != nullptr;
  if (!no_narrow && dtype == SV_F32 && t.N % 128 == 0 && !t.cls_n && lds > 78 * 1024) {
    TileConvArgs b;
    int c = 0;
    int64_t l2 = 0;
    if (tile_conv_plan_impl(t, dtype, B, &b, &c, true, &l2) && l2 <= 78 * 1024) { *a = b; *cfg_out = c; }
  }
  return true;
}
static bool tile_conv_plan_impl(const TapGemmArgs& t, int dtype, int B, TileConvArgs* a, int* cfg_out, bool narrow, int64_t* lds_out) {
  if (t.splitk != 1 || t.accum) return false;   // (accumulating fp32 targets: the im2col kernel's epilogue adds)
  if (t.lOY < 0 || t.lOX < 0 || t.S > 2) return false;    // power-of-two grids, stride <= 2 (the tile maps shift and mask)
  const int OY = 1 << t.lOY, OX = 1 << t.lOX;
  if (t.ups && t.S != 1) return false;
  if (t.d2s && ((t.N != 16 && !(t.N == 32 && t.d2s_y)) || !t.out_f32)) return false;
  if (t.d2s_y && (!t.d2s || t.N != 32 || ((2 * t.d2s * 4) & 7))) return false;
  if (t.clampin && (t.ups || t.S != 1)) return false;
  if (t.s2d3 && (dtype != SV_F32 || t.S != 1 || t.SX != 1 || t.ups || t.clampin || t.cl2 != 2)) return false;
  if (t.fix_nc && !t.d2s_y && ((dtype != SV_F32 && t.S != 2) || !t.fix || !t.fix2 || (t.N & 15) || t.out_f32)) return false;     // (bf16: only the stride-2 polyphase input gradient has instantiations with the border-term epilogue)
  if (t.nll_part && (!t.d2s_y || t.d2s != 6 || OY * OX < 256 || !t.nll_img || !t.nll_grad)) return false;
  if (t.cls_n && (t.OS != 2 || t.N != 4 * t.cls_n || (t.cls_n & 7) || t.out_f32 || t.bias)) return false;
  if (OY * OX < 16) return false;                       // dense / tiny spatial: im2col path
  if (!t.d2s && !t.cls_n) {                             // the epilogue stores 8- / 16-byte pieces (see Nst in the kernel)
    const int oe = t.out_f32 ? 4 : (dtype == SV_BF16 ? 2 : 4);
    const int nst = ((t.N * oe) & 7) ? (t.ldo < ((t.N + 7) & ~7) ? t.ldo : ((t.N + 7) & ~7)) : t.N;
    if ((nst * oe) & 7) return false;
  }
  const int esz = dtype == SV_BF16 ? 2 : 4, epp = 16 / esz;
  const int cin = (1 << t.cl2) * epp;
  // N tile
  int BN, cfgN;
  // small launches (up to 256-image shards): a 128-column layer whose 128-row tiles give at most about one workgroup per CU
  // runs on 64-column tiles instead -- twice the workgroups, half the weight streaming each (SV_TC_SMALL_WGS: the threshold)
  static const int small_wgs = getenv("SV_TC_SMALL_WGS") ? atoi(getenv("SV_TC_SMALL_WGS")) : 200;   // (per problem) measured: helps at 128 such workgroups per network (B = 256: -1.6 %; 128: -3.6 %, 64: -3.4 %), hurts at 256 (B = 512: +1.8 %)
  const int64_t wgs128 = (((int64_t)B * OY * OX + 127) / 128) * (t.N / 128);
  // (fp32 too since round 5: at 64 images per network d2 / e3 -- 8 x 8 grids, 128 columns -- ran 64 workgroups on 256 CUs, 14 % of the fp32 matrix peak; SV_TC_SMALL_F32=0: off)
  static const bool small_f32 = !(getenv("SV_TC_SMALL_F32") && atoi(getenv("SV_TC_SMALL_F32")) == 0);
  const bool small = t.N % 128 == 0 && (wgs128 < small_wgs || narrow) && (dtype == SV_BF16 || small_f32) && !t.cls_n;
  static const int tiny_wgs = getenv("SV_TC_TINY_WGS") ? atoi(getenv("SV_TC_TINY_WGS")) : 100;   // ... and on 32-column tiles below this (64-image shards: -1.3 .. -1.9 %)
  const bool tiny = small && wgs128 < tiny_wgs;
  // (round 6) the same for 64-column stride-1 layers: d3 of a 64-image SVHN-32 shard ran 64-128 workgroups of 128 x 64 on 256 CUs (89 us); 32-column tiles below
  // SV_TC_SMALL64_WGS workgroups per problem (fp32; 0: off).  Measured (profiles/r06_small64.txt): SVHN-32 64 images 1.062-1.074 -> 1.029-1.033 ms, CelebA-64 64 images
  // +-0; the stride-2 64-column layer (e2) LOSES on 32-column tiles (CelebA-64 64 images 1.69 -> 1.71): stride 1 only (SV_TC_SMALL64_S: 0 every stride, 2 stride 2)
  static const int small64_wgs = getenv("SV_TC_SMALL64_WGS") ? atoi(getenv("SV_TC_SMALL64_WGS")) : 300;
  const int64_t wgs64 = (((int64_t)B * OY * OX + 127) / 128) * (t.N / 64);
  static const int small64_s = getenv("SV_TC_SMALL64_S") ? atoi(getenv("SV_TC_SMALL64_S")) : 1;
  // (forward layers only -- OS == 1: the parity-class problems of a stride-2 input gradient are stride-1 problems too, and e3's at 512 images, 256 workgroups
  //  each, went 0.172 -> 0.201 ms on 32-column tiles)
  // (the stride-2 polyphase input gradient of d4 on 32-column tiles: measured, slower -- 64 images 1.679 -> 1.694 ms, 128: 2.732 -> 2.783)
  const bool tiny64 = dtype == SV_F32 && t.N % 64 == 0 && t.N % 128 != 0 && wgs64 < small64_wgs && !t.cls_n && !t.d2s && t.OS == 1 && (!small64_s || t.S == small64_s);
  if (t.N % 128 == 0 && !small) { BN = 128; cfgN = 0; }
  else if (t.N % 64 == 0 && !tiny && !tiny64) { BN = 64; cfgN = 1; }
  else if (t.N % 32 == 0) { BN = 32; cfgN = 2; }
  else if (t.N <= 16) { BN = 16; cfgN = 3; }
  else return false;
  int y_lo = 127, y_hi = -127, x_lo = 127, x_hi = -127;
  for (int i = 0; i < t.ntaps; ++i) {
    y_lo = t.dy[i] < y_lo ? t.dy[i] : y_lo; y_hi = t.dy[i] > y_hi ? t.dy[i] : y_hi;
    x_lo = t.dx[i] < x_lo ? t.dx[i] : x_lo; x_hi = t.dx[i] > x_hi ? t.dx[i] : x_hi;
  }
  // LDS bytes per pixel: +32 B makes PS an odd multiple of 32, which spreads the 16 pixels of a
  // ds_read_b128 fragment over all 64 banks (linear 64/128/256-B pixels are 2/4/8-way conflicted).
  // Measured: pays for >= 128-B pixels at stride 1; for 64-B pixels the extra LDS costs more in
  // occupancy than the 2-way conflict, and at stride 2 no padding can make 2*PS/32 odd.
  // At stride 2 the fragment's pixels are 2*PS apart: +16 B makes 2*PS an odd multiple of 32 (linear
  // 64/128-B pixels are 4/8-way conflicted there).  scripts/lds_bank_model.py has the lane-group model.
  static const bool s2pad = getenv("SV_TC_NO_S2PAD") == nullptr, p64 = getenv("SV_TC_PAD64") != nullptr;
  const int pb = cin * esz;
  // Stride-1 tiles of >= 64-B pixels are PLANAR (32-B planes: conflict-free with no padding; SV_TC_NO_PLANAR = the
  // padded linear layout for A/B)
  static const bool planar_on = getenv("SV_TC_NO_PLANAR") == nullptr;
  // layout of a pixel record of `bytes` channel bytes: planar (PS = 32) or linear with conflict-avoiding padding
  auto layout = [&](int bytes, bool* planar_out) {
    const bool pl = planar_on && t.SX == 1 && t.S == 1 && bytes >= 64;
    *planar_out = pl;
    return pl ? 32 : bytes + (t.SX == 1 ? ((bytes >= 128 || (bytes == 64 && p64)) ? 32 : 0) : ((bytes % 32 == 0 && s2pad) ? 16 : 0));
  };
  const int lTW = OX >= 16 ? 4 : t.lOX;
  const int off_bytes = (((t.P + 31) / 32 * 32) * 4 + 15) / 16 * 16;   // padded to the largest K step
  // try MF = 4 (256-row tile) then MF = 2 (128 rows); BN = 128 only with MF = 2, BN = 16/32 only with MF = 4
  static const char* mf2 = getenv("SV_TC_MF2");       // tuning knob: BN values (as letters a=16,b=32,c=64) forced to 128-row tiles
  for (int MF = 4; MF >= 2; MF -= 2) {
    // 256 x 128 tiles (SV_TC_BN128_MF4=1): half the weight streaming per output of the small-grid 128-column layers
    static const bool big128 = getenv("SV_TC_BN128_MF4") != nullptr;
    if (MF == 4 && BN == 128 && !(big128 && dtype == SV_BF16)) continue;
    if (MF == 4 && mf2 && strchr(mf2, BN == 16 ? 'a' : BN == 32 ? 'b' : 'c')) continue;
    if (MF == 4 && small) continue;                       // (128-row tiles: the point is more workgroups)
    if (MF == 4 && BN == 32 && OY * OX <= 256 && t.OS == 2 && !t.d2s_y) continue;   // measured: e2's dgrad parity classes (16x16 grids) run 20 % faster on 128-row tiles
    const int BM = 64 * MF;
    int lTH = 0;
    while ((1 << (lTW + lTH)) < BM && (1 << lTH) < OY) ++lTH;
    int lNB = 0;
    while ((1 << (lTW + lTH + lNB)) < BM) ++lNB;
    const int TW = 1 << lTW, TH = 1 << lTH, NB = 1 << lNB;
    if (t.nll_part && (BM != 256 || NB != 1)) return false;   // the fused loss writes one partial per 256-pixel tile of one image
    const int TIW = (TW - 1) * t.SX + (x_hi - x_lo) + 1, TIH = (TH - 1) * t.S + (y_hi - y_lo) + 1;
    // channel phases (see the kernel): halve the channels per staging pass while the tile alone would keep a CU at
    // <= 2 workgroups, the launch has several rounds of workgroups, and a phase keeps >= 32 B (two pieces) per pixel
    // row-window reuse (kernel comment, YR): stride-1 rows, 16-pixel tile rows, x-major taps in columns of KH = 4 / 6
    // consecutive dy, >= 32 channels (a phase = exactly 32: one tap per MFMA)
    int yr = 0;
    // off for the 16-column kernel (BN letter a): the packed d5's un-phased 62 KB tile leaves two workgroups per CU
    // instead of three, which costs what the window saves (0.200 vs 0.191 ms in the step)
    static const char* yr_off = getenv("SV_TC_NO_YR") ? getenv("SV_TC_NO_YR") : "a";   // "1" = off everywhere, or BN letters (a=16, b=32, c=64)
    if (dtype == SV_BF16 && MF == 4 && t.S == 1 && lTW == 4 && TH >= MF && cin >= 32 && BN <= 64 &&
        !(yr_off && (yr_off[0] == '1' || strchr(yr_off, BN == 16 ? 'a' : BN == 32 ? 'b' : 'c')))) {
      int kh = 1;
      while (kh < t.ntaps && t.dx[kh] == t.dx[0]) ++kh;
      static const bool yr5 = getenv("SV_TC_YR5") != nullptr;         // A/B (off: measured 0.155 vs 0.141 ms for the polyphase head -- a step is one
                                                                       // whole 5-tap column, 20 pieces: 128 VGPRs, 7 spilled, bigger weight slots)
      bool ok = (kh == 4 || kh == 6 || (kh == 5 && yr5 && BN == 32)) && t.ntaps % kh == 0;
      for (int i = 0; i < t.ntaps && ok; ++i)
        ok = t.dx[i] == t.dx[i / kh * kh] && t.dy[i] == t.dy[0] + i % kh && t.dy[0] == y_lo;
      if (ok) yr = kh;
    }
    static const int nph_max = getenv("SV_TC_NPH") ? atoi(getenv("SV_TC_NPH")) : 4;     // tuning knob (1 = off)
    const int64_t wgs = (int64_t)(OX / TW) * (OY / TH) * ((B + NB - 1) / NB) * ((t.N + BN - 1) / BN);
    // (round 6) fp32 32-column layers of small launches: 256-row tiles left 64-128 workgroups per problem on 256 CUs (SVHN-32, 64 images: e1 / the d4 class
    // problems); 128-row tiles below SV_TC_SMALL32_WGS workgroups per problem.  SVHN-32 64 images 1.029-1.038 -> 0.980 ms, 256 images 1.79 -> 1.78; CelebA-64 64 images
    // 1.689 -> 1.680, 128 images +-0 (forcing 128-row tiles on EVERY 32-column layer, SV_TC_MF2=b, costs CelebA-64 64 images 1 %).  Not the fused-loss head (one
    // partial per 256-pixel tile).  profiles/r06_small64.txt
    static const int small32_wgs = getenv("SV_TC_SMALL32_WGS") ? atoi(getenv("SV_TC_SMALL32_WGS")) : 300;
    if (MF == 4 && BN == 32 && dtype == SV_F32 && !t.nll_part && wgs < small32_wgs) continue;
    int lnph = 0, PS = 0, plane_bytes = 0;
    int64_t in_bytes = 0;
    bool planar = false;
    for (;; ++lnph) {
      if (yr) lnph = t.cl2 - 2;                        // 4 pieces (32 channels) per tap and phase
      const int pbh = pb >> lnph;
      PS = layout(pbh, &planar);
      plane_bytes = planar ? NB * TIH * TIW * 32 : 0;
      in_bytes = planar ? (int64_t)plane_bytes * (pbh / 32) : (int64_t)NB * TIH * TIW * PS;
      static const int ph_kb = getenv("SV_TC_PH_KB") ? atoi(getenv("SV_TC_PH_KB")) : 53;   // LDS per workgroup the split aims below (3 workgroups per CU)
      static const bool s2_phases = getenv("SV_TC_NPH_NO_S2") == nullptr;               // A/B: phases for the padded stride-2 layouts too
      static const int ph_wgs = getenv("SV_TC_PH_WGS") ? atoi(getenv("SV_TC_PH_WGS")) : 256;   // launches smaller than this keep one pass
      // (a tile that does not fit the CU at all is split whatever the launch size: e3's stride-2 forward at fp32 -- 2 x 18 x 18 pixels of 528 B -- fell to the
      //  im2col kernel below 256 workgroups: 0.102 ms for 64 images per network against 0.198 ms for 512)
      const int64_t tot = in_bytes + 2 * BN * tile_pps(BN) * 16 + off_bytes;
      if (yr || t.s2d3 || !((2 << lnph) <= nph_max && (pbh >> 1) >= 32 && (planar || s2_phases) &&
                            ((wgs >= ph_wgs && tot > ph_kb * 1024) || tot > 150 * 1024))) break;
    }
    // whole K resident in LDS (kernel comment): one phase, <= 6 K steps, <= 24 KB of weights
    static const bool wres_on = getenv("SV_TC_NO_WRES") == nullptr;
    const int nks = (int)(((t.P >> lnph) + tile_pps(BN) - 1) / tile_pps(BN));
    // (measured: d5's input gradient -5 %; e1's stride-2 forward +10 %, so stride 1 only)
    const int wslots = (wres_on && !yr && dtype == SV_BF16 && BN == 32 && MF == 4 && t.S == 1 && lnph == 0 && nks > 2 && nks <= 6 && nks * BN * tile_pps(BN) * 16 <= 24 * 1024) ? nks : 2;
    const int64_t lds = (int64_t)wslots * BN * tile_pps(BN, yr) * 16 + off_bytes + in_bytes;
    if (lds > 78 * 1024 && MF == 4 && BN >= 64) continue;   // prefer 2 workgroups per CU: retry with 128 rows
    // (the stride-2 9 x 9-tap polyphase input gradient, conv_geom.h svg_polyd: its 39 x 39-pixel hi-res tile leaves ONE workgroup per CU -- no overlap
    //  of one tile's staging with another's MFMAs; SV_TC_S2_MF4=1: the 256-row tile for A/B)
    static const bool s2_mf4 = getenv("SV_TC_S2_MF4") != nullptr;
    if (lds > 78 * 1024 && MF == 4 && t.S == 2 && t.fix_nc && !s2_mf4) continue;
    *lds_out = lds;
    if (lds > 150 * 1024) {
      if (MF == 4) continue;
      if (getenv("SV_TC_VERBOSE")) fprintf(stderr, "tile_conv plan: REFUSED (LDS %lld) N=%d cin=%d S=%d ntaps=%d\n", (long long)lds, t.N, cin, t.S, t.ntaps);
      return false;
    }
    memset(a, 0, sizeof(*a));
    a->A = t.A; a->Wt = t.Wt; a->bias = t.bias; a->out = t.out; a->mask = t.mask;
    a->B = B; a->IH = t.IH; a->IW = t.IW; a->lda = t.lda;
    a->cl2 = t.cl2; a->P = t.P; a->Ktot = t.Ktot; a->S = t.S; a->SX = t.SX; a->d2s = t.d2s; a->cls_n = t.cls_n; a->d2s_y = t.d2s_y; a->clampin = t.clampin; a->fix = t.fix;
    a->fix2 = t.fix2; a->fix_nc = t.d2s_y ? 0 : t.fix_nc; a->fix_pad = t.fix_pad; a->s2d3 = t.s2d3;
    a->nll_img = t.nll_img; a->nll_grad = t.nll_grad; a->nll_part = t.nll_part; a->nll_ch = t.nll_ch; a->nll_gscale = t.nll_gscale; a->nll_noout = t.nll_noout;
    a->lTW = lTW; a->lTH = lTH; a->lNB = lNB;
    a->OY = OY; a->OX = OX;
    a->tilesX = OX / TW; a->tilesY = OY / TH;
    a->ntiles = a->tilesX * a->tilesY * ((B + NB - 1) / NB);
    a->TIW = TIW; a->TIH = TIH; a->y_lo = y_lo; a->x_lo = x_lo; a->PS = PS; a->plane_bytes = plane_bytes;
    a->nph = 1 << lnph; a->lnph = lnph; a->wslots = wslots;
    {
      // (fp32: the register staging keeps four loads per lane in flight and waits -- three to four dependent L2 round trips per tile; SV_TC_NO_DMA: A/B)
      static const bool no_dma = getenv("SV_TC_NO_DMA") != nullptr;
      a->dma = (!no_dma && dtype == SV_F32 && !t.ups && !t.s2d3 && (planar || PS == (pb >> lnph))) ? 1 : 0;
    }
    if (getenv("SV_TC_VERBOSE")) fprintf(stderr, "tile_conv plan: yr=%d wslots=%d N=%d BN=%d MF=%d cin=%d pb=%d planar=%d nph=%d tile=%dx%dx%d PS=%d in_bytes=%lld lds=%lld ntiles=%d S=%d SX=%d\n", yr, wslots, t.N, BN, MF, cin, pb, (int)planar, 1 << lnph, NB, TIH, TIW, PS, (long long)in_bytes, (long long)lds, a->ntiles, t.S, t.SX);
    {
      static const bool xcd = getenv("SV_TC_NO_XCD") == nullptr;
      a->xcd_chunk = (xcd && a->ntiles >= 64 && (a->ntiles & 7) == 0) ? a->ntiles / 8 : 0;
    }
    a->off_bytes = off_bytes; a->in_bytes = (int)in_bytes;
    a->N = t.N; a->OHF = t.OHF; a->OWF = t.OWF; a->OS = t.OS; a->ooy = t.ooy; a->oox = t.oox; a->ldo = t.ldo;
    a->act = t.act; a->out_f32 = t.out_f32; a->ntaps = t.ntaps; a->ups = t.ups;
    memcpy(a->dy, t.dy, sizeof(a->dy));
    memcpy(a->dx, t.dx, sizeof(a->dx));
    *cfg_out = cfgN * 2 + (MF == 4 ? 0 : 1) + (yr == 4 ? 32 : yr == 6 ? 64 : yr == 5 ? 96 : 0) + (wslots > 2 ? 128 : 0);
    // 256-row tiles that leave room for at most two workgroups per CU: 8 waves share the tile
    static const char* nw8 = getenv("SV_TC_NW8");       // tuning knob: BN classes (a=16, b=32, c=64) run with 8-wave workgroups
    if (MF == 4 && dtype == SV_BF16 && nw8 && strchr(nw8, BN == 16 ? 'a' : BN == 32 ? 'b' : BN == 64 ? 'c' : 'd')) *cfg_out = 16 + cfgN;
    return true;
  }
  return false;
}

// n problems of identical geometry (same cfg / tile grid / N) in one launch
int svk_tile_conv_multi(const TileConvArgs* a, int n, int dtype, int cfg, hipStream_t st) {
  // (A persistent variant that staged tile t+1 in per-K-step slices into a second LDS buffer was
  // built and measured 2-2.5x SLOWER: CDNA's vmcnt retires in order, so every K-step's weight-tile
  // wait also waited for that step's HBM slice loads, and the second buffer halved the resident
  // workgroups.  Overlap needs producer waves with their own load queue, not in-loop slices.)
  if (n < 1 || n > SV_MAX_MULTI) return SV_E_BADARG;
  if (dtype == SV_BF16 && a[0].fix_nc) {              // the polyphase input gradient at bf16 (SV_POLYD_BF16=1): the border-term epilogue compiled in
    switch (cfg) {
      case 2: return launch_tile<bf16_t, 64, 4, 4, 0, false, true>(a, n, st);
      case 3: return launch_tile<bf16_t, 64, 2, 4, 0, false, true>(a, n, st);
      case 4: return launch_tile<bf16_t, 32, 4, 4, 0, false, true>(a, n, st);
      case 5: return launch_tile<bf16_t, 32, 2, 4, 0, false, true>(a, n, st);
    }
    return SV_E_UNSUPPORTED;
  }
  if (dtype == SV_BF16) {
    switch (cfg) {
      case 0: return launch_tile<bf16_t, 128, 4>(a, n, st);
      case 1: return launch_tile<bf16_t, 128, 2>(a, n, st);
      case 2: return launch_tile<bf16_t, 64, 4>(a, n, st);
      case 3: return launch_tile<bf16_t, 64, 2>(a, n, st);
      case 4: return launch_tile<bf16_t, 32, 4>(a, n, st);
      case 5: return launch_tile<bf16_t, 32, 2>(a, n, st);
      case 7: return launch_tile<bf16_t, 16, 2>(a, n, st);
      case 6: return launch_tile<bf16_t, 16, 4>(a, n, st);
      case 128 + 4: return launch_tile<bf16_t, 32, 4, 4, 0, true>(a, n, st);   // whole K resident in LDS
      case 32 + 2: return launch_tile<bf16_t, 64, 4, 4, 4>(a, n, st);     // row-window reuse, KH = 4
      case 32 + 4: return launch_tile<bf16_t, 32, 4, 4, 4>(a, n, st);
      case 32 + 6: return launch_tile<bf16_t, 16, 4, 4, 4>(a, n, st);
      case 96 + 4: return launch_tile<bf16_t, 32, 4, 4, 5>(a, n, st);     // KH = 5 (polyphase head)
      case 64 + 2: return launch_tile<bf16_t, 64, 4, 4, 6>(a, n, st);     // KH = 6
      case 64 + 4: return launch_tile<bf16_t, 32, 4, 4, 6>(a, n, st);
      case 64 + 6: return launch_tile<bf16_t, 16, 4, 4, 6>(a, n, st);
      case 16: return launch_tile<bf16_t, 128, 2, 8>(a, n, st);   // (SV_TC_BN128_MF4=1 SV_TC_NW8=d)
      case 17: return launch_tile<bf16_t, 64, 2, 8>(a, n, st);
      case 18: return launch_tile<bf16_t, 32, 2, 8>(a, n, st);
      case 19: return launch_tile<bf16_t, 16, 2, 8>(a, n, st);
    }
  } else if (dtype == SV_F32) {
    switch (cfg) {
      case 1: return launch_tile<float, 128, 2>(a, n, st);
      case 2: return launch_tile<float, 64, 4>(a, n, st);
      case 3: return launch_tile<float, 64, 2>(a, n, st);
      case 4: return launch_tile<float, 32, 4>(a, n, st);
      case 5: return launch_tile<float, 32, 2>(a, n, st);
      case 7: return launch_tile<float, 16, 2>(a, n, st);
      case 6: return launch_tile<float, 16, 4>(a, n, st);
    }
  }
  return SV_E_BADARG;
}

int svk_tile_conv(const TileConvArgs& a, int dtype, int cfg, hipStream_t st) { return svk_tile_conv_multi(&a, 1, dtype, cfg, st); }

// n tap-GEMM problems of the same shape: one multi launch of the tile kernel when they all plan
// to the same configuration, individual launches otherwise
int svk_conv_dispatch_multi(const TapGemmArgs* t, int n, int dtype, int tap_cfg, hipStream_t st) {
  static const bool force_tap = getenv("SV_FORCE_IM2COL") != nullptr;   // A/B switch for tests and profiling
  static const bool no_multi = getenv("SV_NO_MULTI") != nullptr;        // A/B: one launch per problem
  static const int dbg = SV_DBG(getenv("SV_TC_DBG") ? atoi(getenv("SV_TC_DBG")) : 0);
  if (n < 1 || n > SV_MAX_MULTI) return SV_E_BADARG;
  if (!force_tap && n <= 8) {                          // weights in registers, rows rolling through LDS (row_conv.hip takes one problem, the x / x-hat
                                                       // twins, or the up to eight class problems of a stride-2 layer's input gradient)
    const int rc = svk_row_conv_try(t, n, dtype, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  for (int i = 0; i < n; ++i)
    if (t[i].adj) return SV_E_UNSUPPORTED;            // the fused resize adjoint exists on the row-ring kernel only: the caller
                                                      // runs the plain input gradient and sv_upsample2x_bwd instead
  TileConvArgs a[SV_MAX_MULTI];
  int cfg[SV_MAX_MULTI];
  bool all_tile = !force_tap;
  for (int i = 0; i < n && all_tile; ++i) {
    all_tile = svk_tile_conv_plan(t[i], dtype, t[i].M / (t[i].OY * t[i].OX), &a[i], &cfg[i]);
    a[i].dbg = dbg;
    if (all_tile && i > 0 && (cfg[i] != cfg[0] || a[i].ntiles != a[0].ntiles || a[i].N != a[0].N)) all_tile = false;
  }
  if (all_tile && !no_multi) return svk_tile_conv_multi(a, n, dtype, cfg[0], st);
  // none of them plans to the tile kernel (dense layers, 1x1 grids): im2col GEMMs, SV_TAP_MAX_MULTI per launch
  bool none_tile = !no_multi && n > 1;
  for (int i = 0; i < n && none_tile; ++i) {
    TileConvArgs b;
    int c;
    none_tile = !(t[i].ups || t[i].d2s || t[i].cls_n || t[i].clampin || t[i].fix_nc || t[i].s2d3) && (force_tap || !svk_tile_conv_plan(t[i], dtype, t[i].M / (t[i].OY * t[i].OX), &b, &c));
  }
  if (none_tile) {
    for (int i = 0; i < n; i += SV_TAP_MAX_MULTI) {
      const int rc = svk_tap_gemm_multi(t + i, n - i < SV_TAP_MAX_MULTI ? n - i : SV_TAP_MAX_MULTI, dtype, tap_cfg, st);
      if (rc) return rc;
    }
    return SV_OK;
  }
  for (int i = 0; i < n; ++i) {
    int rc;
    TileConvArgs b;
    int c;
    if (!force_tap && svk_tile_conv_plan(t[i], dtype, t[i].M / (t[i].OY * t[i].OX), &b, &c)) {
      b.dbg = dbg;
      rc = svk_tile_conv(b, dtype, c, st);
    } else if (t[i].ups || t[i].d2s || t[i].cls_n || t[i].clampin || t[i].fix_nc || t[i].s2d3) {
      rc = SV_E_UNSUPPORTED;               // the im2col kernel needs the materialised hi-res tensor / has no depth-to-space store
    } else {
      rc = svk_tap_gemm(t[i], dtype, tap_cfg, st);
    }
    if (rc) return rc;
  }
  return SV_OK;
}

int svk_conv_dispatch(const TapGemmArgs& t, int dtype, int tap_cfg, hipStream_t st) {
  return svk_conv_dispatch_multi(&t, 1, dtype, tap_cfg, st);
}
