// POLYPHASE INPUT GRADIENT of the upsample -> conv layers (conv_geom.h: svg_polyd has the algebra; tests/test_polyphase_math.py pins it):
//   Conv2DBackpropInput + ResizeBilinearGrad + ReluGrad of vae/model.py:163-167 (d4 / d5) in one pass, delivered at the LOW-RES tensor.
// The main term -- one stride-2 conv with 9 x 9 taps over the hi-res dY -- runs on the tile kernel (tile_conv.hip); this file holds
//   * polyd_edge_kernel: the corrections of the first / last low-res row and column (zero padding of the upsampled image + the resize's edge clamp):
//     per edge a conv along the border strip of dY (up to four hi-res rows / columns from the edge) with (row from the edge q, hi-res offset d) taps,
//     stride 2 along the strip, K = dY channels.  A workgroup owns (edge, 16 input channels, a group of images); wave q holds the 2R+1 taps of strip
//     row q in registers, the four partial sums are added through LDS in wave order (deterministic).  Output in the layout of the epilogue's border
//     terms: rows [B][2][w][Cin], columns [B][h][2][Cin];
//   * polyd_corner_kernel: the four corner pixels' cross term (<= 4 x 4 dY pixels each), one thread per (corner, input channel), added into the row terms.
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"
#include "conv_geom.h"
#include "fix_mma.hip.h"

namespace {

struct PolydEdgeMulti { const void* dy[2]; const void* wedge[2]; const void* wcorner[2]; float* erow[2]; float* ecol[2]; };

// NT = 2R+1 taps along the strip, NGRP = MFMA groups over the (padded) dY channels, NLI = strip pieces per thread and image (host-checked bound).
// One image per barrier pair; the NEXT image's strip is fetched into registers before the current image's MFMAs and written to LDS after them (the loop is
// latency otherwise: 174 us per launch at 2 x 512 images for 8 GFLOP).
template <typename T, int NT, int NGRP, int NLI, int DEPTH>
__global__ __launch_bounds__(256) void polyd_edge_kernel(const PolydEdgeMulti mg, int B, int h, int w, int Cin, int gdy, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPP = ElemTraits<T>::EPP, CPG = FixMma<T>::CPG, COP = NGRP * CPG, NPC = COP / EPP, R = (NT - 1) / 2;
  constexpr int PSB = COP * (int)sizeof(T) + 16;              // strip pixel pitch (+16: lanes two pixels apart on different banks)
  const T* __restrict__ dy = (const T*)mg.dy[blockIdx.z];
  const T* __restrict__ wedge = (const T*)mg.wedge[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int ncif = Cin >> 4, e = (int)blockIdx.x / ncif, cif = (int)blockIdx.x % ncif;       // edge: 0 top, 1 bottom, 2 left, 3 right
  const bool rows = e < 2, hi_edge = e & 1;
  const int H2 = 2 * h, W2 = 2 * w, pad = (K - 1) / 2;
  const int nq = hi_edge ? K - pad : pad + 1;                 // strip rows (distance 0 .. nq-1 from the edge)
  const int n = rows ? w : h, L2 = 2 * n, SW = L2 + 2 * R;    // low-res positions along the strip; strip width with the zero halo
  // this wave's taps: strip row q = wave
  uint4 wv[NT][NGRP];
  {
    const T* wp = wedge + ((((int64_t)e * 4 + wave) * NT) * Cin + cif * 16 + lr) * COP + lg * EPP;
#pragma unroll
    for (int d = 0; d < NT; ++d)
#pragma unroll
      for (int gq = 0; gq < NGRP; ++gq) wv[d][gq] = *(const uint4*)(wp + (int64_t)d * Cin * COP + gq * CPG);
  }
  char* sStrip = smem;                                         // [4][SW] pixels of PSB bytes
  const int npf = n >> 4;
  float* sRed = (float*)(smem + 4 * SW * PSB);                // [4 waves][npf][256] partial sums
  // ---- this thread's strip pieces (the same for every image): source element offset inside the image (-1: zero) and LDS slot
  int s_src[NLI], s_dst[NLI];
#pragma unroll
  for (int s = 0; s < NLI; ++s) {
    const int it = tid + s * 256;
    s_src[s] = -1; s_dst[s] = -1;
    if (it < 4 * SW * NPC) {
      const int ch = it % NPC, a = (it / NPC) % SW - R, q = it / (NPC * SW);
      s_dst[s] = (q * SW + a + R) * PSB + ch * 16;
      if (q < nq && a >= 0 && a < L2 && ch * EPP < gdy) {
        const int across = hi_edge ? (rows ? H2 : W2) - 1 - q : q;
        s_src[s] = (rows ? across * W2 + a : a * W2 + across) * gdy + ch * EPP;
      }
    }
  }
  // DEPTH 2 (round 6, opt-in): two images' strips in flight per thread -- register set A holds image b, B image b + 1, a set is refilled with image b + 2 as soon as
  // it has been written to LDS.  The loop is one global-load latency per image at DEPTH 1; the second register set costs a resident workgroup, which costs more (launcher).
  uint4 rsA[NLI], rsB[NLI];
  auto fetch = [&](int b, uint4 (&rs)[NLI]) {
    const T* dyb = dy + (int64_t)b * H2 * W2 * gdy;
#pragma unroll
    for (int s = 0; s < NLI; ++s) rs[s] = s_src[s] >= 0 ? *(const uint4*)(dyb + s_src[s]) : make_uint4(0, 0, 0, 0);
  };
  const int per = (B + (int)gridDim.y - 1) / (int)gridDim.y, b_lo = (int)blockIdx.y * per, b_hi = min(B, b_lo + per);
  if (b_lo < b_hi) fetch(b_lo, rsA);
  if (DEPTH == 2 && b_lo + 1 < b_hi) fetch(b_lo + 1, rsB);
  auto image = [&](int b, uint4 (&rs)[NLI]) {
    __syncthreads();                                           // the previous image's strip and partial sums are consumed
#pragma unroll
    for (int s = 0; s < NLI; ++s)
      if (s_dst[s] >= 0) *(uint4*)(sStrip + s_dst[s]) = rs[s];
    __syncthreads();
    if (b + DEPTH < b_hi) fetch(b + DEPTH, rs);                // in flight during this image's (and, DEPTH 2, the next image's) MFMAs
    for (int pf = 0; pf < npf; ++pf) {                         // every wave its strip row
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (wave < nq) {
        const char* sp = sStrip + (wave * SW + 2 * (16 * pf + lr)) * PSB + lg * 16;      // pixel 2 pos + d (d = 0 .. 2R: the halo shifts it by R)
#pragma unroll
        for (int d = 0; d < NT; ++d)
#pragma unroll
          for (int gq = 0; gq < NGRP; ++gq) FixMma<T>::run(wv[d][gq], *(const uint4*)(sp + d * PSB + gq * (CPG * (int)sizeof(T))), acc);
        mfma_result_fence(acc);                                  // (fix_mma.hip.h: the store below sat two instructions behind the last v_mfma, across an s_branch)
      }
      *(float4*)(sRed + ((wave * npf + pf) * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    for (int it = tid; it < npf * 64; it += 256) {             // sum the four strip rows in wave order; lane (lr = position, lg -> channels 4 lg ..)
      const int ln = it & 63, pf = it >> 6;
      float4 sm = *(const float4*)(sRed + ((0 * npf + pf) * 64 + ln) * 4);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const float4 v = *(const float4*)(sRed + ((q * npf + pf) * 64 + ln) * 4);
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
      const int pos = 16 * pf + (ln & 15), ci = cif * 16 + (ln >> 4) * 4;
      float* p = rows ? mg.erow[blockIdx.z] + (((int64_t)b * 2 + (e & 1)) * w + pos) * Cin + ci
                      : mg.ecol[blockIdx.z] + (((int64_t)b * h + pos) * 2 + (e & 1)) * Cin + ci;
      *(float4*)p = sm;
    }
  };
  if (DEPTH == 1) {
    for (int b = b_lo; b < b_hi; ++b) image(b, rsA);
  } else {
    for (int b = b_lo; b < b_hi; b += 2) {
      image(b, rsA);
      if (b + 1 < b_hi) image(b + 1, rsB);
    }
  }
}

// corner (cr, cc) = (bottom?, right?): dx[row edge][column edge][ci] += sum_{qr, qs, co} Wc[corner][qr][qs][ci][co] dy[row qr from the edge][column qs from the edge][co]
// -- a [16 channels] x [16 images] GEMM tile per wave with K = (qr, qs, co): weights and dY pixels straight from memory (both 16-B pieces), added into the row terms.
// (The first version -- one thread per (corner, channel), every workgroup re-reading the 0.5 MB of corner weights -- took 113 us per launch.)
template <typename T, int NGRP>
__global__ __launch_bounds__(256) void polyd_corner_kernel(const PolydEdgeMulti mg, int B, int h, int w, int Cin, int gdy, int K) {
  constexpr int EPP = ElemTraits<T>::EPP, CPG = FixMma<T>::CPG, COP = NGRP * CPG;
  const T* __restrict__ dy = (const T*)mg.dy[blockIdx.z];
  const T* __restrict__ wc = (const T*)mg.wcorner[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int ncif = Cin >> 4, c = (int)blockIdx.x / ncif, cif = (int)blockIdx.x % ncif, cr = c >> 1, cc = c & 1;
  const int H2 = 2 * h, W2 = 2 * w, pad = (K - 1) / 2;
  const int nqr = cr ? K - pad : pad + 1, nqs = cc ? K - pad : pad + 1;
  const int b = ((int)blockIdx.y * 4 + wave) * 16 + lr;                        // this lane's image (B operand column)
  const bool bok = b < B;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  // one strip row (up to four corner pixels x NGRP operand pairs) per round: every load of the round is issued before its first MFMA.  (As a plain (qr, qs) loop
  // with run-time bounds each pair was load -> wait -> multiply: up to 16 x NGRP dependent round trips, 20-28 us per launch at 2 x 64 images for 0.03 GFLOP.)
  // The (qr, qs, gq) order of the multiplies -- the accumulation order -- is unchanged.
  for (int qr = 0; qr < nqr; ++qr) {
    uint4 av[4][NGRP], bv[4][NGRP];
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
      const bool on = qs < nqs;                                                 // (block-uniform)
      const T* wp = wc + (((int64_t)((c * 4 + qr) * 4 + (on ? qs : 0))) * Cin + cif * 16 + lr) * COP + lg * EPP;
      const T* dp = dy + (((int64_t)(bok ? b : 0) * H2 + (cr ? H2 - 1 - qr : qr)) * W2 + (cc ? W2 - 1 - (on ? qs : 0) : (on ? qs : 0))) * gdy + lg * EPP;
#pragma unroll
      for (int gq = 0; gq < NGRP; ++gq) {
        av[qs][gq] = on ? *(const uint4*)(wp + gq * CPG) : make_uint4(0, 0, 0, 0);
        bv[qs][gq] = (on && bok && gq * CPG + lg * EPP < gdy) ? *(const uint4*)(dp + gq * CPG) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
      if (qs >= nqs) break;
#pragma unroll
      for (int gq = 0; gq < NGRP; ++gq) FixMma<T>::run(av[qs][gq], bv[qs][gq], acc);        // D rows = channels 4 lg .., columns = images
    }
  }
  mfma_result_fence(acc);                                        // (fix_mma.hip.h: at -O1 the accumulator was read one instruction behind the loop's last v_mfma)
  if (!bok) return;
  float4* p = (float4*)(mg.erow[blockIdx.z] + (((int64_t)b * 2 + cr) * w + (cc ? w - 1 : 0)) * Cin + cif * 16 + lg * 4);
  float4 v = *p;
  v.x += acc[0]; v.y += acc[1]; v.z += acc[2]; v.w += acc[3];
  *p = v;
}

template <typename T, int NT, int NGRP, int NLI>
static int launch_edge_n(const PolydEdgeMulti& m, int n, int B, int h, int w, int Cin, int gdy, int K, size_t lds, hipStream_t st) {
  int groups = (B + 7) / 8;                                 // ~8 images per workgroup
  if (groups < 1) groups = 1;
  // strips in flight per thread: SV_POLYD_DEPTH (1: one image ahead, the default; 2: two register sets).  Measured and NOT kept (profiles/r06_polyd_depth.txt): depth 2 costs
  // 20 VGPRs (152 -> 172: two resident workgroups per CU instead of three) and the 512-image fp32 step goes 9.13 -> 9.29 ms, the 64-image one 1.690 -> 1.697
  static const int depth = getenv("SV_POLYD_DEPTH") ? atoi(getenv("SV_POLYD_DEPTH")) : 1;
  if (depth >= 2) {
    sv_ensure_dynamic_lds((const void*)polyd_edge_kernel<T, NT, NGRP, NLI, 2>, lds);
    hipLaunchKernelGGL((polyd_edge_kernel<T, NT, NGRP, NLI, 2>), dim3(4 * (Cin >> 4), groups, n), dim3(256), lds, st, m, B, h, w, Cin, gdy, K);
  } else {
    sv_ensure_dynamic_lds((const void*)polyd_edge_kernel<T, NT, NGRP, NLI, 1>, lds);
    hipLaunchKernelGGL((polyd_edge_kernel<T, NT, NGRP, NLI, 1>), dim3(4 * (Cin >> 4), groups, n), dim3(256), lds, st, m, B, h, w, Cin, gdy, K);
  }
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL((polyd_corner_kernel<T, NGRP>), dim3(4 * (Cin >> 4), (B + 63) / 64, n), dim3(256), 0, st, m, B, h, w, Cin, gdy, K);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
template <typename T, int NT, int NGRP>
static int launch_edge(const PolydEdgeMulti& m, int n, int B, int h, int w, int Cin, int gdy, int K, hipStream_t st) {
  constexpr int NPC = NGRP * FixMma<T>::CPG / ElemTraits<T>::EPP;
  const int big = h > w ? h : w, SW = 2 * big + NT - 1, npf = big >> 4;
  const size_t lds = (size_t)4 * SW * (NGRP * FixMma<T>::CPG * sizeof(T) + 16) + (size_t)4 * (npf < 1 ? 1 : npf) * 256 * 4;
  if (lds > 150 * 1024 || (h & 15) || (w & 15)) return SV_E_UNSUPPORTED;
  const int items = 4 * SW * NPC;                          // strip pieces per image: 16-pixel low-res lines 1280 (d4) / 32-pixel lines 1152 (the head), 2304 (d4 at 128 x 128)
  if (items <= 5 * 256) return launch_edge_n<T, NT, NGRP, 5>(m, n, B, h, w, Cin, gdy, K, lds, st);
  if (items <= 9 * 256) return launch_edge_n<T, NT, NGRP, 9>(m, n, B, h, w, Cin, gdy, K, lds, st);
  return SV_E_UNSUPPORTED;
}

}  // namespace

// n <= 2 twin layers: edge + corner terms into edgews[i] (svg_polyd_ws_bytes), then the main stride-2 conv whose epilogue adds them and applies the ReLU
// gate of the low-res activation mask_lo[i] (may be null).  w_polyd[i]: the main / edge / corner images (svg_prep_job_polyd 0, 1, 2 back to back).
int svk_polyd_dgrad_multi(const sv_conv_desc* d, int n, const void* const* dy, const void* const* w_polyd, const void* const* mask_lo, void* const* dx_lo,
                          void* const* edgews, hipStream_t st) {
  if (!svg_polyd(d) || n < 1 || n > 2) return SV_E_UNSUPPORTED;
  const size_t esz = d->dtype == SV_BF16 ? 2 : 4;
  const int h = d->H / 2, w = d->W / 2, cin = svg_cin_pad(d), gdy = svg_gdy(d), cop = svg_polyd_cop(d), K = d->KH;
  const int64_t o1 = svg_polyd_elems(d, 0), o2 = o1 + svg_polyd_elems(d, 1);
  PolydEdgeMulti m;
  TapGemmArgs a[2];
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    m.dy[i] = dy[k];
    m.wedge[i] = (const char*)w_polyd[k] + o1 * esz; m.wcorner[i] = (const char*)w_polyd[k] + o2 * esz;
    m.erow[i] = (float*)edgews[k]; m.ecol[i] = m.erow[i] + (int64_t)d->B * 2 * w * cin;
  }
  int rc = SV_E_UNSUPPORTED;
  if (d->dtype == SV_F32 && K == 6 && cop == 32) rc = launch_edge<float, 9, 2>(m, n, d->B, h, w, cin, gdy, K, st);
  else if (d->dtype == SV_F32 && K == 6 && cop == 16) rc = launch_edge<float, 9, 1>(m, n, d->B, h, w, cin, gdy, K, st);
  else if (d->dtype == SV_BF16 && K == 6 && cop == 32) rc = launch_edge<bf16_t, 9, 1>(m, n, d->B, h, w, cin, gdy, K, st);      // (32 dY channels = one bf16 MFMA group: d4 exactly, the head padded)
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    svg_polyd_args(d, &a[i]);
    a[i].A = dy[i]; a[i].Wt = w_polyd[i]; a[i].out = dx_lo[i]; a[i].mask = mask_lo ? mask_lo[i] : nullptr;
    a[i].fix = m.erow[i]; a[i].fix2 = m.ecol[i];
  }
  return svk_conv_dispatch_multi(a, n, d->dtype, svg_pick_cfg(d->Cin), st);
}
