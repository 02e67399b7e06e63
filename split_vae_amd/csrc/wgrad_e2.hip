// Weight gradient of the second encoder layer e2 = Conv2D(64, 6, strides=2) on the 32-channel 32 x 32 map (vae/model.py:35; its
// Conv2DBackpropFilter + BiasAddGrad in vae/trainer.py:137's tape.gradient), bf16, MFMA 16x16x32:
//
//   dW[ky][kx][c][co] = sum over (image, oy, ox) of  x[2 oy + ky - 2, 2 ox + kx - 2, c] * dY[oy, ox, co]        (dY: 16 x 16 x 64 per image)
//
// 36 taps x 32 channels x 64 columns = 288 accumulator fragments: exactly what the 8 waves of one workgroup hold at 36 each -- the shape of
// wgrad_roll.hip (d4) without the resize.  The stride-2 conv is a stride-1 3 x 3 conv over the space-to-depth view of the input (an s2d pixel
// = 2 x 2 input pixels x 32 channels), so, as in wgrad_e1.hip, an s2d row sits in LDS as its two INPUT rows, moved by global_load_lds.
//   * one 8-wave workgroup per CU marches down whole images (the 16-pixel dY rows are one MFMA strip), two dY rows per step;
//   * wave w owns the 16-channel fragment (py, px, half) = (w >> 2, (w >> 1) & 1, w & 1) of the s2d pixel -- taps (2 ty + py, 2 tx + px),
//     channels 16 half .. -- for all nine (ty, tx) and the four column fragments: 36 accumulators;
//   * it keeps a register WINDOW of the 4 s2d rows of the chunk x 3 x shifts of ITS sub-row plane (24 VGPRs), shifted by two rows per step:
//     6 transposed reads + 8 for dY per 36 MFMAs;
//   * every wave issues two of the step's 16 transfers two steps ahead (counted s_waitcnt vmcnt, inline assembly as in wgrad_roll.hip); one
//     barrier per step.  An input row is 36 pixels x 64 B with an s2d-pixel pitch of 128 B -- 16 consecutive pixels of a transposed read
//     would share two 32-B bank groups --, so the DMA SOURCE is permuted: the 32-B unit u of s2d pixel X lands in unit u ^ ((X >> 1) & 3)
//     (the write stays linear), and the reads un-swizzle: two passes per read, the minimum for 512 B.  dY rows (128 B per pixel) likewise;
//   * slabs in the fragment order of wgrad_reduce <TPW 9, CIF 2, COF 4>, summed by svk_wgrad_reduce_all in workgroup order; the bias gradient
//     is an all-ones MFMA tap in waves 0..3.
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"

namespace {

struct E2Args {
  const bf16_t* A;        // input [B][32][32][32]
  const bf16_t* dY;       // [B][16][16][64]
  float* slab;            // [gridDim.x][4 virtual waves][72 fragments][4][64]
  float* bslab;           // [gridDim.x][128] or null
  int B;
};
struct E2Multi { E2Args a[SV_WGRAD_MAX_MULTI]; };

__device__ __forceinline__ void dma16(const void* base, uint32_t off, const char* lds) {     // see wgrad_roll.hip
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ short4_t tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

constexpr int IH = 32, IW = 32, OH = 16, OW = 16, CI = 32, CO = 64;
constexpr int ROWB = 36 * 64;                 // an input row of the strip: 36 pixels (s2d columns -1 .. 16) x 64 B
constexpr int IN_SLOT = 4 * ROWB;             // the four input rows of a step (s2d rows 2s + 1, 2s + 2 x sub-rows py)
constexpr int DY_SLOT = 2 * OW * CO * 2;      // 2 rows x 16 pixels x 128 B
constexpr int SLOT = IN_SLOT + DY_SLOT;       // 13312
constexpr int NSL = 4;
constexpr int LDS_BYTES = NSL * SLOT + 1024;  // + a dump area for the transfers of rows outside the image
constexpr int HS = OH / 2, SPS = HS + 1;      // steps per image: one lead-in (s2d rows -1, 0)

__global__ __launch_bounds__(512, 1) void wgrad_e2_kernel(const E2Multi mg) {
  const E2Args g = mg.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3, pxl = 4 * lg + lq;
  const int wpy = wave >> 2, wu = wave & 3;                      // this wave: sub-row plane, 32-B unit (px * 2 + half) of the s2d pixel
  for (int q = tid; q < LDS_BYTES / 16; q += 512) *(uint4*)(smem + q * 16) = make_uint4(0, 0, 0, 0);     // the x halo outside the image stays zero
  __syncthreads();

  f32x4 acc[9][4];                            // [ty * 3 + tx][j]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const short8_t ones = (short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  short4_t win[4][3];                         // [s2d row 2c-1 .. 2c+2][tx], this wave's plane and unit
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) win[r][tx] = (short4_t){0, 0, 0, 0};
  // byte offsets of the transposed reads inside an input row / a dY row (un-swizzled per lane)
  int aoff[3], boff[4];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) { const int p = pxl + tx; aoff[tx] = (p * 4 + (wu ^ ((p >> 1) & 3))) * 32 + lp * 8; }
#pragma unroll
  for (int j = 0; j < 4; ++j) boff[j] = (pxl * 4 + (j ^ ((pxl >> 1) & 3))) * 32 + lp * 8;

  // this workgroup's images
  const int per = (g.B + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_lo = (int)blockIdx.x * per, b_hi = min(g.B, b_lo + per);
  const int T = b_hi > b_lo ? (b_hi - b_lo) * SPS : 0;

  // producer: wave w issues transfers 2w, 2w + 1 of the 16 of step u = (image pb, s = ps): 12 = input row k (0..3) x 64-lane segment (0..2),
  // 4 = dY row h x half.  Past the last step: the last one again (the in-flight count stays constant).
  int pb = b_lo, ps = -1;
  auto produce = [&](int u) {
    char* slot = smem + (u & (NSL - 1)) * SLOT;
    char* dump = smem + NSL * SLOT;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int job = 2 * wave + jj;                             // wave-uniform
      if (job < 12) {
        const int k = job / 3, seg = job - k * 3;
        const int r = 4 * ps + 2 + k;                            // input rows 4s+2 .. 4s+5
        const bool in = (unsigned)r < (unsigned)IH;
        const int q = seg * 64 + lane;                           // 16-B position in the LDS row (144 of them)
        const int pos = q >> 1, X = pos >> 2, u32 = (pos & 3) ^ ((X >> 1) & 3);      // position -> s2d pixel X, source unit
        const int col = 2 * X + (u32 >> 1) - 2;                  // input column
        const bool lane_on = q < 144 && (unsigned)col < (unsigned)IW;
        const bf16_t* rowb = g.A + ((int64_t)pb * IH + (in ? r : 0)) * IW * CI;
        const uint32_t off = (uint32_t)(col * 64 + (u32 & 1) * 32 + (q & 1) * 16);
        if (!in && q < 144) *(uint4*)(slot + k * ROWB + q * 16) = make_uint4(0, 0, 0, 0);
        if (lane_on) dma16(rowb, off, in ? slot + k * ROWB + seg * 1024 : dump);
      } else {
        const int h = (job - 12) >> 1, hf = (job - 12) & 1;
        const int y = 2 * max(ps, 0) + h;
        const int q = hf * 64 + lane, px = q >> 3, pos = (q >> 1) & 3, j = pos ^ ((px >> 1) & 3);       // position -> pixel, source unit
        const bf16_t* src = g.dY + (((int64_t)pb * OH + y) * OW) * CO;
        dma16(src, (uint32_t)(px * 128 + j * 32 + (q & 1) * 16), slot + IN_SLOT + h * 2048 + hf * 1024);
      }
    }
    if (pb < b_hi - 1 || ps < HS - 1) { if (++ps == HS) { ps = -1; ++pb; } }
  };

  if (T > 0) {
    produce(0);
    produce(1);
    int ms = -1;
    for (int u = 0; u < T; ++u) {
      produce(u + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           // this wave's transfers of step u have landed (two steps of two stay in flight) ...
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // ... and everybody else's
      asm volatile("" ::: "memory");
      const char* slot = smem + (u & (NSL - 1)) * SLOT;
      const char* r0 = slot + (0 * 2 + wpy) * ROWB;
      const char* r1 = slot + (1 * 2 + wpy) * ROWB;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        win[0][tx] = win[2][tx];
        win[1][tx] = win[3][tx];
        win[2][tx] = tr16(r0 + aoff[tx]);
        win[3][tx] = tr16(r1 + aoff[tx]);
      }
      if (ms >= 0) {
        const char* sd = slot + IN_SLOT;
        short8_t bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const short4_t lo = tr16(sd + boff[j]), hi = tr16(sd + 2048 + boff[j]);
          bfr[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const short4_t lo = win[ty][tx], hi = win[ty + 1][tx];
            const short8_t af = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[ty * 3 + tx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[j]), acc[ty * 3 + tx][j], 0, 0, 0);
          }
        if (g.bslab && wave < 4) {                               // wave j: column sums of dY fragment j
          const short8_t bs = wave == 0 ? bfr[0] : wave == 1 ? bfr[1] : wave == 2 ? bfr[2] : bfr[3];
          bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bs), bacc, 0, 0, 0);
        }
      }
      if (++ms == HS) ms = -1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the surplus transfers

  // ---- flush in the fragment order of wgrad_reduce <TPW 9, CIF 2, COF 4>: tap (2 ty + py) * 6 + 2 tx + px, virtual wave tap / 9,
  // fragment ((tap % 9) * 2 + half) * 4 + j
  const int px = wu >> 1, half = wu & 1;
  float* sl = g.slab + (int64_t)blockIdx.x * (4 * 72 * 256) + lane;
#pragma unroll
  for (int ty = 0; ty < 3; ++ty)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int tap = (2 * ty + wpy) * 6 + 2 * tx + px;           // wave-uniform
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* p = sl + ((tap / 9) * 72 + ((tap % 9) * 2 + half) * 4 + j) * 256;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) p[r4 * 64] = acc[ty * 3 + tx][j][r4];
      }
    }
  if (g.bslab && wave < 4 && lane < 16) g.bslab[(int64_t)blockIdx.x * 128 + wave * 16 + lane] = bacc[0];
}

int e2_wgs(int n, int B) {
  int X = 256 / n;
  if (X > B) X = B;
  return X < 1 ? 1 : X;
}

}  // namespace

bool svk_wgrad_e2_supported(const WgradArgs* wv, int n) {
  static const bool off = getenv("SV_NO_WGRAD_E2") != nullptr;
  if (off || n < 1 || n > SV_WGRAD_MAX_MULTI) return false;
  const WgradArgs& w = wv[0];
  if (w.ups || w.S != 2 || w.SX != 2 || w.ntaps != 36 || w.Cin_pad != CI || w.Cin_real != CI || w.lda != CI || w.ldy != CO || w.ycols != CO || w.N != CO) return false;
  if (w.fold_kw || w.clampin || w.dy_s2d || w.assign) return false;
  if (w.OY != OH || w.OX != OW || w.IH != IH || w.IW != IW) return false;
  for (int t = 0; t < 36; ++t)
    if (w.dy[t] != t / 6 - 2 || w.dx[t] != t % 6 - 2) return false;
  const int B = w.M / (OH * OW);
  static const int min_images = getenv("SV_WGRAD_E2_MIN") ? atoi(getenv("SV_WGRAD_E2_MIN")) : 512;
  if (n * B < min_images) return false;        // small launches: a workgroup per image leaves the chip idle and still writes a full slab each
  const int X = e2_wgs(n, B);
  const int64_t need = (int64_t)X * 4 * 72 * 256 * 4 + (int64_t)X * 128 * 4;
  for (int i = 0; i < n; ++i)
    if (!wv[i].ws || wv[i].ws_bytes < need) return false;
  return true;
}

int svk_wgrad_e2_multi(const WgradArgs* wv, int n, hipStream_t st) {
  if (!svk_wgrad_e2_supported(wv, n)) return SV_E_UNSUPPORTED;
  const WgradArgs& w = wv[0];
  const int B = w.M / (OH * OW);
  const int X = e2_wgs(n, B);
  E2Multi m;
  WgradReduceDesc rd[SV_WGRAD_MAX_MULTI];
  for (int i = 0; i < n; ++i) {
    E2Args& a = m.a[i];
    a.A = (const bf16_t*)wv[i].A; a.dY = (const bf16_t*)wv[i].dY;
    a.slab = wv[i].ws;
    a.bslab = wv[i].dbias ? wv[i].ws + (int64_t)X * 4 * 72 * 256 : nullptr;
    a.B = B;
    rd[i] = WgradReduceDesc{a.slab, wv[i].dW, a.bslab, wv[i].dbias, X, 1, 1, CI, CI, CO, 36, 0, 0, 0, 0, 9, 2, 4};
  }
  sv_ensure_dynamic_lds((const void*)wgrad_e2_kernel, LDS_BYTES);
  hipLaunchKernelGGL(wgrad_e2_kernel, dim3(X, 1, n), dim3(512), LDS_BYTES, st, m);
  SV_LAUNCH_CHECK();
  if (w.ev_mid[0]) { (void)hipEventRecord(w.ev_mid[0], st); (void)hipEventRecord(w.ev_mid[1], st); }
  if (w.defer && w.n_defer && *w.n_defer + n <= 64) {
    for (int i = 0; i < n; ++i) w.defer[(*w.n_defer)++] = rd[i];
    return SV_OK;
  }
  return svk_wgrad_reduce_all(rd, n, st);
}
