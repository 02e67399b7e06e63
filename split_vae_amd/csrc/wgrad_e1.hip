// Weight gradient of the first encoder layer e1 = Conv2D(32, 6, strides=2) on the 8-channel padded RGB input (vae/model.py:34; its
// Conv2DBackpropFilter + BiasAddGrad in vae/trainer.py:137's tape.gradient), bf16, MFMA 16x16x32:
//
//   dW[ky][kx][c][co] = sum over (image, oy, ox) of  x[2 oy + ky - 2, 2 ox + kx - 2, c] * dY[oy, ox, co]
//
// The whole gradient is 36 taps x 8 channels x 32 columns = 36 accumulator fragments: it fits ONE wave's registers.  The tile kernel
// (wgrad_tile.hip, id 6) still treats the layer like the wide ones -- 2-D tiles, workgroup barriers between staging and MFMAs, one slab
// per workgroup -- and takes 75 + 14 us for work whose MFMA time is 9 us and whose HBM time (134 MB) is 27 us, alone at the tail of the
// backward pass.  Here every WAVE is its own pipeline, no barrier in the loop:
//   * a wave takes a 16-pixel-wide column strip of one image's 32 x 32 dY grid and marches down it, two dY rows (one 32-pixel K chunk)
//     per step;
//   * the stride-2 conv is a stride-1 3 x 3 conv over the space-to-depth view of the input: an s2d pixel = 2 x 2 input pixels x 8
//     channels; the 16 rows of an A fragment are (px, c) of sub-row py at x shift tx, i.e. taps (2 ty + py, 2 tx + px) -- the `pairx` row
//     meaning of the tile kernel's reduce.  In LDS an s2d row is just its two INPUT rows, linear (36 pixels x 16 B each): one
//     global_load_lds per input row, 32-B "pixel" records for the transposed reads (conflict-free);
//   * a register WINDOW of the 4 s2d rows of the chunk x 3 x shifts x 2 sub-rows (48 VGPRs), shifted by two rows per step: 12 transposed
//     reads + 4 for dY per 36 + 2 MFMAs;
//   * DMA two steps ahead into a 3-slot ring private to the wave, counted s_waitcnt vmcnt (inline assembly, as in wgrad_roll.hip);
//   * at the end the 8 waves of a workgroup add their accumulators through LDS and the workgroup writes ONE slab in the fragment order of
//     wgrad_reduce <TPW 5, CIF 1, COF 2, pairx> (9 MB per launch instead of the tile kernel's 31); svk_wgrad_reduce_all sums the
//     workgroups in a fixed order.  The bias gradient is an all-ones MFMA tap.
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"

namespace {

struct E1Args {
  const bf16_t* A;        // input [B][64][64][8]
  const bf16_t* dY;       // [B][32][32][32]
  float* slab;            // [gridDim.x][4 virtual waves][10 fragments][4][64]
  float* bslab;           // [gridDim.x][128] or null
  int B, ntasks;          // ntasks = 2 B (image, strip)
};
struct E1Multi { E1Args a[SV_WGRAD_MAX_MULTI]; };

__device__ __forceinline__ void dma16(const void* base, uint32_t off, const char* lds) {     // see wgrad_roll.hip
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ short4_t tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

constexpr int IH = 64, IW = 64, OH = 32, OW = 32;
constexpr int ROWB = 608;                     // an input row of the strip: 36 pixels x 16 B (+ 32)
constexpr int IN_SLOT = 4 * ROWB;             // the four input rows of a step (s2d rows 2s + 1, 2s + 2, sub-rows py = 0, 1)
constexpr int DY_SLOT = 2048;                 // 2 rows x 16 pixels x 32 channels
constexpr int SLOT = IN_SLOT + DY_SLOT;
constexpr int NSL = 3;
constexpr int WAVE_LDS = NSL * SLOT + ROWB;   // + a dump row for the transfers of rows outside the image
constexpr int HS = OH / 2, SPS = HS + 1;      // steps per strip: one lead-in (s2d rows -1, 0)
constexpr int NDMA = 6;                       // transfers per step and wave

__global__ __launch_bounds__(512, 1) void wgrad_e1_kernel(const E1Multi mg) {
  const E1Args g = mg.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3, pxl = 4 * lg + lq;
  char* my = smem + wave * WAVE_LDS;
  for (int q = lane; q < WAVE_LDS / 16; q += 64) *(uint4*)(my + q * 16) = make_uint4(0, 0, 0, 0);   // the x halo outside the image stays zero
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  f32x4 acc[9][2][2];                         // [ty * 3 + tx][py][j]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[2];
  bacc[0] = bacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const short8_t ones = (short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  short4_t win[4][3][2];                      // [s2d row 2c-1 .. 2c+2][tx][py]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) win[r][tx][0] = win[r][tx][1] = (short4_t){0, 0, 0, 0};

  // this wave's tasks: (image, strip) = task / 2, task & 1
  const int wslot = (int)blockIdx.x * 8 + wave, wstride = (int)gridDim.x * 8;
  const int ntask_w = wslot < g.ntasks ? (g.ntasks - wslot + wstride - 1) / wstride : 0;
  const int T = ntask_w * SPS;

  // producer: the transfers of step u = (task pk, s = ps); past the last step the last one again (the in-flight count stays constant)
  int pk = 0, ps = -1;
  auto produce = [&](int u) {
    const int task = wslot + pk * wstride, b = task >> 1, x0 = (task & 1) * 16;
    char* slot = my + (u % NSL) * SLOT;
    const int col = 2 * x0 - 2 + lane;                           // input column of this lane's 16-B pixel
    const bool lane_on = lane < 36 && (unsigned)col < (unsigned)IW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 4 * ps + 2 + k;                              // input rows 4s+2 .. 4s+5 = s2d rows 2s+1, 2s+2 x sub-rows
      const bool in = (unsigned)r < (unsigned)IH;
      const bf16_t* rowb = g.A + ((int64_t)b * IH + (in ? r : 0)) * IW * 8;
      char* dst = in ? slot + k * ROWB : my + NSL * SLOT;        // rows outside the image: the transfer goes to the dump row ...
      if (!in && lane < 36) *(uint4*)(slot + k * ROWB + lane * 16) = make_uint4(0, 0, 0, 0);     // ... and the slot gets zeros
      if (lane_on) dma16(rowb, (uint32_t)col * 16u, dst);       // ONE instruction per row whatever the mask (never empty): produce_wait counts them
    }
    const int y = 2 * max(ps, 0);
    const bf16_t* src = g.dY + (((int64_t)b * OH + y) * OW + x0) * 32;
#pragma unroll
    for (int h = 0; h < 2; ++h) dma16(src + (int64_t)h * OW * 32, (uint32_t)((lane >> 2) * 32 + (lane & 3) * 8) * 2u, slot + IN_SLOT + h * 1024);
    if (pk < ntask_w - 1 || ps < HS - 1) { if (++ps == HS) { ps = -1; ++pk; } }
  };

  if (T > 0) {
    produce(0);
    produce(1);
    int ms = -1;
    for (int u = 0; u < T; ++u) {
      produce(u + 2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");      // everything but the two newest steps has landed: step u
      const char* slot = my + (u % NSL) * SLOT;
      // window <- s2d rows 2s+1, 2s+2 of this step
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          win[0][tx][py] = win[2][tx][py];
          win[1][tx][py] = win[3][tx][py];
          win[2][tx][py] = tr16(slot + (0 * 2 + py) * ROWB + (pxl + tx) * 32 + lp * 8);
          win[3][tx][py] = tr16(slot + (1 * 2 + py) * ROWB + (pxl + tx) * 32 + lp * 8);
        }
      if (ms >= 0) {
        const char* sd = slot + IN_SLOT + pxl * 64 + lp * 8;
        short8_t bfr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const short4_t lo = tr16(sd + j * 32), hi = tr16(sd + 1024 + j * 32);
          bfr[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const short4_t lo = win[ty][tx][py], hi = win[ty + 1][tx][py];
              const short8_t af = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[ty * 3 + tx][py][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[j]),
                                                                                  acc[ty * 3 + tx][py][j], 0, 0, 0);
            }
        if (g.bslab) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bfr[j]), bacc[j], 0, 0, 0);
        }
      }
      if (++ms == HS) ms = -1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the surplus transfers
  __syncthreads();

  // ---- the workgroup's 8 waves add up through LDS, four fragments a round; waves 0..3 each sum one and write it to the slab in the order of
  // wgrad_reduce <5, 1, 2, pairx>: tap u = (2 ty + py) * 3 + tx of the halved tap list, virtual wave u / 5, fragment (u % 5) * 2 + j
  float* xch = (float*)smem;                                     // [4 fragments][8 waves][64 lanes][4]
#pragma unroll
  for (int r = 0; r < 9; ++r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int fi = 4 * r + k, t9 = fi >> 2, py = (fi >> 1) & 1, j = fi & 1;
      *(f32x4*)(xch + ((k * 8 + wave) * 64 + lane) * 4) = acc[t9][py][j];
    }
    __syncthreads();
    if (wave < 4) {
      const int fi = 4 * r + wave, t9 = fi >> 2, py = (fi >> 1) & 1, j = fi & 1;
      const int ty = t9 / 3, tx = t9 - ty * 3, u = (2 * ty + py) * 3 + tx;
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 8; ++w) s += *(const f32x4*)(xch + ((wave * 8 + w) * 64 + lane) * 4);
      float* p = g.slab + (((int64_t)blockIdx.x * 4 + u / 5) * 10 + (u % 5) * 2 + j) * 256 + lane;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) p[r4 * 64] = s[r4];
    }
    __syncthreads();
  }
  if (g.bslab) {
    if (lane < 16) { xch[(0 * 8 + wave) * 16 + lane] = bacc[0][0]; xch[(1 * 8 + wave) * 16 + lane] = bacc[1][0]; }
    __syncthreads();
    if (tid < 32) {
      float s = 0.f;
      for (int w = 0; w < 8; ++w) s += xch[((tid >> 4) * 8 + w) * 16 + (tid & 15)];
      g.bslab[(int64_t)blockIdx.x * 128 + tid] = s;
    }
  }
}

int e1_wgs(int n, int ntasks) {               // workgroups per problem: 8 waves each, one per CU over the launch
  int X = 256 / n;
  const int cap = (ntasks + 7) / 8;
  if (X > cap) X = cap;
  return X < 1 ? 1 : X;
}

}  // namespace

bool svk_wgrad_e1_supported(const WgradArgs* wv, int n) {
  static const bool off = getenv("SV_NO_WGRAD_E1") != nullptr;
  if (off || n < 1 || n > SV_WGRAD_MAX_MULTI) return false;
  const WgradArgs& w = wv[0];
  if (w.ups || w.S != 2 || w.SX != 2 || w.ntaps != 36 || w.Cin_pad != 8 || w.lda != 8 || w.ldy != 32 || w.ycols != 32 || w.N != 32) return false;
  if (w.fold_kw || w.clampin || w.dy_s2d || w.assign) return false;
  if (w.OY != OH || w.OX != OW || w.IH != IH || w.IW != IW || w.Cin_real > 8) return false;
  for (int t = 0; t < 36; ++t)
    if (w.dy[t] != t / 6 - 2 || w.dx[t] != t % 6 - 2) return false;
  const int B = w.M / (OH * OW);
  static const int min_tasks = getenv("SV_WGRAD_E1_MIN") ? atoi(getenv("SV_WGRAD_E1_MIN")) : 512;
  if (n * 2 * B < min_tasks) return false;     // small launches: too few strips for the waves of the chip (the tile kernel cuts 2-D tiles)
  const int X = e1_wgs(n, 2 * B);
  const int64_t need = (int64_t)X * 4 * 10 * 256 * 4 + (int64_t)X * 128 * 4;
  for (int i = 0; i < n; ++i)
    if (!wv[i].ws || wv[i].ws_bytes < need) return false;
  return true;
}

int svk_wgrad_e1_multi(const WgradArgs* wv, int n, hipStream_t st) {
  if (!svk_wgrad_e1_supported(wv, n)) return SV_E_UNSUPPORTED;
  const WgradArgs& w = wv[0];
  const int B = w.M / (OH * OW);
  const int X = e1_wgs(n, 2 * B);
  E1Multi m;
  WgradReduceDesc rd[SV_WGRAD_MAX_MULTI];
  for (int i = 0; i < n; ++i) {
    E1Args& a = m.a[i];
    a.A = (const bf16_t*)wv[i].A; a.dY = (const bf16_t*)wv[i].dY;
    a.slab = wv[i].ws;
    a.bslab = wv[i].dbias ? wv[i].ws + (int64_t)X * 4 * 10 * 256 : nullptr;
    a.B = B; a.ntasks = 2 * B;
    // the reduce of the tile kernel's pairx form: 18 tap entries (ky, kx / 2), rows 8..15 of a fragment = the odd kx
    rd[i] = WgradReduceDesc{a.slab, wv[i].dW, a.bslab, wv[i].dbias, X, 1, 1, 8, w.Cin_real, 32, 18, 0, 0, 1, 0, 5, 1, 2};
  }
  const size_t lds = 8 * WAVE_LDS;
  sv_ensure_dynamic_lds((const void*)wgrad_e1_kernel, lds);
  hipLaunchKernelGGL(wgrad_e1_kernel, dim3(X, 1, n), dim3(512), lds, st, m);
  SV_LAUNCH_CHECK();
  if (w.ev_mid[0]) { (void)hipEventRecord(w.ev_mid[0], st); (void)hipEventRecord(w.ev_mid[1], st); }
  if (w.defer && w.n_defer && *w.n_defer + n <= 64) {
    for (int i = 0; i < n; ++i) w.defer[(*w.n_defer)++] = rd[i];
    return SV_OK;
  }
  return svk_wgrad_reduce_all(rd, n, st);
}
