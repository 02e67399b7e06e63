// Main term of the POLYPHASE weight gradient of the decoder head (d5: UpSampling2D(bilinear) -> Conv2D(6, 6x6), vae/model.py:156,:167; its
// Conv2DBackpropFilter in vae/trainer.py:137's tape.gradient), "rolling window" form (bf16, MFMA 16x16x32), round 4:
//
//   dW'[t = (tx+2)*5 + (ty+2)][ci][col = (py*2+px)*8 + co] = sum over (image, i, j) of  x~[i + ty, j + tx, ci] * dy[2i + py, 2j + px, co]
//
// on the LOW-RES 32 x 32 grid: x~ = the edge-clamped low-res input (32 channels), dy read as its space-to-depth view (4 parities x 8 channels =
// 32 columns); conv_api.hip: svg_poly_wgrad_args, poly_wgrad.hip for the frame / projection terms, tests/test_polyphase_math.py for the algebra.
//
// wgrad_tile.hip (id 8) stages 2-D tiles and runs staging, MFMA loop and flush one after the other (94 us of the layer's 134 at 2 x 512 images,
// 768 slabs of 102 KB).  Here, as in wgrad_roll.hip, a workgroup marches down 16-pixel-wide column strips two low-res rows (= one 32-pixel MFMA
// K chunk) per step with every stage inside the MFMA waves -- and there is NO blend: the rows go global -> LDS by LDS-DMA with the edge clamp in
// the per-lane source address, and the A operands are transposed reads of the DMA image as it lies.
//   * K <-> pixels: k = 8g + 4h + q <-> row h of the pair, pixel 4g + q.  An A operand is the register pair {window row r, row r + 1} at x shift tx;
//     tap ty of the chunk (rows 2s, 2s + 1) uses the input rows 2s + ty, 2s + ty + 1.
//   * a TEAM of four waves = (input-channel fragment cf, column fragment jf) holds the whole 25-tap x 32 x 32 gradient between them: 25 accumulator
//     fragments (100 registers) per wave.  A workgroup is TWO teams (8 waves, two per SIMD) on different strips; they add up through LDS at the
//     end and write ONE slab per workgroup (115 KB; 256 slabs per launch).
//   * a wave keeps the WINDOW of the 6 input rows of its chunk x 5 x shifts in registers -- each window column is one 16-register vector (three
//     rotating row-pair slots + a mirror of the first row behind the last, so every operand is four consecutive registers: no copies; the scheme
//     wgrad_roll.hip got in round 4) -- and reads only the two NEW rows per step: 10 transposed reads + 2 for its dY fragment per 25 MFMAs.
//   * DMA: each wave moves its share of the row pair / dY chunk of step t + AHEAD (six 1-KB transfers per team and step) as inline assembly with
//     counted s_waitcnt vmcnt; one barrier per step.
// A strip takes 16 chunks + 2 lead-in steps that only fill the window (rows -2, -1 clamp to row 0).
// The accumulators leave in the fragment order of wgrad_reduce <TPW 7, CIF 2, COF 2> (svk_wgrad_reduce_all sums the workgroups' slabs in a fixed
// order: deterministic; assign mode: dW' is written); the bias gradient' (column sums of dy) is an all-ones MFMA tap.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.hip.h"
#include "kernels.h"

namespace {

struct P5Args {
  const bf16_t* A;        // low-res input [B][h][w][lda]
  const bf16_t* dY;       // hi-res gradient [B][2h][2w][8]
  float* slab;            // [gridDim.x][4 virtual waves][28 fragments][4][64]
  float* bslab;           // [gridDim.x][128] or null
  int B, h, w, lda, nxs, nstrips, per;      // per: strips per TEAM
};
struct P5Multi { P5Args a[SV_WGRAD_MAX_MULTI]; };

__device__ __forceinline__ void p5_dma16(const void* base, uint32_t off, const char* lds) {    // base: wave-uniform; off: this lane's byte offset
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ short4_t p5_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

constexpr int TW = 5, PADL = 2, NTAP = TW * TW;       // 5 x 5 taps, offsets -2 .. 2
constexpr int XSLOT = 2048;                           // one staged input row: 32 pixels x 32 channels (pixel q <-> column clamp(j0 - 2 + q); 20 used)
constexpr int DSLOT = 1024;                           // one dY row of the chunk: 16 low-res pixels x 32 columns
constexpr int NDMA = 8;                               // ring depth in STEPS (a step = one row pair + one dY row pair): 48 KB per team
constexpr int AHEAD = 6;                              // steps between a DMA and its readers (>= 2 400 MFMA cycles; NDMA > AHEAD: the slot written at
                                                      // the top of step t was last read in step t + AHEAD - NDMA < t, behind a barrier)
constexpr int LEAD = 2;                               // lead-in steps of a strip (row pairs -1, 0 only fill the window)
constexpr int TPWr = 7, NFRr = TPWr * 2 * 2;          // the reduce's fragment order: 4 virtual waves x 7 taps x (2 ci fragments x 2 column fragments)

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void wgrad_p5_kernel(const P5Multi mg) {
  const P5Args g = mg.a[blockIdx.z];                   // by value: every field lives in SGPRs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, wt = wave & 3, cf = wt & 1, jf = wt >> 1;
  char* sX = smem + team * (NDMA * 2 * XSLOT + NDMA * 2 * DSLOT);          // [NDMA][2 rows][XSLOT]
  char* sD = sX + NDMA * 2 * XSLOT;                                        // [NDMA][2 rows][DSLOT]
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int pxl = 4 * lg + lq;                                             // pixel of this lane within a 16-pixel row (the transposed read's order)
  const int x_lane = (pxl + PADL) * 64 + cf * 32 + lp * 8;                 // + slot + row * XSLOT + tx * 64   (tx = -2 .. 2)
  const int d_lane = pxl * 64 + jf * 32 + lp * 8;                          // + slot + row * DSLOT
  const short8_t ones = (short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
  const int H2 = 2 * g.h, W2 = 2 * g.w;

  // this team's strips [q_lo, q_hi) and steps; a team without strips still walks the steps of the longest team (barriers) without MFMAs
  const int tq = ((int)blockIdx.x * 2 + team) * g.per;
  const int q_lo = min(tq, g.nstrips), q_hi = min(tq + g.per, g.nstrips);
  const int SPS = g.h / 2 + LEAD;                                          // steps per strip
  const int T = g.per * SPS;                                               // steps of every team of the launch (idle ones included)
  const bool any = q_hi > q_lo;

  f32x4 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bacc = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- producers: every wave issues its share of step u's transfers: waves 0 / 1 of the team input row 2 ps - 2 / 2 ps - 1 of the strip (both
  // 16-pixel halves; ps = the step inside the strip), waves 2 / 3 the dY rows 2s / 2s + 1 of the chunk s = ps - LEAD the same step multiplies
  int pq = q_lo, ps = 0;                                                    // strip and step-in-strip of the next transfer
  auto produce = [&](int u) {
    const int q = min(pq, max(q_hi - 1, 0));
    const int b = q / g.nxs, j0 = (q - b * g.nxs) * 16;
    char* xs = sX + (u & (NDMA - 1)) * 2 * XSLOT;
    char* ds = sD + (u & (NDMA - 1)) * 2 * DSLOT;
    if (wt < 2) {
      const int r = min(max(2 * (ps - 1) + wt, 0), g.h - 1);                // step ps of a strip brings rows 2 ps - 2, 2 ps - 1 (clamped)
      const bf16_t* rowb = g.A + ((int64_t)b * g.h + r) * g.w * g.lda;      // wave-uniform
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int col = min(max(j0 - PADL + hf * 16 + (lane >> 2), 0), g.w - 1);
        p5_dma16(rowb, (uint32_t)(col * g.lda + (lane & 3) * 8) * 2u, xs + wt * XSLOT + hf * 1024);
      }
    } else {
      // dY row i of the chunk of step ps (chunk s = ps - LEAD uses it; lead-in steps fetch rows of chunks that do not exist: clamped, unused)
      const int i = min(max(2 * (ps - LEAD) + (wt - 2), 0), g.h - 1);
      const int py = (lane >> 1) & 1, pxp = lane & 1, p = lane >> 2;
      const bf16_t* src = g.dY + ((int64_t)b * H2 + 2 * i) * W2 * 8;        // wave-uniform: hi-res row 2 i of the image
      p5_dma16(src, (uint32_t)((py * W2 + 2 * (j0 + p) + pxp) * 8) * 2u, ds + (wt - 2) * DSLOT);
    }
    if (++ps == SPS) { ps = 0; ++pq; }
  };
  // before the barrier that ends step t: everything but the newest AHEAD - 1 steps has landed (waves 0 / 1 issue two transfers per step, 2 / 3 one)
  auto produce_wait = [&]() {
    if (wt < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AHEAD - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD - 1) : "memory");
  };
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  for (int u = 0; u < AHEAD; ++u) produce(u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();

  // The window: three row-PAIR slots per x shift (slot of pair P = P mod 3; elements 4 slot + 2 row .. + 1), element pair 12, 13 = the mirror
  // of slot 0's first row.  Step t reads pair t of the walk into slot t % 3; chunk t - 2 multiplies: tap rows ty = -2 .. 2 are the consecutive
  // register rows (2 (t - 2) + ty + 2, + 1) of the rotated window.
  i32x16 win[TW];
#pragma unroll
  for (int x = 0; x < TW; ++x)
#pragma unroll
    for (int e = 0; e < 16; ++e) win[x][e] = 0;

  using std::integral_constant;
  int ms = 0, mq = q_lo;                                                    // step-in-strip / strip of step t
  auto body = [&](int t, auto PHc) {
    constexpr int PH = decltype(PHc)::value;                               // t % 3: the slot that receives this step's row pair
    produce(t + AHEAD);
    {
      const char* rp = sX + (t & (NDMA - 1)) * 2 * XSLOT + x_lane;
#pragma unroll
      for (int x = 0; x < TW; ++x) {
        const int2 r0 = __builtin_bit_cast(int2, p5_tr16(rp + (x - PADL) * 64));
        const int2 r1 = __builtin_bit_cast(int2, p5_tr16(rp + XSLOT + (x - PADL) * 64));
        win[x][4 * PH] = r0.x; win[x][4 * PH + 1] = r0.y; win[x][4 * PH + 2] = r1.x; win[x][4 * PH + 3] = r1.y;
        if constexpr (PH == 0) { win[x][12] = r0.x; win[x][13] = r0.y; }
      }
    }
    const bool mm = ms >= LEAD && mq < q_hi;                                // wave-uniform: this step multiplies (not a lead-in step, not an idle team)
    if (mm) {
      const char* sd = sD + (t & (NDMA - 1)) * 2 * DSLOT + d_lane;
      const short4_t d0 = p5_tr16(sd), d1 = p5_tr16(sd + DSLOT);
      const short8_t bfr = (short8_t){d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
      // chunk rows (2s, 2s + 1), s = t' - 2 in pair units: tap ty's first row is register row 2 (t - 2) + (ty + 2) + ... of the walk = window row
      // rho = ty + 2 counted from the OLDEST pair (t - 2), whose slot is (PH + 1) % 3
      auto do_ty = [&](auto TYc) {
        constexpr int ty = decltype(TYc)::value;
        // window row rho = ty (counted from the oldest pair, whose slot is so = (PH + 1) % 3) is register row (2 so + ty) % 6; the pair (5, 0) = (5, mirror)
        constexpr int rr = (2 * ((PH + 1) % 3) + ty) % 6;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < TW; ++x) {
          const i32x4 a4 = __builtin_shufflevector(win[x], win[x], 2 * rr, 2 * rr + 1, 2 * rr + 2, 2 * rr + 3);
          acc[x * TW + ty] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, bfr), acc[x * TW + ty], 0, 0, 0);
        }
      };
      do_ty(integral_constant<int, 0>{}); do_ty(integral_constant<int, 1>{}); do_ty(integral_constant<int, 2>{});
      do_ty(integral_constant<int, 3>{}); do_ty(integral_constant<int, 4>{});
      if (g.bslab && cf == 0) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bfr), bacc, 0, 0, 0);
    }
    if (++ms == SPS) { ms = 0; ++mq; }
    produce_wait();
    barrier();
  };
  int t = 0;
  for (; t + 3 <= T; t += 3) {
    body(t, integral_constant<int, 0>{});
    body(t + 1, integral_constant<int, 1>{});
    body(t + 2, integral_constant<int, 2>{});
  }
  // (T = per * SPS and SPS = 18: a multiple of 3 -- checked by the host)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the surplus transfers
  __syncthreads();

  // ---- the two teams add up through LDS (the rings are dead), then ONE slab in the fragment order of wgrad_reduce <TPW 7, CIF 2, COF 2>:
  // virtual wave v = tap / 7, fragment f = ((tap % 7) * 2 + ci-fragment) * 2 + column-fragment;  tap = (tx + 2) * 5 + (ty + 2)
  float* sSum = (float*)smem;                                              // [4 waves][25][256]
  if (team == 1) {
#pragma unroll
    for (int k = 0; k < NTAP; ++k)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) sSum[(wt * NTAP + k) * 256 + r4 * 64 + lane] = acc[k][r4];
    if (cf == 0) sSum[4 * NTAP * 256 + jf * 64 + lane] = bacc[0];
  }
  __syncthreads();
  if (team == 1) return;
  float* sl = g.slab + (int64_t)blockIdx.x * (4 * NFRr * 256) + lane;
#pragma unroll
  for (int k = 0; k < NTAP; ++k) {
    const int tap = k;                                                      // acc index x * 5 + ty IS the tap index (tx-major)
    float* p = sl + ((tap / TPWr) * NFRr + ((tap % TPWr) * 2 + cf) * 2 + jf) * 256;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) p[r4 * 64] = acc[k][r4] + sSum[(wt * NTAP + k) * 256 + r4 * 64 + lane];
  }
  if (g.bslab && cf == 0 && lane < 16) g.bslab[(int64_t)blockIdx.x * 128 + jf * 16 + lane] = bacc[0] + sSum[4 * NTAP * 256 + jf * 64 + lane];
}

constexpr int P5_LDS = 2 * (NDMA * 2 * XSLOT + NDMA * 2 * DSLOT) > (4 * NTAP * 256 + 128) * 4 ? 2 * (NDMA * 2 * XSLOT + NDMA * 2 * DSLOT) : (4 * NTAP * 256 + 128) * 4;

}  // namespace

static int p5_wgs(int n, int nstrips, int* per) {
  int X = 256 / n;                                     // one 8-wave workgroup per CU over the launch
  int p = (nstrips + 2 * X - 1) / (2 * X);             // strips per team
  if (p < 1) p = 1;
  X = (nstrips + 2 * p - 1) / (2 * p);                 // no workgroup without work
  *per = p;
  return X < 1 ? 1 : X;
}

bool svk_wgrad_p5_supported(const WgradArgs* wv, int n) {
  static const bool off = getenv("SV_NO_WGRAD_P5") != nullptr;
  static const int min_images = getenv("SV_WGRAD_P5_MIN") ? atoi(getenv("SV_WGRAD_P5_MIN")) : 128;
  if (off || n < 1 || n > SV_WGRAD_MAX_MULTI) return false;
  const WgradArgs& w = wv[0];
  if (!w.dy_s2d || !w.clampin || !w.assign || w.ups || w.fold_kw || w.S != 1 || w.SX != 1 || w.ntaps != NTAP) return false;
  if (w.Cin_pad != 32 || w.Cin_real != 32 || w.N != 32 || w.ldy != 32 || w.lda != 32) return false;
  if (w.OY != w.OX || w.OY < 16 || (w.OY & 15) || w.lOY < 0 || ((w.OY / 2 + LEAD) % 3)) return false;      // steps per strip: a multiple of the window's rotation period
  for (int t = 0; t < NTAP; ++t)
    if (w.dx[t] != t / TW - PADL || w.dy[t] != t % TW - PADL) return false;
  const int B = w.M >> (2 * w.lOY);
  if (B * n < min_images) return false;
  int per;
  const int X = p5_wgs(n, B * (w.OX / 16), &per);
  const int64_t need = (int64_t)X * 4 * NFRr * 256 * 4 + (int64_t)X * 128 * 4;
  for (int i = 0; i < n; ++i)
    if (!wv[i].ws || wv[i].ws_bytes < need) return false;
  return true;
}

int svk_wgrad_p5_multi(const WgradArgs* wv, int n, hipStream_t st) {
  if (!svk_wgrad_p5_supported(wv, n)) return SV_E_UNSUPPORTED;
  const WgradArgs& w = wv[0];
  const int B = w.M >> (2 * w.lOY);
  const int nxs = w.OX / 16, nstrips = B * nxs;
  int per;
  const int X = p5_wgs(n, nstrips, &per);
  P5Multi m;
  WgradReduceDesc rd[SV_WGRAD_MAX_MULTI];
  for (int i = 0; i < n; ++i) {
    P5Args& a = m.a[i];
    a.A = (const bf16_t*)wv[i].A; a.dY = (const bf16_t*)wv[i].dY;
    a.slab = wv[i].ws;
    a.bslab = wv[i].dbias ? wv[i].ws + (int64_t)X * 4 * NFRr * 256 : nullptr;
    a.B = B; a.h = w.OY; a.w = w.OX; a.lda = w.lda; a.nxs = nxs; a.nstrips = nstrips; a.per = per;
    rd[i] = WgradReduceDesc{a.slab, wv[i].dW, a.bslab, wv[i].dbias, X, 1, 1, 32, 32, 32, NTAP, 0, 0, 0, 1, TPWr, 2, 2};
  }
  sv_ensure_dynamic_lds((const void*)wgrad_p5_kernel, P5_LDS);
  hipLaunchKernelGGL(wgrad_p5_kernel, dim3(X, 1, n), dim3(512), P5_LDS, st, m);
  SV_LAUNCH_CHECK();
  if (w.ev_mid[0]) { (void)hipEventRecord(w.ev_mid[0], st); (void)hipEventRecord(w.ev_mid[1], st); }
  if (w.defer && w.n_defer && *w.n_defer + n <= 64) {
    for (int i = 0; i < n; ++i) w.defer[(*w.n_defer)++] = rd[i];
    return SV_OK;
  }
  return svk_wgrad_reduce_all(rd, n, st);
}
