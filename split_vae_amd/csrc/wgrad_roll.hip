// Weight gradient of a 6x6 SAME conv over a 2x-upsampled input (decoder layer d4: vae/model.py:154,:165-166; its
// Conv2DBackpropFilter in vae/trainer.py:137's tape.gradient), "rolling window" form (bf16, MFMA 16x16x32):
//
//   dW[ky][kx][ci][co] = sum over (image, Y, X) of  U(x)[Y + ky - 2, X + kx - 2, ci] * dY[Y, X, co]      (U = tf.image.resize 2x, zero outside)
//
// wgrad_tile.hip stages a 2-D tile, reads every tap's A operand back from LDS (one transposed read pair per TWO MFMAs: 32 output
// channels are only 2 column fragments) and runs its phases -- fused-resize staging, MFMAs, flush -- one after the other, overlapping
// only across the workgroups of a CU.  Here ONE 8-wave workgroup per CU marches down 16-pixel-wide column strips of the images, two
// output rows (= one 32-pixel MFMA K chunk) per step, as a pipeline whose stages run in the same waves:
//   * K <-> pixels as in wgrad_tile.hip (k = 8g + 4h + q <-> row h of the pair, pixel 4g + q): an A operand is the register pair
//     {input row r, input row r + 1} shifted by kx pixels; tap ky of chunk c uses the hi-res rows 2c + ky - 2, 2c + ky - 1.
//   * wave w owns the 16-channel fragment w & 3 and the x taps 3 (w >> 2) .. +2, all six y taps, both column fragments: 36 accumulator
//     fragments (144 registers; the two waves of a SIMD hold the whole 36-tap x 64 x 32 gradient of the CU between them).
//   * it keeps a WINDOW of the 7 input rows of the chunk x its 3 x shifts in registers.  A step shifts the window by two rows and reads
//     only the two NEW rows from LDS: 6 transposed reads per 36 MFMAs (the tile kernel: 36); tap rows 0..3 do not touch the new rows,
//     so the reads' latency hides under 24 MFMAs.
//   * DMA stage: wave 2 / wave 3 move the raw low-res row / the dY chunk of step t + 8 into LDS rings with global_load_lds (no staging
//     registers; issued as inline assembly with counted s_waitcnt vmcnt(N), because the compiler orders every LDS read behind a
//     vmcnt(0) once a DMA builtin may alias it).
//   * blend stage: the 2x resize of the block row of step t + 2 runs ON THE MATRIX PIPE -- two MFMAs per wave against a constant weight
//     operand (see roll_loop) -- from the raw ring into the blended ring, two steps ahead of its readers: one barrier per TWO steps.
// A step brings the block row i = c + 1, i.e. the hi-res rows (2i + 1, 2i + 2), which blend the same two low-res rows.  Three lead-in
// steps per strip fill the window (rows outside the image are zeros, as SAME padding wants).
//
// Measured (MI355X, 2 x 512 images, in the step's serial table): 0.147 ms against the tile kernel's 0.170 (42 % of the bf16 MFMA peak
// against 37 %); stand-alone at 1024 images 187 against 219 us.  What the ablation builds (-DROLL_ABL, scripts/r03_roll_abl_build.sh)
// showed on the way: VALU instructions and MFMAs of the two waves of a SIMD do NOT overlap in issue -- the blend as ~36 VALU
// instructions per lane and channel pair cost 42 of 194 us, the window copies and address arithmetic most of the rest of the gap to
// the 68 us of pure MFMA time; LDS traffic (21 -> 6 reads per step) and the barrier rate (every step -> every other) each moved the
// launch by < 2 %.
//
// The accumulators leave through the partial-sum slab of wgrad_tile.hip, written in the fragment order of its <TPW 9, CIF 4, COF 2>
// reduce (svk_wgrad_reduce_all sums the workgroups' slabs in a fixed order: deterministic); the bias gradient is an all-ones MFMA tap.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.hip.h"
#include "kernels.h"

namespace {

struct RollArgs {
  const bf16_t* A;        // low-res input [B][OH/2][OW/2][lda]
  const bf16_t* dY;       // [B][OH][OW][ldy]
  float* slab;            // [gridDim.x][4 virtual waves][72 fragments][4][64]
  float* bslab;           // [gridDim.x][128] or null
  int B, OH, OW, lda, ldy, nxs, nstrips;
};
struct RollMulti { RollArgs a[SV_WGRAD_MAX_MULTI]; };

// 64 lanes x 16 B, global -> LDS (lane l lands at lds + 16 l), as inline assembly: the compiler orders every later read of an LDS object
// a DMA builtin may have written behind s_waitcnt vmcnt(0), which would park the two producer waves on their NEWEST transfer every step;
// here the only waits are the counted ones in produce_wait() (the consumers are other waves, behind the step barrier)
__device__ __forceinline__ void dma16(const void* base, uint32_t off, const char* lds) {     // base: wave-uniform (SGPR pair); off: this lane's byte offset
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ short4_t tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

// -DSV_ROLL_STAMP (diagnostic builds): every wave of workgroup 0 records the shader clock at NSTAMP points of steps STAMP_T0 .. +3 and
// the workgroup overwrites the head of its slab with them (results are then wrong; scripts/r03_roll_stamps.py prints the timeline)
// -DROLL_ABL=<mask> (diagnostic builds, wrong results): 1 no MFMAs, 2 no blend, 4 no DMA, 8 no window reads, 16 no step barrier, 32 no flush
#ifndef ROLL_ABL
#define ROLL_ABL 0
#endif
#ifdef SV_ROLL_STAMP
constexpr int NSTAMP = 8, STAMP_T0 = 40;
#define ROLL_STAMP(id) do { if (blockIdx.x == 0 && t >= STAMP_T0 && t < STAMP_T0 + 4 && lane == 0) stamps[((t - STAMP_T0) * 8 + wave) * NSTAMP + (id)] = clock64(); } while (0)
#else
#define ROLL_STAMP(id) do {} while (0)
#endif
constexpr int KS = 6, PAD = 2, NT = KS * KS;        // taps
constexpr int KXW = 3;                               // x taps per wave (waves 0-3: kx 0..2, waves 4-7: kx 3..5)
constexpr int COF = 2;                               // output-channel fragments (32 channels)
constexpr int PW = 22;                               // patch pixels per staged row (16 + 5 halo, +1)
constexpr int PS = 160;                              // bytes per blended input pixel record: 64 channels + 32 (PS / 32 odd: conflict-free transposed reads)
constexpr int IN_SLOT = 2 * PW * PS;                 // one block row = two hi-res rows
constexpr int RAW_SLOT = 2048;                       // one low-res row of the strip: 16 columns (12 used) x 64 channels, as the DMA writes it
constexpr int DY_SLOT = 2048;                        // one dY chunk: 2 rows x 16 pixels x 32 channels, as the DMA writes it
constexpr int NSLOT = 8;                             // ring depth of the blended rows (power of two; a block row is written two steps before it is
                                                     // read and the waves of a workgroup are at most one step apart: four would do)
constexpr int NDMA = 16;                             // ring depth of the two DMA rings
constexpr int AHEAD = 8;                             // steps between a DMA and the MFMAs that use it (the blend runs two steps ahead of those)
constexpr int LEAD = 3;                              // lead-in steps of a strip (block rows -2, -1, 0)

struct RollLds {
  long long* stamps;
  char* in;                                          // [PS + NSLOT * IN_SLOT] blended block rows (one pixel of front padding)
  char* raw;                                         // [NDMA * RAW_SLOT] raw low-res rows (DMA)
  char* dy;                                          // [NDMA * DY_SLOT] dY chunks (DMA)
  char* bw;                                          // [512 threads][2][16 B] the blend's per-lane weight operands (read back per step: 8 VGPRs less)
};

// One workgroup's walk.  Roles per step: every wave multiplies (its channel fragment x its three x taps x all six y taps) and blends its
// share of the block row two steps ahead (its channel fragment x one of the two 16-pixel segments of the patch); wave 2 / wave 3 also
// move the raw low-res row / the dY chunk of step t + AHEAD.
__device__ __forceinline__ void roll_loop(const RollArgs& g, const RollLds& L, f32x4 (&acc)[KS * KXW][COF], f32x4& bacc, int q_lo, int q_hi, int T) {
  using std::integral_constant;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cf = wave & 3, kx0 = (wave >> 2) * KXW;              // this wave: input-channel fragment, first x tap of its three
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int pxl = 4 * lg + lq;                                   // pixel of this lane within a 16-pixel row
  const int in_lane = (pxl + kx0) * PS + cf * 32 + lp * 8;       // + slot + (row * PW + kx) * PS
  const int dy_lane = pxl * 64 + lp * 8;                         // + slot + row * 1024 + j * 32
  const int LH = g.OH >> 1, LW = g.OW >> 1, HS = g.OH >> 1;
  const short8_t ones = (short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
  long long* stamps = L.stamps;
  (void)stamps;

  // ---- producers: wave 2 moves the raw low-res row of step u (row clamp(i + 1), i = s + 1), wave 3 its dY chunk, AHEAD steps early.
  // (image, x0, s) of the next step to fetch advance without divisions and stay on the last step when the strips run out (the surplus
  // transfers keep the in-flight count, which produce_wait() relies on, constant)
  const bool producer = wave == 2 || wave == 3;
  int pq = q_lo, pb = q_lo / g.nxs, px0 = (q_lo - pb * g.nxs) * 16, ps = -LEAD;
  auto produce = [&](int u) {
    if (ROLL_ABL & 4) return;
    if (wave == 2) {
      const int r = min(max(ps + 2, 0), LH - 1), jlo = (px0 >> 1) - 2;
      const bf16_t* rowb = g.A + ((int64_t)pb * LH + r) * LW * g.lda;             // wave-uniform
      const char* dst = L.raw + (u & (NDMA - 1)) * RAW_SLOT;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = min(max(jlo + h * 8 + (lane >> 3), 0), LW - 1);
        dma16(rowb, (uint32_t)(col * g.lda + (lane & 7) * 8) * 2u, dst + h * 1024);
      }
    } else {
      const int y = 2 * max(ps, 0);
      const bf16_t* src = g.dY + (((int64_t)pb * g.OH + y) * g.OW + px0) * g.ldy;  // wave-uniform
      const char* dst = L.dy + (u & (NDMA - 1)) * DY_SLOT;
      const uint32_t off = (uint32_t)((lane >> 2) * g.ldy + (lane & 3) * 8) * 2u;
#pragma unroll
      for (int h = 0; h < 2; ++h) dma16(src + (int64_t)h * g.OW * g.ldy, off, dst + h * 1024);
    }
    if (pq < q_hi - 1 || ps < HS - 1) {
      if (++ps == HS) { ps = -LEAD; ++pq; px0 += 16; if (px0 == g.OW) { px0 = 0; ++pb; } }
    }
  };
  // (before the barrier that ends an odd step t) everything but the newest AHEAD - 4 steps (two transfers each) has landed: steps t + 1 and
  // t + 2 blend the block rows of steps t + 3, t + 4 (raw rows up to t + 4) and multiply the dY chunks t + 1, t + 2
  auto produce_wait = [&]() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AHEAD - 4)) : "memory"); };
  auto barrier = [&]() {
    if (ROLL_ABL & 16) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- blend, ON THE MATRIX PIPE.  The 2x resize of a block row is linear: hi-res pixel (dyb, p) of the patch = sum over the two low-res
  // rows h and <= 2 low-res columns of wy(h, dyb) * wx(col, p) * raw[h][col] (weights 1/16, 3/16, 9/16: exact in bf16), i.e. for one
  // 16-channel fragment   D [16 channels x 16 patch pixels] = A [16 channels x (2 rows x 16 columns)] . W [(2 x 16) x 16 pixels]
  // -- ONE 16x16x32 MFMA whose A operand is the usual transposed read of the raw rows (K = low-res pixels) and whose B operand is a
  // constant of the lane (eight registers).  Operands in this order leave each lane with 4 consecutive
  // channels of one pixel: one 8-B store.  (As ~36 VALU instructions per lane and channel pair the blend cost 42 of the launch's 194 us
  // at 1024 images: VALU issue and MFMA issue do not overlap on a SIMD.)  Products are exact and the fp32 sum of <= 4 terms rounds once
  // to bf16: within one bf16 ulp of blend2x2's two-stage lerp (equal for all but ~1e-4 of the values).
  // This wave: channel fragment cf, pixel segment seg = wave >> 2 (patch pixels 16 seg .. 16 seg + 15; 21 used), both rows dyb.
  const int seg = wave >> 2;
  const int bp = seg * 16 + (lane & 15);                          // patch pixel of this lane's output column
  const int raw_lane = pxl * 128 + cf * 32 + lp * 8;              // transposed read of a raw row: K pixel = low-res column pxl
  const int bout_lane = bp * PS + cf * 32 + (lane >> 4) * 8;      // D rows 4 (lane >> 4) + r = 4 consecutive channels
  int wx0 = (q_lo % g.nxs) * 16, ws = -LEAD;                      // x0 / s of the step to blend
  // B operands (weights) of the two rows dyb, constant per lane: K index 4 h + q of the lane's 8 <-> low-res row h, raw column
  // jj = 4 (lane >> 4) + q; output column lane & 15 <-> patch pixel bp = 2 jj0 - 1 + dxb.  (Columns outside the image are zeroed at the store.)
  {
    short8_t bw[2];
    const int dxb = (bp & 1) ? 0 : 1, jj0 = (bp + 1 - dxb) >> 1, gq = (lane >> 4) * 4;
#pragma unroll
    for (int dyb = 0; dyb < 2; ++dyb)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jj = gq + q;
          const float wx = jj == jj0 ? (dxb ? 0.25f : 0.75f) : jj == jj0 + 1 ? (dxb ? 0.75f : 0.25f) : 0.f;
          const float wy = (h == 0) == (dyb == 0) ? 0.75f : 0.25f;
          bw[dyb][h * 4 + q] = (short)(__float_as_uint(wx * wy) >> 16);       // 0, 1/16, 3/16, 9/16: exact in bf16
        }
    *(short8_t*)(L.bw + tid * 32) = bw[0];
    *(short8_t*)(L.bw + tid * 32 + 16) = bw[1];
  }
  const char* bw_lane = L.bw + tid * 32;
  short4_t ba_lo, ba_hi;                                          // raw rows of the step being blended (A operand), read at the top of the step
  ba_lo = ba_hi = (short4_t){0, 0, 0, 0};
  f32x4 bd[2];                                                    // the two rows' products in flight
  bd[0] = bd[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int w_slot = 0, w_i = 0, w_x = 0;
  auto blend_load = [&](int u) {
    if (ROLL_ABL & 2) return;
    ba_lo = tr16(L.raw + ((u - 1) & (NDMA - 1)) * RAW_SLOT + raw_lane);
    ba_hi = tr16(L.raw + (u & (NDMA - 1)) * RAW_SLOT + raw_lane);
    w_slot = (u & (NSLOT - 1)) * IN_SLOT; w_i = ws + 1; w_x = wx0 - PAD;
    if (++ws == HS) { ws = -LEAD; wx0 += 16; if (wx0 == g.OW) wx0 = 0; }
  };
  auto blend_mfma = [&]() {
    if (ROLL_ABL & 2) return;
    const short8_t af = (short8_t){ba_lo[0], ba_lo[1], ba_lo[2], ba_lo[3], ba_hi[0], ba_hi[1], ba_hi[2], ba_hi[3]};
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    const short8_t bw0 = *(const short8_t*)bw_lane, bw1 = *(const short8_t*)(bw_lane + 16);
    bd[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bw0), z, 0, 0, 0);
    bd[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bw1), z, 0, 0, 0);
  };
  auto blend_store = [&]() {
    if (ROLL_ABL & 2) return;
    char* out = L.in + PS + w_slot + bout_lane;
    const bool colin = (unsigned)(w_x + bp) < (unsigned)g.OW;     // hi-res column inside the image
#pragma unroll
    for (int dyb = 0; dyb < 2; ++dyb) {
      const bool rowin = (unsigned)(2 * w_i + 1 + dyb) < (unsigned)g.OH;       // wave-uniform: SAME padding rows are zeros
      const bf16x2 a = __builtin_convertvector((f32x2){bd[dyb][0], bd[dyb][1]}, bf16x2), b = __builtin_convertvector((f32x2){bd[dyb][2], bd[dyb][3]}, bf16x2);
      uint2 v = make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
      if (!(rowin && colin)) v = make_uint2(0u, 0u);
      if (bp < 16 + KS - 1) *(uint2*)(out + dyb * (PW * PS)) = v;
    }
  };

  // ---- prologue: the DMAs of steps 0 .. AHEAD-1, then the blended row of step 0
  if (producer) {
    for (int u = 0; u < AHEAD; ++u) produce(u);
    produce_wait();
  }
  barrier();
  blend_load(0);
  blend_mfma();
  blend_store();
  barrier();

  // The window: input rows rho = 0..6 of the step being multiplied (hi-res rows 2c-2 .. 2c+4) x this wave's three x shifts, in registers.
  // Every step -- lead-in steps too -- shifts it by two rows and reads the block row of that step (rows 5, 6) from its LDS slot: 6
  // transposed reads per step instead of 21; tap rows ky = 0..3 do not touch the new rows, so their latency hides under 24 MFMAs.
  // Round 4: a window column is ONE 16-register vector (register row r = elements 2r, 2r + 1; row 7 mirrors row 0), so that an A operand --
  // the register pair {row r, row r + 1} -- is four CONSECUTIVE registers of it and reaches the MFMA as a sub-register tuple: as separate
  // 2-register values every operand cost two v_mov (36 per step, beside 36 MFMAs: VALU issue is step time one to one).
  typedef int i32x16 __attribute__((ext_vector_type(16)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x16 win[KXW];
#pragma unroll
  for (int kx = 0; kx < KXW; ++kx)
#pragma unroll
    for (int e = 0; e < 16; ++e) win[kx][e] = 0;
  auto win_set = [&](int kx, auto RR, short4_t v) {            // register row RR (compile-time) <- a transposed read
    constexpr int rr = decltype(RR)::value;
    const int2 w = __builtin_bit_cast(int2, v);
    win[kx][2 * rr] = w.x; win[kx][2 * rr + 1] = w.y;
    if constexpr (rr == 0) { win[kx][14] = w.x; win[kx][15] = w.y; }       // the mirror of row 0 behind row 6: the pair (6, 0) is (6, 7)
  };
  short4_t nb[COF][2];                                            // dY fragments of the next step (read under this step's MFMAs)
#pragma unroll
  for (int j = 0; j < COF; ++j) nb[j][0] = nb[j][1] = (short4_t){0, 0, 0, 0};

  // iteration t: the DMAs of step t + AHEAD, the blend of step t + 1, the window update + MFMAs of step t - 1 (the blend runs two steps
  // ahead of its readers: one barrier per two iterations)
  int ms = -LEAD - 1;                                             // s of step t - 1 (iteration 0: no step)
  // PHc = -1: the window SHIFTS by two rows per step (30 v_mov per step: VALU issue is step time one to one);
  // PHc = 0..6: it ROTATES instead -- logical row r lives in register row (r + 2 PH) % 7, nothing moves; seven consecutive steps
  // (phases 1, 2, .., 6, 0) bring it back to the identity, so the main loop is unrolled by seven and the tail runs the shifting form.
  auto body = [&](int t, auto BL, auto PHc) {
    constexpr int PH = decltype(PHc)::value;
    auto P = [](int r) constexpr { return PH < 0 ? r : (r + 2 * PH) % (KS + 1); };
    const bool mm = ms >= 0;                                      // wave-uniform: this step multiplies (not a lead-in step)
    ROLL_STAMP(0);
    if (producer) produce(t + AHEAD);
    if (BL.value) blend_load(t + 1);
    ROLL_STAMP(1);
    {
      const char* rp = L.in + PS + ((t - 1) & (NSLOT - 1)) * IN_SLOT + in_lane;
#pragma unroll
      for (int kx = 0; kx < KXW; ++kx) {
        if constexpr (PH < 0) {
#pragma unroll
          for (int e = 0; e + 4 < 2 * (KS + 1); ++e) win[kx][e] = win[kx][e + 4];
        }
        if (!(ROLL_ABL & 8)) {
          win_set(kx, std::integral_constant<int, P(KS - 1)>{}, tr16(rp + kx * PS));
          win_set(kx, std::integral_constant<int, P(KS)>{}, tr16(rp + (PW + kx) * PS));
        }
      }
    }
    if (mm) {
      short8_t bfr[COF];
#pragma unroll
      for (int j = 0; j < COF; ++j)
        bfr[j] = (short8_t){nb[j][0][0], nb[j][0][1], nb[j][0][2], nb[j][0][3], nb[j][1][0], nb[j][1][1], nb[j][1][2], nb[j][1][3]};
      auto do_ky = [&](auto KYc) {                                // (ky as a compile-time constant: the operand is a shufflevector of the window column)
        constexpr int ky = decltype(KYc)::value;
        if (BL.value && ky == 1) blend_mfma();
        if (BL.value && ky == 3) blend_store();
        if (ky == 0) ROLL_STAMP(2);
        if (ky == 3) ROLL_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int ra = P(ky), rb = P(ky + 1);                 // register rows of the pair: consecutive, or (6, 0) = (6, 7: the mirror)
        static_assert(rb == ra + 1 || (ra == KS && rb == 0), "window pair");
#pragma unroll
        for (int kx = 0; kx < KXW; ++kx) {
          const i32x4 a4 = __builtin_shufflevector(win[kx], win[kx], 2 * ra, 2 * ra + 1, 2 * ra + 2, 2 * ra + 3);
          const short8_t af = __builtin_bit_cast(short8_t, a4);
#pragma unroll
          for (int j = 0; j < COF; ++j)
            if (!(ROLL_ABL & 1)) acc[ky * KXW + kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[j]),
                                                                            acc[ky * KXW + kx][j], 0, 0, 0);
        }
      };
      do_ky(integral_constant<int, 0>{}); do_ky(integral_constant<int, 1>{}); do_ky(integral_constant<int, 2>{});
      do_ky(integral_constant<int, 3>{}); do_ky(integral_constant<int, 4>{}); do_ky(integral_constant<int, 5>{});
      if (g.bslab) {                                              // waves 4, 5: column sums of dY fragment j = wave - 4 (an all-ones A operand)
        if (wave == 4) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bfr[0]), bacc, 0, 0, 0);
        else if (wave == 5) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bfr[1]), bacc, 0, 0, 0);
      }
    } else if (BL.value) {
      blend_mfma();
      blend_store();
    }
    ROLL_STAMP(4);
    if (++ms == HS) ms = -LEAD;
    if (ms >= 0) {                                                // the next step multiplies: its dY fragments (chunk t, published long ago)
      const char* sd = L.dy + (t & (NDMA - 1)) * DY_SLOT + dy_lane;
#pragma unroll
      for (int j = 0; j < COF; ++j) { nb[j][0] = tr16(sd + j * 32); nb[j][1] = tr16(sd + 1024 + j * 32); }
    }
    ROLL_STAMP(5);
    if (t & 1) {                                                  // one barrier per two steps
      if (producer) produce_wait();
      barrier();
    }
    ROLL_STAMP(6);
  };
  int t = 0;
  static const bool no_rot = false;
  for (; !no_rot && t + 7 < T; t += 7) {                          // (every t < T - 1 here: the blending form)
    body(t, std::true_type{}, integral_constant<int, 1>{});
    body(t + 1, std::true_type{}, integral_constant<int, 2>{});
    body(t + 2, std::true_type{}, integral_constant<int, 3>{});
    body(t + 3, std::true_type{}, integral_constant<int, 4>{});
    body(t + 4, std::true_type{}, integral_constant<int, 5>{});
    body(t + 5, std::true_type{}, integral_constant<int, 6>{});
    body(t + 6, std::true_type{}, integral_constant<int, 0>{});
  }
  for (; t + 1 < T; ++t) body(t, std::true_type{}, integral_constant<int, -1>{});
  for (; t <= T; ++t) body(t, std::false_type{}, integral_constant<int, -1>{});
}

__global__ __launch_bounds__(512, 1) void wgrad_roll_kernel(const RollMulti mg) {
  const RollArgs g = mg.a[blockIdx.z];                         // by value: every field lives in SGPRs (a reference re-reads the kernel argument segment inside the loop)
  __shared__ __attribute__((aligned(16))) char lds_in[PS + NSLOT * IN_SLOT];      // separate objects: the compiler orders LDS reads after
  __shared__ __attribute__((aligned(16))) char lds_raw[NDMA * RAW_SLOT];    // the DMAs only where they can alias
  __shared__ __attribute__((aligned(16))) char lds_dy[NDMA * DY_SLOT];
  __shared__ __attribute__((aligned(16))) char lds_bw[512 * 32];
#ifdef SV_ROLL_STAMP
  __shared__ long long lds_stamps[4 * 8 * NSTAMP];
  for (int i = threadIdx.x; i < 4 * 8 * NSTAMP; i += 512) lds_stamps[i] = 0;
  __syncthreads();
  const RollLds L = {lds_stamps, lds_in, lds_raw, lds_dy, lds_bw};
#else
  const RollLds L = {nullptr, lds_in, lds_raw, lds_dy, lds_bw};
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cf = wave & 3, kx0 = (wave >> 2) * KXW;
  const int SPS = (g.OH >> 1) + LEAD;                            // steps per strip
  // this workgroup's strips
  const int per = (g.nstrips + (int)gridDim.x - 1) / (int)gridDim.x;
  const int q_lo = (int)blockIdx.x * per, q_hi = min(g.nstrips, q_lo + per);
  const int T = q_hi > q_lo ? (q_hi - q_lo) * SPS : 0;          // steps of this workgroup

  f32x4 acc[KS * KXW][COF];                                       // [ky * KXW + kx - kx0][j]
#pragma unroll
  for (int t = 0; t < KS * KXW; ++t)
#pragma unroll
    for (int j = 0; j < COF; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (T > 0 && !(ROLL_ABL & 64)) {
    roll_loop(g, L, acc, bacc, q_lo, q_hi, T);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the producers' surplus DMAs
  }

  // ---- flush in the fragment order of wgrad_reduce <TPW 9, CIF 4, COF 2>: virtual wave v = tap / 9, fragment f = ((tap % 9) * 4 + ci-fragment) * 2 + j
  if (ROLL_ABL & 32) return;
  float* sl = g.slab + (int64_t)blockIdx.x * (4 * 72 * 256) + lane;
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int kx = 0; kx < KXW; ++kx)
#pragma unroll
      for (int j = 0; j < COF; ++j) {
        const int tap = ky * KS + kx0 + kx;                       // wave-uniform
        float* p = sl + ((tap / 9) * 72 + ((tap % 9) * 4 + cf) * 2 + j) * 256;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) p[r4 * 64] = acc[ky * KXW + kx][j][r4];
      }
#ifdef SV_ROLL_STAMP
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < 4 * 8 * NSTAMP; i += 512) ((long long*)g.slab)[i] = lds_stamps[i];
#endif
  if (g.bslab && (wave == 4 || wave == 5) && lane < 16) g.bslab[(int64_t)blockIdx.x * 128 + (wave - 4) * 16 + lane] = bacc[0];
}

}  // namespace

// workgroups per problem: one per CU over the launch, and at least `SV_ROLL_MIN_STRIPS` strips each (every workgroup flushes a 295-KB slab
// whatever it computed: at 64 images per network 128 one-strip workgroups wrote and re-read 75 MB for 19 steps of work each)
// (round 4: with two strips or fewer per workgroup, half as many workgroups -- 64 images per network: two strips each, step 0.607 -> 0.600 ms; 128: four
//  each, 0.766 -> 0.754; SV_ROLL_MIN_STRIPS forces a minimum instead: profiles/r04_b64_sweep4.txt)
static int roll_wgs(int n, int nstrips) {
  static const int min_strips = getenv("SV_ROLL_MIN_STRIPS") ? atoi(getenv("SV_ROLL_MIN_STRIPS")) : 0;
  int X = 256 / n;
  if (min_strips > 0) {
    const int cap = (nstrips + min_strips - 1) / min_strips;
    if (X > cap) X = cap;
  } else {
    if (nstrips <= 2 * X) X /= 2;
    if (X > nstrips) X = nstrips;
  }
  return X < 1 ? 1 : X;
}

bool svk_wgrad_roll_supported(const WgradArgs* wv, int n) {
  static const bool off = getenv("SV_NO_WGRAD_ROLL") != nullptr;
  if (off || n < 1 || n > SV_WGRAD_MAX_MULTI) return false;
  const WgradArgs& w = wv[0];
  if (!w.ups || w.S != 1 || w.SX != 1 || w.ntaps != NT || w.Cin_pad != 64 || w.Cin_real != 64 || w.ldy != 32 || w.ycols != 32 || w.N != 32) return false;
  if (w.fold_kw || w.clampin || w.dy_s2d || w.assign) return false;
  if (w.lOY < 1 || w.lOX < 4 || w.IH != w.OY || w.IW != w.OX) return false;
  for (int t = 0; t < NT; ++t)
    if (w.dy[t] != t / KS - PAD || w.dx[t] != t % KS - PAD) return false;
  const int B = w.M >> (w.lOY + w.lOX);
  const int nstrips = B * (w.OX / 16);
  const int X = roll_wgs(n, nstrips);
  const int64_t need = (int64_t)X * 4 * 72 * 256 * 4 + (int64_t)X * 128 * 4;
  for (int i = 0; i < n; ++i)
    if (!wv[i].ws || wv[i].ws_bytes < need) return false;
  return true;
}

int svk_wgrad_roll_multi(const WgradArgs* wv, int n, hipStream_t st) {
  if (!svk_wgrad_roll_supported(wv, n)) return SV_E_UNSUPPORTED;
  const WgradArgs& w = wv[0];
  const int B = w.M >> (w.lOY + w.lOX);
  RollMulti m;
  const int nxs = w.OX / 16, nstrips = B * nxs;
  const int X = roll_wgs(n, nstrips);
  WgradReduceDesc rd[SV_WGRAD_MAX_MULTI];
  for (int i = 0; i < n; ++i) {
    RollArgs& a = m.a[i];
    a.A = (const bf16_t*)wv[i].A; a.dY = (const bf16_t*)wv[i].dY;
    a.slab = wv[i].ws;
    a.bslab = wv[i].dbias ? wv[i].ws + (int64_t)X * 4 * 72 * 256 : nullptr;
    a.B = B; a.OH = w.OY; a.OW = w.OX; a.lda = w.lda; a.ldy = w.ldy; a.nxs = nxs; a.nstrips = nstrips;
    rd[i] = WgradReduceDesc{a.slab, wv[i].dW, a.bslab, wv[i].dbias, X, 1, 1, 64, 64, 32, NT, 0, 0, 0, 0, 9, 4, 2};
  }
  hipLaunchKernelGGL(wgrad_roll_kernel, dim3(X, 1, n), dim3(512), 0, st, m);
  SV_LAUNCH_CHECK();
  if (w.ev_mid[0]) { (void)hipEventRecord(w.ev_mid[0], st); (void)hipEventRecord(w.ev_mid[1], st); }
  if (w.defer && w.n_defer && *w.n_defer + n <= 64) {
    for (int i = 0; i < n; ++i) w.defer[(*w.n_defer)++] = rd[i];
    return SV_OK;
  }
  return svk_wgrad_reduce_all(rd, n, st);
}
