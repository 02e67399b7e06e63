// Polyphase WEIGHT GRADIENT of the decoder head (conv_geom.h: svg_poly; tests/test_polyphase_math.py has the algebra):
//
//   dW[ky,kx] = P(dW')[ky,kx] - dW_frame[ky,kx]
//
// dW' = the weight gradient of the 5x5 polyphase conv on the LOW-RES grid (wgrad_tile.hip, id 8: edge-clamped input tile, no
// blend arithmetic, dY read as its space-to-depth view) -- 25/42 of the x-packed form's MFMAs and a quarter of the pixels to
// stage.  This file holds the two small terms:
//   * poly_wgrad_frame_kernel: dW_frame = the taps of the five border rows / columns that leave the zero-padded image.  Per
//     border class c (poly_fix.hip's ten classes) and tap t: G[c][t][ci][co] = sum_b sum_pos line_b[pos + t][ci] dy_b[class pixel
//     pos][co] -- 1-D weight gradients along the fix lines, K = line pixels: the transposed-read MFMA scheme of
//     wgrad_tile.hip on the line buffers.  Six waves = six (line, class pair) blocks of 16 dY columns; a workgroup accumulates
//     over its images in registers and flushes one slab.
//   * poly_wgrad_reduce_kernel / poly_wgrad_project_kernel: slab sum in a fixed order, then
//     dW[ky,kx,ci,co] += sum_{py,px,ty,tx} Cy[py][ky][ty] Cx[px][kx][tx] dW'[(tx+2)*5+(ty+2)][ci][(py*2+px)*8+co]
//                        - sum_{row classes c: ky in Ey(c)} G[c][kx] - sum_{column classes c: kx in Ex(c)} G[c][ky],
//     dbias[co] += sum over the four parities (dbias' is zeroed for the next step; dW' is WRITTEN by the tile reduce: assign mode).
#include "common.hip.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ short4_t tr16p(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

struct PolyWgradMulti { const bf16_t* x[2]; const bf16_t* dy[2]; float* slab[2]; };

// blocks: (line, class pair): 0 top (0,1) | 1 bottom (2,3) | 2 bottom (4,-) | 3 left (5,6) | 4 right (7,8) | 5 right (9,-)
__device__ __forceinline__ void block_info(int blk, int& line, int& c0, int& c1) {
  line = blk == 0 ? 0 : blk <= 2 ? 1 : blk == 3 ? 2 : 3;
  c0 = blk == 0 ? 0 : blk == 1 ? 2 : blk == 2 ? 4 : blk == 3 ? 5 : blk == 4 ? 7 : 9;
  c1 = (blk == 2 || blk == 5) ? -1 : c0 + 1;
}

template <int CIF>    // Cin / 16
__global__ __launch_bounds__(384) void poly_wgrad_frame_kernel(const PolyWgradMulti mg, int B, int h, int w, int lda) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CIN = 16 * CIF, PSL = CIN * 2 + 32;       // line pixel pitch: PS/32 odd (conflict-free transposed reads)
  const bf16_t* __restrict__ x = mg.x[blockIdx.y];
  const bf16_t* __restrict__ dy = mg.dy[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H2 = 2 * h, W2 = 2 * w, L = H2 > W2 ? H2 : W2, LW = L + 5;
  char* sLine = smem;                                      // [4][LW] pixels of PSL bytes
  char* sDy = smem + 4 * LW * PSL;                         // [6][L] records of 32 B (two classes x 8 channels)
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int r = 4 * lg + lq;
  int bl, c0, c1;
  block_info(wave, bl, c0, c1);
  const int npos = bl < 2 ? W2 : H2;                       // pixels along this block's line
  f32x4 acc[6][CIF];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int i = 0; i < CIF; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto edge_of = [&](int cls, int n2) { const int c5 = cls % 5; return c5 == 0 ? 0 : c5 == 1 ? 1 : n2 - 5 + c5; };

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();                                       // the previous image is consumed
    const bf16_t* xb = x + (int64_t)b * h * w * lda;
    const bf16_t* dyb = dy + (int64_t)b * H2 * W2 * 8;
    // ---- lines (as poly_fix.hip): rows replicate-extended, columns zero-extended
    for (int it = tid; it < 4 * LW * (CIN / 8); it += 384) {
      const int ch = it % (CIN / 8), li = (it / (CIN / 8)) % LW, line = it / ((CIN / 8) * LW);
      const bool is_row = line < 2;
      const int n = is_row ? w : h;
      int u = li - 2;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (li < 2 * n + 5 && (is_row || (u >= 0 && u < 2 * n))) {
        u = min(max(u, 0), 2 * n - 1);
        const int m = u >> 1;
        const int i0 = (u & 1) ? m : max(m - 1, 0), i1 = (u & 1) ? min(m + 1, n - 1) : m;
        const float f = (u & 1) ? 0.25f : 0.75f;
        const int64_t o0 = is_row ? ((int64_t)(line == 0 ? 0 : h - 1) * w + i0) * lda : ((int64_t)i0 * w + (line == 2 ? 0 : w - 1)) * lda;
        const int64_t o1 = is_row ? ((int64_t)(line == 0 ? 0 : h - 1) * w + i1) * lda : ((int64_t)i1 * w + (line == 2 ? 0 : w - 1)) * lda;
        const uint4 a0 = *(const uint4*)(xb + o0 + ch * 8), a1 = *(const uint4*)(xb + o1 + ch * 8);
        f32x2 p0[4], p1[4], rr[4];
        Piece<bf16_t>::unpack(a0, p0); Piece<bf16_t>::unpack(a1, p1);
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[e] = lerp2(p0[e], p1[e], f);
        v = Piece<bf16_t>::pack(rr);
      }
      *(uint4*)(sLine + (line * LW + li) * PSL + ch * 16) = v;
    }
    // ---- dY of the border classes: [block][pos][class-in-pair][8]
    for (int it = tid; it < 6 * L * 2; it += 384) {
      const int half = it & 1, pos = (it >> 1) % L, blk = (it >> 1) / L;
      int line, k0, k1;
      block_info(blk, line, k0, k1);
      const int cls = half ? k1 : k0;
      const bool rows = line < 2;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (cls >= 0 && pos < (rows ? W2 : H2)) {
        const int e = edge_of(cls, rows ? H2 : W2);
        const int rr2 = rows ? e : pos, cc = rows ? pos : e;
        v = *(const uint4*)(dyb + ((int64_t)rr2 * W2 + cc) * 8);
      }
      *(uint4*)(sDy + ((blk * L + pos) * 2 + half) * 16) = v;
    }
    __syncthreads();
    // ---- MFMA: K = line pixels (32 per chunk); D rows = input channels, columns = (class-in-pair, co)
    const char* lineb = sLine + bl * LW * PSL;
    const char* dyblk = sDy + wave * L * 32;
    for (int kc = 0; kc < npos / 32; ++kc) {
      const short4_t blo = tr16p(dyblk + (kc * 32 + r) * 32 + 8 * lp), bhi = tr16p(dyblk + (kc * 32 + 16 + r) * 32 + 8 * lp);
      const short8_t bfr = (short8_t){blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < CIF; ++i) {
          const short4_t alo = tr16p(lineb + (kc * 32 + r + t) * PSL + 8 * lp + i * 32);
          const short4_t ahi = tr16p(lineb + (kc * 32 + 16 + r + t) * PSL + 8 * lp + i * 32);
          const short8_t af = (short8_t){alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
          acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr), acc[t][i], 0, 0, 0);
        }
    }
  }
  // ---- flush: slab[wg][block][tap][cifrag][4][64]
  float* sl = mg.slab[blockIdx.y] + (((int64_t)blockIdx.x * 6 + wave) * 6 * CIF) * 256 + lane;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int i = 0; i < CIF; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) sl[((t * CIF + i) * 4 + r4) * 64] = acc[t][i][r4];
}

struct PolyWgradFin { const float* slab[2]; float* gsum[2]; float* dWp[2]; float* dbp[2]; float* dW[2]; float* dbias[2]; };

// slab sum in a fixed order: a block owns 32 float4 columns, its 8 thread rows take the slabs k = row, row + 8, ... (16-B loads,
// many in flight) and are combined through LDS in row order (the scheme of wgrad_reduce_kernel)
__global__ __launch_bounds__(256) void poly_wgrad_reduce_kernel(const PolyWgradFin f, int nwg, int per) {
  __shared__ float4 part[8][32];
  const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int e = (blockIdx.x * 32 + col) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < per) {
    const float* s = f.slab[blockIdx.y] + e;
#pragma unroll 8
    for (int k = row; k < nwg; k += 8) {
      const float4 v = *(const float4*)(s + (int64_t)k * per);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  part[row][col] = a;
  __syncthreads();
  if (row || e >= per) return;
#pragma unroll
  for (int r = 1; r < 8; ++r) { const float4 v = part[r][col]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  *(float4*)(f.gsum[blockIdx.y] + e) = a;
}

// gsum element of (block, tap, cifrag, r4, lane): D row = ci (lane>>4)*4 + r4 within the fragment, column = lane & 15
__device__ __forceinline__ float g_at(const float* __restrict__ gs, int CIF, int cls, int tap, int ci, int co) {
  int blk, half;
  if (cls <= 1) { blk = 0; half = cls; }
  else if (cls <= 3) { blk = 1; half = cls - 2; }
  else if (cls == 4) { blk = 2; half = 0; }
  else if (cls <= 6) { blk = 3; half = cls - 5; }
  else if (cls <= 8) { blk = 4; half = cls - 7; }
  else { blk = 5; half = 0; }
  const int i = ci >> 4, cl = ci & 15, lg = cl >> 2, r4 = cl & 3, col = half * 8 + co;
  return gs[(((blk * 6 + tap) * CIF + i) * 4 + r4) * 64 + lg * 16 + col];
}

__device__ __forceinline__ float pcoef(int p, int k, int t) {      // conv_api.hip prep_poly / tests/test_polyphase_math.py
  const int hh = p + k - 2, m = hh >> 1;
  if (hh & 1) return t == m ? 0.75f : t == m + 1 ? 0.25f : 0.f;
  return t == m - 1 ? 0.25f : t == m ? 0.75f : 0.f;
}
__device__ __forceinline__ bool excl(int c5, int k) {              // tap k leaves the image at border class c5 (0..4)
  return c5 == 0 ? k <= 1 : c5 == 1 ? k == 0 : c5 == 2 ? k == 5 : c5 == 3 ? k >= 4 : k >= 3;
}

__global__ __launch_bounds__(256) void poly_wgrad_project_kernel(const PolyWgradFin f, int Cin, int Cout) {
  const int z = blockIdx.y, CIF = Cin / 16;
  const int idx = blockIdx.x * 256 + threadIdx.x, total = 36 * Cin * Cout;
  const float* __restrict__ dWp = f.dWp[z];
  const float* __restrict__ gs = f.gsum[z];
  if (idx < total) {
    const int co = idx % Cout, ci = (idx / Cout) % Cin, kk = idx / (Cout * Cin), ky = kk / 6, kx = kk % 6;
    float v = 0.f;
    for (int py = 0; py < 2; ++py)
      for (int ty = -2; ty <= 2; ++ty) {
        const float cy = pcoef(py, ky, ty);
        if (cy == 0.f) continue;
        for (int px = 0; px < 2; ++px)
          for (int tx = -2; tx <= 2; ++tx) {
            const float cx = pcoef(px, kx, tx);
            if (cx != 0.f) v += cy * cx * dWp[(((tx + 2) * 5 + (ty + 2)) * Cin + ci) * 32 + (py * 2 + px) * 8 + co];
          }
      }
    for (int c5 = 0; c5 < 5; ++c5) {
      if (excl(c5, ky)) v -= g_at(gs, CIF, c5, kx, ci, co);          // row class: excluded ky, tap = kx
      if (excl(c5, kx)) v -= g_at(gs, CIF, 5 + c5, ky, ci, co);      // column class: excluded kx, tap = ky
    }
    f.dW[z][idx] += v;
  }
  if (blockIdx.x == 0 && threadIdx.x < Cout && f.dbias[z]) {      // dbias = the four parities of the polyphase dbias' (no frame term)
    float s = 0.f;
    for (int p = 0; p < 4; ++p) s += f.dbp[z][p * 8 + threadIdx.x];
    f.dbias[z][threadIdx.x] += s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 32) f.dbp[z][threadIdx.x] = 0.f;   // the main kernel adds to it atomically: zero for the next step
}

}  // namespace

// floats per problem of the frame slabs / sums
static inline int poly_wgrad_per(int Cin) { return 6 * 6 * (Cin / 16) * 256; }
int64_t svk_poly_wgrad_ws_floats(int Cin, int nwg) { return (int64_t)25 * Cin * 32 + 32 + poly_wgrad_per(Cin) * (1 + (int64_t)nwg); }

static inline size_t poly_wgrad_frame_lds(int h, int w, int Cin) {
  const int L = 2 * (h > w ? h : w), LW = L + 5, PSL = Cin * 2 + 32;
  return (size_t)4 * LW * PSL + (size_t)6 * L * 32;
}
// the frame kernel's line buffers must fit the CU's LDS (160 KB on gfx950): low-res extents up to 128 at Cin = 32
bool svk_poly_wgrad_supported(int h, int w, int Cin, int Cout) {
  return (Cin == 32 || Cin == 64) && Cout <= 8 && poly_wgrad_frame_lds(h, w, Cin) <= 160 * 1024;
}

// frame term + reduce + projection for n <= 2 twin problems.  ws[i] (floats, zero on first use): [dWp 25*Cin*32][dbp 32][gsum][slabs nwg]
int svk_poly_wgrad_finish(int n, const void* const* x_lo, const void* const* dy, float* const* ws, float* const* dW, float* const* dbias,
                          int B, int h, int w, int lda, int Cin, int Cout, int nwg, hipStream_t st) {
  if (n < 1 || n > 2 || nwg < 1) return SV_E_UNSUPPORTED;
  const int per = poly_wgrad_per(Cin);
  PolyWgradMulti m;
  PolyWgradFin f;
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    float* base = ws[k];
    f.dWp[i] = base; f.dbp[i] = base + 25 * Cin * 32; f.gsum[i] = f.dbp[i] + 32;
    float* slab = f.gsum[i] + per;
    m.x[i] = (const bf16_t*)x_lo[k]; m.dy[i] = (const bf16_t*)dy[k]; m.slab[i] = slab;
    f.slab[i] = slab; f.dW[i] = dW[k]; f.dbias[i] = dbias ? dbias[k] : nullptr;
  }
  const size_t lds = poly_wgrad_frame_lds(h, w, Cin);
  if (!svk_poly_wgrad_supported(h, w, Cin, Cout)) return SV_E_UNSUPPORTED;   // (callers check BEFORE they launch the main term)
  // 128x128 images: 75.6 KB of line buffers -- above the 64 KB a kernel gets without the attribute, well inside gfx950's 160 KB
  if (Cin == 32) {
    sv_ensure_dynamic_lds((const void*)poly_wgrad_frame_kernel<2>, lds);
    hipLaunchKernelGGL((poly_wgrad_frame_kernel<2>), dim3(nwg, n), dim3(384), lds, st, m, B, h, w, lda);
  } else {
    sv_ensure_dynamic_lds((const void*)poly_wgrad_frame_kernel<4>, lds);
    hipLaunchKernelGGL((poly_wgrad_frame_kernel<4>), dim3(nwg, n), dim3(384), lds, st, m, B, h, w, lda);
  }
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL(poly_wgrad_reduce_kernel, dim3((per + 127) / 128, n), dim3(256), 0, st, f, nwg, per);
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL(poly_wgrad_project_kernel, dim3((36 * Cin * Cout + 255) / 256, n), dim3(256), 0, st, f, Cin, Cout);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
