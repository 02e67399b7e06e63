// POLYPHASE WEIGHT GRADIENT at the reference's precision (fp32), for every (UpSampling2D(bilinear) -> Conv2D(padding='same')) layer of the decoders
// (vae/model.py:154-156,:163-167: d4 = Conv2D(32, 6), d5 = Conv2D(6, 6); Conv2DBackpropFilter + BiasAddGrad of vae/trainer.py:137):
//
//   dW[ky,kx] = sum_{classes c = (py,px)} sum_{ty,tx} cy(py,ky,ty) cx(px,kx,tx) dW'_c[ty,tx]  -  dW_frame[ky,kx]
//
//   dW'_c[ty,tx][ci][co] = sum_{b,i,j} x~[b, i+ty, j+tx, ci] dy[b, 2i+py, 2j+px, co]      on the LOW-RES grid, x~ = the edge-clamped low-res input
//
// (tests/test_polyphase_math.py pins the algebra).  The main terms run on wgrad_tile_f32.hip: per parity class with only the offsets that parity touches
// (svg_polyc: 25 / 20 / 20 / 16 taps, 81 of the direct form's 144 tap products) or, for the thin head, as ONE 25-tap problem whose 32 columns are the four
// parities x 8 channels (svg_poly; the x-packed direct form issues 44 tap slots x 16 columns per pixel pair).  This file holds the small terms:
//   * polyc_wgrad_frame_kernel: dW_frame = the taps of the K-1 border rows / columns per side that leave the zero-padded image: per border class c and
//     tap t, G[c][t][ci][co] = sum_b sum_pos line_b[pos + t - pad][ci] dy_b[class pixel pos][co] -- 1-D weight gradients along the upsampled edge lines
//     (rows replicate-extended, columns zero-extended: poly_fix.hip's lines), K = line pixels, four per v_mfma_f32_16x16x4_f32.  A workgroup owns one
//     class and a group of images, accumulates in registers and leaves one slab;
//   * polyc_wgrad_project_kernel: the projection above, the frame slabs summed in group order (deterministic), added into dW.
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"
#include "conv_geom.h"

namespace {

// image groups per border class (workgroups = 2 (K-1) x groups x networks).  Measured (2 x 512 images, d4 / d5 frame kernel + projection): 16 groups, one image per
// barrier pair: 177 / 131 + 30 us; 32 groups, 4 images (72-86 KB of LDS: one workgroup per CU): 160 / 165 + 52 us; the loop is all latency (global loads -> lerp ->
// LDS -> barrier -> 8-16 short K steps), so the next image's loads are issued BEFORE the current image's MFMAs and land in LDS after them (register prefetch).
constexpr int FRAME_GROUPS_MAX = 64; // (round 5, with the group sum as its own streaming pass: more, shorter workgroups hide each other's load latency)
static int frame_groups() {
  static const int g = getenv("SV_POLYC_FRAME_GROUPS") ? atoi(getenv("SV_POLYC_FRAME_GROUPS")) : 32;      // in the step (2 x 512 images): 16 groups 9.80-9.84 ms, 32: 9.795, 64: 9.86-9.89 (serial: 137 / 101 / 101 us for d4)
  return g < 1 ? 1 : g > FRAME_GROUPS_MAX ? FRAME_GROUPS_MAX : g;
}

struct PolycFrameMulti { const float* x[2]; const float* dy[2]; float* slab[2]; };

// CIF / COF: 16-channel fragments of the input / of dY (the head's 6 -> 8 channels fill half a fragment); NFW = fragments per wave = ceil(K CIF COF / 4);
// NLI / NDI: line / dY 16-B items per thread and image (host-checked upper bounds)
template <int K, int CIF, int COF, int NLI, int NDI>
__global__ __launch_bounds__(256) void polyc_wgrad_frame_kernel(const PolycFrameMulti mg, int B, int h, int w, int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CIN = 16 * CIF, NFRAG = K * CIF * COF, NFW = (NFRAG + 3) / 4, NC = K - 1, PAD = (K - 1) / 2;
  constexpr int PSL = CIN * 4 + 64, YSL = COF * 64 + 64;      // pitches: the four pixels of an operand (lane >> 4) land 16 banks apart
  static_assert(4 % COF == 0, "a wave keeps one dY column fragment");
  const float* __restrict__ x = mg.x[blockIdx.z];
  const float* __restrict__ dy = mg.dy[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int cls = blockIdx.x, c = cls % NC;
  const bool rows = cls < NC;
  const int H2 = 2 * h, W2 = 2 * w, L = H2 > W2 ? H2 : W2, LW = L + K - 1;
  const int npos = rows ? W2 : H2, m2 = rows ? H2 : W2;
  const int edge = c < PAD ? c : m2 - (K - 1 - PAD) + (c - PAD);
  const int line = (rows ? 0 : 2) + (c >= PAD ? 1 : 0), n = rows ? w : h;
  char* sLine = smem;                    // [LW] pixels of PSL bytes
  char* sDy = smem + LW * PSL;           // [L] pixels of YSL bytes
  f32x4 acc[NFW];
  int aoff[NFW];
#pragma unroll
  for (int q = 0; q < NFW; ++q) {
    acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int f = min(wave + 4 * q, NFRAG - 1), gq = f / COF, tap = gq / CIF, cif = gq % CIF;
    aoff[q] = (tap + kq) * PSL + (cif * 16 + lr) * 4;
  }
  const int boff = kq * YSL + ((wave % COF) * 16 + lr) * 4;
  // ---- this thread's items (the same for every image): line pieces (two source pixels + blend weight -> one LDS slot), dY pieces
  int l_o0[NLI], l_o1[NLI], l_dst[NLI], d_src[NDI], d_dst[NDI];
  float l_f[NLI];
#pragma unroll
  for (int s = 0; s < NLI; ++s) {
    const int it = tid + s * 256;
    l_dst[s] = -1; l_o0[s] = l_o1[s] = -1; l_f[s] = 0.f;
    if (it < LW * (CIN / 4)) {
      const int ch = it % (CIN / 4), li = it / (CIN / 4);
      int u = li - PAD;
      l_dst[s] = li * PSL + ch * 16;
      if (li < 2 * n + K - 1 && (rows || (u >= 0 && u < 2 * n))) {
        u = min(max(u, 0), 2 * n - 1);
        const int m = u >> 1;
        const int i0 = (u & 1) ? m : max(m - 1, 0), i1 = (u & 1) ? min(m + 1, n - 1) : m;
        l_f[s] = (u & 1) ? 0.25f : 0.75f;
        l_o0[s] = (rows ? ((line == 0 ? 0 : h - 1) * w + i0) : (i0 * w + (line == 2 ? 0 : w - 1))) * CIN + ch * 4;
        l_o1[s] = (rows ? ((line == 0 ? 0 : h - 1) * w + i1) : (i1 * w + (line == 2 ? 0 : w - 1))) * CIN + ch * 4;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NDI; ++s) {
    const int it = tid + s * 256;
    d_dst[s] = -1; d_src[s] = -1;
    if (it < L * (COF * 4)) {
      const int ch = it % (COF * 4), pos = it / (COF * 4);
      d_dst[s] = pos * YSL + ch * 16;
      if (pos < npos && ch * 4 < ldy) d_src[s] = ((rows ? edge : pos) * W2 + (rows ? pos : edge)) * ldy + ch * 4;
    }
  }
  uint4 ra0[NLI], ra1[NLI];
  float4 rd[NDI];
  auto fetch = [&](int b) {
    const float* xb = x + (int64_t)b * h * w * CIN;
    const float* dyb = dy + (int64_t)b * H2 * W2 * ldy;
#pragma unroll
    for (int s = 0; s < NLI; ++s) {
      ra0[s] = ra1[s] = make_uint4(0, 0, 0, 0);
      if (l_o0[s] >= 0) { ra0[s] = *(const uint4*)(xb + l_o0[s]); ra1[s] = *(const uint4*)(xb + l_o1[s]); }
    }
#pragma unroll
    for (int s = 0; s < NDI; ++s) rd[s] = d_src[s] >= 0 ? *(const float4*)(dyb + d_src[s]) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  const int per = (B + (int)gridDim.y - 1) / (int)gridDim.y, b_lo = (int)blockIdx.y * per, b_hi = min(B, b_lo + per);
  if (b_lo < b_hi) fetch(b_lo);
  for (int b = b_lo; b < b_hi; ++b) {
    __syncthreads();                     // the previous image is consumed
#pragma unroll
    for (int s = 0; s < NLI; ++s) {
      if (l_dst[s] < 0) continue;
      f32x2 p0[2], p1[2], r[2];
      Piece<float>::unpack(ra0[s], p0); Piece<float>::unpack(ra1[s], p1);
      r[0] = lerp2(p0[0], p1[0], l_f[s]); r[1] = lerp2(p0[1], p1[1], l_f[s]);          // the arithmetic of the forward's lines (poly_fix.hip); unsourced slots: 0
      *(uint4*)(sLine + l_dst[s]) = Piece<float>::pack(r);
    }
#pragma unroll
    for (int s = 0; s < NDI; ++s)
      if (d_dst[s] >= 0) *(float4*)(sDy + d_dst[s]) = rd[s];
    __syncthreads();
    if (b + 1 < b_hi) fetch(b + 1);      // in flight during this image's MFMAs
#pragma unroll 4
    for (int p0 = 0; p0 < npos; p0 += 4) {
      const float bv = *(const float*)(sDy + p0 * YSL + boff);
#pragma unroll
      for (int q = 0; q < NFW; ++q) {
        const float av = *(const float*)(sLine + p0 * PSL + aoff[q]);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[q], 0, 0, 0);     // D rows = input channels, columns = dY channels
      }
    }
  }
  // slab[group][class][fragment f = (tap * CIF + cif) * COF + cof][register][lane]
  float* sl = mg.slab[blockIdx.z] + ((int64_t)blockIdx.y * (2 * NC) + cls) * (NFRAG * 256) + lane;
#pragma unroll
  for (int q = 0; q < NFW; ++q) {
    const int f = wave + 4 * q;
    if (f >= NFRAG) continue;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) sl[(f * 4 + r4) * 64] = acc[q][r4];
  }
}

template <int K, int CIF, int COF>
static int launch_frame(const PolycFrameMulti& m, dim3 grid, int B, int h, int w, int ldy, hipStream_t st) {
  constexpr int NLI = CIF == 4 ? 5 : 3, NDI = 2;           // up to 64-pixel lines (69 line slots x CIN / 4 pieces; 64 x COF x 4 dY pieces)
  const int L = 2 * (h > w ? h : w), LW = L + K - 1;
  if (LW * (CIF * 4) > NLI * 256 || L * (COF * 4) > NDI * 256) return SV_E_UNSUPPORTED;
  const size_t lds = (size_t)LW * (CIF * 64 + 64) + (size_t)L * (COF * 64 + 64);
  sv_ensure_dynamic_lds((const void*)polyc_wgrad_frame_kernel<K, CIF, COF, NLI, NDI>, lds);
  hipLaunchKernelGGL((polyc_wgrad_frame_kernel<K, CIF, COF, NLI, NDI>), grid, dim3(256), lds, st, m, B, h, w, ldy);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// frame slabs of all image groups -> one slab, groups added in group order (deterministic): 16-B streaming reads, the projection then reads one value per term
struct PolycFrameSum { const float* slab[2]; float* sum[2]; };
__global__ __launch_bounds__(256) void polyc_frame_sum_kernel(const PolycFrameSum f, int groups, int per) {
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= per) return;
  const float* p = f.slab[blockIdx.y] + e;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int g = 0; g < groups; ++g) {
    const float4 v = *(const float4*)(p + (int64_t)g * per);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *(float4*)(f.sum[blockIdx.y] + e) = s;
}

struct PolycProject {
  const float* dWp[2]; float* dbp[2]; const float* slab[2]; float* dW[2]; float* dbias[2];
  int K, Cin, Cout, merged, groups, coff[4];      // coff: float offsets of the class buffers in dWp (per-class form)
};

__global__ __launch_bounds__(256) void polyc_wgrad_project_kernel(const PolycProject f) {
  const int z = blockIdx.y, K = f.K, Cin = f.Cin, Cout = f.Cout, pad = (K - 1) / 2, nc = K - 1;
  const int CIF = Cin / 16, COF = (Cout + 15) / 16, NFRAG = K * CIF * COF;
  const int idx = blockIdx.x * 256 + threadIdx.x, total = K * K * Cin * Cout;
  const float* __restrict__ dWp = f.dWp[z];
  const float* __restrict__ gs = f.slab[z];
  if (idx < total) {
    const int co = idx % Cout, ci = (idx / Cout) % Cin, kk = idx / (Cout * Cin), ky = kk / K, kx = kk % K;
    float v = 0.f;
    for (int py = 0; py < 2; ++py) {
      int ty0;
      const int nty = svg_polyc_taps(K, py, &ty0);
      for (int tyi = 0; tyi < nty; ++tyi) {
        const float cy = svg_pcoef(py, ky, ty0 + tyi, pad);
        if (cy == 0.f) continue;
        for (int px = 0; px < 2; ++px) {
          int tx0;
          const int ntx = svg_polyc_taps(K, px, &tx0);
          for (int txi = 0; txi < ntx; ++txi) {
            const float cx = svg_pcoef(px, kx, tx0 + txi, pad);
            if (cx == 0.f) continue;
            const int cls = py * 2 + px;
            // merged head: dW'[t = (tx+2)*5 + (ty+2)][ci][(py*2+px)*8 + co] (conv_api.hip: svg_poly_wgrad_args); per class: dW'_c[t = txi*nty + tyi][ci][co]
            const float g = f.merged ? dWp[(((tx0 + txi + 2) * 5 + (ty0 + tyi + 2)) * Cin + ci) * 32 + cls * 8 + co]
                                     : dWp[f.coff[cls] + ((txi * nty + tyi) * Cin + ci) * Cout + co];
            v += cy * cx * g;
          }
        }
      }
    }
    // frame: the fragment element of (class, tap, ci, co), summed over the image groups in group order
    const int cl = ci & 15, lane = (cl >> 2) * 16 + (co & 15), r4 = cl & 3;
    auto gsum = [&](int cls, int tap) {
      const int fr = (tap * CIF + (ci >> 4)) * COF + (co >> 4);
      const float* p = gs + (int64_t)cls * (NFRAG * 256) + (fr * 4 + r4) * 64 + lane;
      float s = 0.f;
      for (int g = 0; g < f.groups; ++g) s += p[(int64_t)g * (2 * nc) * (NFRAG * 256)];
      return s;
    };
    for (int c = 0; c < nc; ++c) {
      if (svg_polyc_excl(K, c, ky)) v -= gsum(c, kx);            // row class: excluded ky, tap = kx
      if (svg_polyc_excl(K, c, kx)) v -= gsum(nc + c, ky);       // column class: excluded kx, tap = ky
    }
    f.dW[z][idx] += v;
  }
  if (blockIdx.x == 0 && threadIdx.x < Cout && f.dbias[z]) {       // dbias' blocks: four parities x 8 columns (merged head) / four classes x 32
    float s = 0.f;
    for (int p = 0; p < 4; ++p) s += f.dbp[z][p * (f.merged ? 8 : 32) + threadIdx.x];
    f.dbias[z][threadIdx.x] += s;
  }
}

static inline int64_t frame_per(int K, int Cin, int Cout) { return (int64_t)K * (Cin / 16) * ((Cout + 15) / 16) * 256; }
static inline int64_t dwp_floats(const sv_conv_desc* d, int merged, int coff[4]) {
  if (merged) return (int64_t)25 * svg_cin_pad(d) * 32;
  int64_t n = 0;
  for (int c = 0; c < 4; ++c) {
    int t0;
    if (coff) coff[c] = (int)n;
    n += (int64_t)svg_polyc_taps(d->KH, c >> 1, &t0) * svg_polyc_taps(d->KH, c & 1, &t0) * svg_cin_pad(d) * d->Cout;
  }
  return n;
}

}  // namespace

// does the layer have a polyphase weight gradient at fp32?  1: per class (svg_polyc), 2: merged head (svg_poly)
int svg_polyc_wgrad_form(const sv_conv_desc* d) {
  static const bool off = getenv("SV_NO_POLYC_WGRAD") != nullptr;
  if (off || d->dtype != SV_F32) return 0;
  const int cin = svg_cin_pad(d);
  if (svg_polyc(d) && d->KH == 6 && d->Cout == 32 && (cin == 64 || cin == 32) && d->H / 2 >= 8 && d->W / 2 >= 8 && d->H <= 64 && d->W <= 64) return 1;    // (instantiations: d4)
  if (svg_poly(d) && cin == 32 && d->H / 2 >= 8 && d->W / 2 >= 8 && d->H <= 64 && d->W <= 64) return 2;      // (frame kernel: lines of up to 64 pixels)
  return 0;
}

// floats of the per-problem workspace: [dW' of every class | the merged dW'] [dbias' 128: four parities x 8 (merged) / four classes x 32] [frame slabs + their sum]
int64_t svk_polyc_wgrad_ws_floats(const sv_conv_desc* d) {
  const int form = svg_polyc_wgrad_form(d);
  if (!form) return 0;
  return dwp_floats(d, form == 2, nullptr) + 128 + (int64_t)(FRAME_GROUPS_MAX + 1) * 2 * (d->KH - 1) * frame_per(d->KH, svg_cin_pad(d), d->Cout);   // (+ 1: the groups' sum)
}

void svg_polyc_wgrad_args(const sv_conv_desc* d, int cls, WgradArgs* a) {
  memset(a, 0, sizeof(*a));
  const int h = d->H / 2, w = d->W / 2, cpad = svg_cin_pad(d), K = d->KH, py = cls >> 1, px = cls & 1;
  int ty0, tx0;
  const int nty = svg_polyc_taps(K, py, &ty0), ntx = svg_polyc_taps(K, px, &tx0);
  a->M = d->B * h * w; a->lOY = ilog2_exact(h); a->lOX = ilog2_exact(w); a->OY = h; a->OX = w;
  a->IH = h; a->IW = w; a->lda = d->ldx; a->S = 1; a->SX = 1;
  a->ldy = svg_gdy(d); a->ycols = svg_gdy(d); a->cl2 = ilog2_exact(cpad / svg_epp(d));
  a->Cin_pad = cpad; a->Cin_real = d->Cin; a->N = d->Cout; a->ntaps = nty * ntx; a->Nrows = a->ntaps * cpad;
  a->clampin = 1; a->dy_os = 2; a->dy_oy = py; a->dy_ox = px; a->assign = 1; a->msplit = a->M;
  for (int txi = 0; txi < ntx; ++txi)
    for (int tyi = 0; tyi < nty; ++tyi) { a->dy[txi * nty + tyi] = (int8_t)(ty0 + tyi); a->dx[txi * nty + tyi] = (int8_t)(tx0 + txi); }
}

// n <= 2 twin layers: main terms (one launch per parity class, both networks each) + frame + projection.  dW / dbias are ADDED to.  slab_ws[i]: >= slab_bytes
// of partial-sum workspace (SV_WGRAD_WS_BYTES), pw[i]: svk_polyc_wgrad_ws_floats(d) floats (no state between calls).  SV_E_UNSUPPORTED (nothing launched)
// when the layer has no such form.
int svk_polyc_wgrad_multi(const sv_conv_desc* d, int n, const void* const* x_lo, const void* const* dy, float* const* dW, float* const* dbias,
                          float* const* slab_ws, int64_t slab_bytes, float* const* pw, hipStream_t st) {
  const int form = svg_polyc_wgrad_form(d);
  if (!form || n < 1 || n > 2) return SV_E_UNSUPPORTED;
  const int merged = form == 2, K = d->KH, cin = svg_cin_pad(d), h = d->H / 2, w = d->W / 2;
  PolycProject pj;
  memset(&pj, 0, sizeof(pj));
  const int64_t ndwp = dwp_floats(d, merged, pj.coff);
  WgradArgs a[2];
  // the class launches leave their slabs in their own quarter of the partial-sum workspace; ONE reduce launch sums them all (four 10-us launches less)
  WgradReduceDesc pend[8];
  int npend = 0;
  const int nparts = merged ? 1 : 4;
  const int64_t part = (slab_bytes / nparts) & ~(int64_t)255;
  // dbias': the head's four parities side by side / one 32-column block per class (the class reduces run in ONE launch: they must not add into the same
  // dbias); the main terms' reduce ADDS its bias partials, the projection sums the blocks in a fixed order
  for (int i = 0; i < n; ++i)
    if (hipMemsetAsync(pw[i] + ndwp, 0, 128 * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
  // OPT-IN forms that stage the input tile once for several classes (wgrad_tile_f32.hip: wgrad_polyc_f32_kernel).  SV_WGRAD_POLYC_FUSED=1: all four classes in one launch
  // (168 accumulator registers; profiles/r05_polyc_fused_ab.txt: 1.078 -> 0.949 ms alone on the chip, the 512-image step +2 %); =2: classes {0, 3} and {1, 2} as two launches
  static const int fuse_mode = getenv("SV_WGRAD_POLYC_FUSED") ? atoi(getenv("SV_WGRAD_POLYC_FUSED")) : 0;
  bool fused = false;
  if (!merged && (fuse_mode == 1 || fuse_mode == 2)) {
    WgradArgs cls[8];
    const int masks[2] = {fuse_mode == 1 ? 0xF : 0x9, 0x6}, nl = fuse_mode == 1 ? 1 : 2;
    for (int c = 0; c < 4; ++c)
      for (int i = 0; i < n; ++i) {
        WgradArgs& q = cls[c * n + i];
        svg_polyc_wgrad_args(d, c, &q);
        q.A = x_lo[i]; q.dY = dy[i]; q.dW = pw[i] + pj.coff[c];
        // ONE bias partial per launch (its classes' dY tiles cover their pixels once): block c of dbias' for the launch's lowest class c; the projection sums the blocks
        const bool lead = c == 0 || (fuse_mode == 2 && c == 1);
        q.dbias = (lead && dbias && dbias[i]) ? pw[i] + ndwp + c * 32 : nullptr;
        q.ws = (float*)((char*)slab_ws[i] + c * part); q.ws_bytes = part;
      }
    for (int l = 0; l < nl; ++l) {
      const int rc = svk_wgrad_polyc_f32_multi(cls, n, masks[l], pend, &npend, st);
      if (rc == SV_OK) fused = true;
      else if (rc != SV_E_UNSUPPORTED || l) return rc == SV_E_UNSUPPORTED ? SV_E_STATE : rc;      // (the second pair cannot fail where the first passed: same geometry)
      else break;
    }
  }
  for (int c = 0; c < (merged ? 1 : 4) && !fused; ++c) {
    for (int i = 0; i < n; ++i) {
      if (merged) svg_poly_wgrad_args(d, &a[i]); else svg_polyc_wgrad_args(d, c, &a[i]);
      a[i].A = x_lo[i]; a[i].dY = dy[i];
      a[i].dW = pw[i] + (merged ? 0 : pj.coff[c]);
      a[i].dbias = (dbias && dbias[i]) ? pw[i] + ndwp + (merged ? 0 : c * 32) : nullptr;
      a[i].ws = (float*)((char*)slab_ws[i] + c * part); a[i].ws_bytes = part;
      a[i].defer = pend; a[i].n_defer = &npend;
    }
    const int rc = svk_wgrad_tile_f32_multi(a, n, st);
    if (rc) return c == 0 ? rc : (rc == SV_E_UNSUPPORTED ? SV_E_STATE : rc);    // (a later class cannot fail where class 0 passed: same geometry)
  }
  if (npend) {
    const int rc = svk_wgrad_reduce_all(pend, npend, st);
    if (rc) return rc;
  }
  PolycFrameMulti m;
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    m.x[i] = (const float*)x_lo[k]; m.dy[i] = (const float*)dy[k]; m.slab[i] = pw[k] + ndwp + 128;
    pj.dWp[i] = pw[k]; pj.dbp[i] = pw[k] + ndwp; pj.slab[i] = m.slab[i]; pj.dW[i] = dW[k]; pj.dbias[i] = dbias ? dbias[k] : nullptr;
  }
  pj.K = K; pj.Cin = cin; pj.Cout = d->Cout; pj.merged = merged; pj.groups = 1;
  const dim3 grid(2 * (K - 1), frame_groups(), n);
  const int ldy = svg_gdy(d);
  int rcf;
  if (cin == 64 && d->Cout == 32) rcf = launch_frame<6, 4, 2>(m, grid, d->B, h, w, ldy, st);
  else if (cin == 32 && d->Cout == 32) rcf = launch_frame<6, 2, 2>(m, grid, d->B, h, w, ldy, st);
  else if (cin == 32 && d->Cout <= 16) rcf = launch_frame<6, 2, 1>(m, grid, d->B, h, w, ldy, st);
  else rcf = SV_E_STATE;                                                            // (svg_polyc_wgrad_form admits only these)
  if (rcf) return rcf == SV_E_UNSUPPORTED ? SV_E_STATE : rcf;                       // (the main terms are already enqueued: the form check below keeps this unreachable)
  {
    const int per = (int)(2 * (K - 1) * frame_per(K, cin, d->Cout));
    PolycFrameSum fs;
    for (int i = 0; i < 2; ++i) {
      fs.slab[i] = m.slab[i]; fs.sum[i] = m.slab[i] + (int64_t)FRAME_GROUPS_MAX * per;
      pj.slab[i] = fs.sum[i];
    }
    pj.groups = 1;
    hipLaunchKernelGGL(polyc_frame_sum_kernel, dim3((per / 4 + 255) / 256, n), dim3(256), 0, st, fs, frame_groups(), per);
    SV_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(polyc_wgrad_project_kernel, dim3((K * K * cin * d->Cout + 255) / 256, n), dim3(256), 0, st, pj);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
