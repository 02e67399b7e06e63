// Host-side geometry helpers shared by conv_api.hip and lgvae_plan.hip.
#pragma once
#include <stdlib.h>
#include <string.h>
#include "../../include/splitvae.h"
#include "kernels.h"
#include "common.hip.h"

static inline int svg_epp(const sv_conv_desc* d) { return d->dtype == SV_BF16 ? 8 : 4; }
static inline int svg_r8(int v) { return (v + 7) / 8 * 8; }
static inline int svg_cin_pad(const sv_conv_desc* d) { return svg_r8(d->Cin); }
// channels per pixel of the GRADIENT tensor dY (the forward y may be an unpadded fp32 head)
static inline int svg_gdy(const sv_conv_desc* d) { return svg_r8(d->Cout); }
static inline int svg_oh(const sv_conv_desc* d) { return (d->H + d->stride - 1) / d->stride; }
static inline int svg_ow(const sv_conv_desc* d) { return (d->W + d->stride - 1) / d->stride; }
static inline void svg_pads(const sv_conv_desc* d, int* pt, int* pl) {
  int ph = (svg_oh(d) - 1) * d->stride + d->KH - d->H;
  int pw = (svg_ow(d) - 1) * d->stride + d->KW - d->W;
  if (ph < 0) ph = 0;
  if (pw < 0) pw = 0;
  *pt = ph / 2;
  *pl = pw / 2;
}
// x-pixel packing of a thin-output stride-1 conv (the 6-channel decoder head d5).  With N = Cout <= 8
// an MFMA tile is 10/16 padding and every A fragment read from LDS feeds a single MFMA.  Written
// for PAIRS of output pixels (y, 2X+px) the same conv has x-stride 2, KW+1 taps in x and 2*8 = 16
// output columns n = px*8 + co with W'[ky][tx][ci][n] = W[ky][tx-px][ci][co] (zero outside the kernel):
// (KW+1)/(2*KW) = 0.58x the MFMAs, A/B fragment reads and K steps for the same LDS tile per output.
// Its dY is the ordinary [B,H,W,8] gradient viewed as [B,H,W/2,16].  Forward (tile_conv.hip, depth-to-
// space stores) and wgrad (wgrad_tile.hip / wgrad_tile_f32.hip, folded reduce) use it, at both precisions; the dgrad keeps the direct form.
static inline int svg_packx(const sv_conv_desc* d) {
  static const bool off = getenv("SV_NO_PACKX") != nullptr;
  return !off && d->stride == 1 && d->Cout <= 8 && !(d->Cout & 1) && d->y_f32 &&
         d->ldy == d->Cout && d->KH * (d->KW + 1) <= SV_MAX_TAPS && d->W >= 32 && d->H >= 16 &&
         svg_cin_pad(d) >= 16 && svg_cin_pad(d) <= 64;
}
// POLYPHASE form of the decoder head (2x bilinear upsample -> 6x6 SAME conv, Cout <= 8: vae/model.py:163-167 + d5).  The resize
// is linear, so conv(U x) is a conv over the LOW-RES tensor: output pixel (2i+py, 2j+px) reads low-res rows i-2..i+2 and
// columns j-2..j+2 through the composite weights W'[py,px][ty,tx] = sum_{ky,kx} Cy[py][ky][ty] Cx[px][kx][tx] w[ky,kx]
// (Cy = the .25/.75 half-pixel blend coefficients): 25 taps x Cin instead of 42 x-packed taps, 32 columns (4 parities x
// 8) instead of 16, a quarter of the pixels to stage and no blend arithmetic in the staging loop.  With the low-res input
// edge-clamped (= the resize's own clamp) this equals the conv over the REPLICATE-padded upsampled image; the reference
// zero-pads it, and the difference -- taps of the 5 hi-res border rows / columns that leave the image -- is 1-D convs
// along the edges, subtracted by svk_poly_fix (poly_fix.hip).  bf16 only: the fp32 parity path keeps the direct form.
// (fp32 since round 5: the composite weights are formed in fp32 from the fp32 masters -- the result differs from the direct form by fp32 rounding only)
static inline int svg_poly(const sv_conv_desc* d) {
  static const bool off = getenv("SV_NO_POLY") != nullptr;           // A/B: the fused-upsample x-packed conv
  static const bool off32 = getenv("SV_NO_POLY_F32") != nullptr;     // A/B: fp32 keeps the x-packed form
  return !off && (d->dtype == SV_BF16 || !off32) && svg_packx(d) && d->ups_in && d->KH == 6 && d->KW == 6 && d->Cin == svg_cin_pad(d) && d->Cin == 32 &&
         d->H >= 16 && d->W >= 16 && !(d->H & (d->H - 1)) && !(d->W & (d->W - 1)) && d->act == SV_ACT_NONE;
}
// PER-CLASS POLYPHASE form of the wide upsample -> conv layers (d4: UpSampling2D -> Conv2D(32, 6); d3: -> Conv2D(64, 4); vae/model.py:154-155,
// :163-165).  The same identity as svg_poly, but every output parity class c = (py, px) is its OWN conv over the low-res tensor with only the
// low-res offsets that parity touches: k = 6 (pad 2): parity 0 reads offsets -2..2, parity 1 reads -1..2; k = 4 (pad 1): -1..1 and -1..2.
//   y[2i+py, 2j+px] = sum_{ty in T(py), tx in T(px)} W'_c[ty,tx] . x~[i+ty, j+tx]  -  border terms,      x~ = the edge-clamped low-res tensor
// (5+4)^2 = 81 tap-class products instead of 4 x 36 = 144 hi-res taps per low-res pixel (0.5625 of the direct form's MFMAs; k = 4: 49 / 64), no blend
// arithmetic, a quarter of the pixels to stage.  Four problems per network in one launch (like the parity classes of a stride-2 input gradient), each
// scattering to its sub-pixel (OS = 2, ooy / oox); the border terms (poly_fix.hip: polyc_fix_kernel) are added by the epilogue BEFORE the activation.
// tests/test_polyphase_math.py pins the algebra for both kernel sizes.
static inline int svg_polyc(const sv_conv_desc* d) {
  static const bool off = getenv("SV_NO_POLYC") != nullptr;
  // kernel sizes that take the form.  Measured (fp32, 2 x 512 images): d4 (k 6) forward 1.182 -> 0.88 ms; d3 (k 4: 49 of 64 tap products, four launches' worth of
  // staging and a border pass) 0.549 -> 0.567 ms: k 4 stays on the fused-resize direct form (SV_POLYC_K=64 enables it)
  const char* ks = getenv("SV_POLYC_K") ? getenv("SV_POLYC_K") : "6";       // (read per call: tests/test_gpu_kernels.py switches it for the k = 4 case)
  if (off || d->dtype != SV_F32 || !d->ups_in || d->stride != 1 || d->KH != d->KW || (d->KH != 6 && d->KH != 4)) return 0;
  if (!strchr(ks, d->KH == 6 ? '6' : '4')) return 0;
  if (d->y_f32 || d->ldy != d->Cout || d->ldx != d->Cin) return 0;
  // (the border kernel's instantiations, poly_fix.hip)
  if (!((d->KH == 6 && d->Cout == 32 && (d->Cin == 64 || d->Cin == 32)) || (d->KH == 4 && d->Cout == 64 && d->Cin == 128))) return 0;
  return d->H >= 16 && d->W >= 16 && !(d->H & (d->H - 1)) && !(d->W & (d->W - 1));
}
// blend coefficient of hi-res tap k of output parity p on the low-res offset t (kernel size K, SAME pad before = (K-1)/2): the hi-res offset from row 2i is
// h = p + k - pad; even h = 2m reads low-res rows i+m-1 (.25) and i+m (.75), odd h = 2m+1 rows i+m (.75) and i+m+1 (.25)
static inline __host__ __device__ float svg_pcoef(int p, int k, int t, int pad) {
  const int h = p + k - pad, m = h >> 1;                              // arithmetic shift = floor
  if (h & 1) return t == m ? 0.75f : t == m + 1 ? 0.25f : 0.f;
  return t == m - 1 ? 0.25f : t == m ? 0.75f : 0.f;
}
// low-res offsets parity p touches: [*t_lo, *t_lo + n)
static inline __host__ __device__ int svg_polyc_taps(int K, int p, int* t_lo) {
  const int pad = (K - 1) / 2;
  const int h0 = p - pad, h1 = p + K - 1 - pad;                       // first / last hi-res offset
  const int lo = (h0 & 1) ? (h0 >> 1) : (h0 >> 1) - 1, hi = (h1 & 1) ? (h1 >> 1) + 1 : (h1 >> 1);
  *t_lo = lo;
  return hi - lo + 1;
}
// border classes of one direction: hi-res rows 0 .. pad-1 (taps k < pad - Y leave the image) and 2h-nb .. 2h-1, nb = K-1-pad (taps k > 2h-1-Y+pad): K-1 classes
static inline __host__ __device__ int svg_polyc_nclass(int K) { return K - 1; }
// tap k leaves the (zero-padded) image at border class c (0 .. K-2) of a K-tap kernel
static inline __host__ __device__ bool svg_polyc_excl(int K, int c, int k) {
  const int pad = (K - 1) / 2;
  if (c < pad) return k < pad - c;                                    // hi-res row Y = c
  const int below = K - 1 - pad - (c - pad);                          // rows from Y to the last row, inclusive (nb .. 1)
  return k > below - 1 + pad;                                         // Y + k - pad > 2h - 1  <=>  k > (2h - 1 - Y) + pad, 2h - 1 - Y = below - 1
}
#define SV_POLY_FIX_ELEMS(cin) (10 * 6 * 16 * (cin))                 // [10 border classes][6 taps][16 columns][Cin]
// SPACE-TO-DEPTH form of the first encoder layer at fp32 (e1: Conv2D(32, 6, strides 2) over RGB, vae/model.py:36).  The padded 8-channel pixels make 62 % of every
// fp32 MFMA zeros (3 of 8 channels: K = 36 x 8 = 288 for 108 real products).  Input row 2 oy + ky - 2 = 2 (oy + t) + py with t in {-1, 0, 1}: over the space-to-depth view
// [B, H/2, W/2, (py, px, c)] the layer is a 3 x 3 stride-1 SAME conv with 12 (padded to 16) channels: K = 9 x 16 = 144, half the MFMAs, and the stride-1 tile path.  The
// view is formed while the tile is staged (tile_stage.hip.h: stage_tile_s2d3) from the unchanged 8-channel tensor; weights / gradients map by dw_index's s2d3 rule.
static inline int svg_s2d3(const sv_conv_desc* d) {
  static const bool off = getenv("SV_NO_S2D3") != nullptr;
  return !off && d->dtype == SV_F32 && d->Cin == 3 && d->ldx == 8 && d->KH == 6 && d->KW == 6 && d->stride == 2 && !d->ups_in && !d->y_f32 &&
         d->H >= 16 && d->W >= 16 && !(d->H & (d->H - 1)) && !(d->W & (d->W - 1)) && d->Cout % 16 == 0;
}
// N tile selection of the tap GEMM: 0: 128, 1: 64, 2: 32, 3: 16 columns
static inline int svg_pick_cfg(int N) {
  if (N % 128 == 0) return 0;
  if (N % 64 == 0) return 1;
  if (N % 32 == 0) return 2;
  return 3;
}
// Position of filter tap (kh, kw) in the K order of the forward weight image / tap tables.  Stride-1 layers run
// x-major, y-minor (t = kw*KH + kh): the KH taps of one filter column are consecutive, which is what the row-window
// reuse of the tile kernel (tile_conv.hip, YR) walks.  Stride 2 keeps the HWIO order.
static inline int svg_fwd_tap(const sv_conv_desc* d, int kh, int kw) {
  return d->stride == 1 ? kw * d->KH + kh : kh * d->KW + kw;
}
// split K across workgroups when the M x N tile grid cannot fill 256 CUs.  *cfg (optional) is the
// tap-GEMM tile: a split-K problem on 128-column tiles drops to 64 columns (twice the workgroups for
// the same number of atomic passes; prepared weight rows are padded to 128, so either tile fits).
static inline int svg_choose_splitk(int M, int N, int nk, int* cfg_io = nullptr) {
  static const int BMt[4] = {128, 128, 256, 256}, BNt[4] = {128, 64, 32, 16};
  static const bool narrow = getenv("SV_SPLITK_NO_NARROW") == nullptr;
  int cfg = svg_pick_cfg(N);
  int tiles = ((M + BMt[cfg] - 1) / BMt[cfg]) * ((N + BNt[cfg] - 1) / BNt[cfg]);
  if (tiles >= 128) return 1;
  if (sv_deterministic()) return 1;        // one workgroup per output tile: a single add per element, whatever the launch order
  static const bool small = getenv("SV_SPLITK_NO_SMALL") == nullptr;   // 64 x 32 tiles (tap-GEMM cfg 4): +0.8 % on the step; knob restores 128 x 64
  if (small && cfg_io && (N % 32) == 0) {
    static const int tgt_small = getenv("SV_SPLITK_WGS") ? atoi(getenv("SV_SPLITK_WGS")) : 512;
    const int t2 = ((M + 63) / 64) * (N / 32);
    *cfg_io = 4;
    int s2 = (tgt_small + t2 - 1) / t2;
    if (s2 > nk / 2) s2 = nk / 2;
    return s2 < 1 ? 1 : s2;
  }
  if (cfg == 0 && cfg_io && narrow) { cfg = 1; tiles *= 2; }
  if (cfg_io) *cfg_io = cfg;
  static const int target = getenv("SV_SPLITK_WGS") ? atoi(getenv("SV_SPLITK_WGS")) : 128;   // measured best of 64/128/256/512 on the heads and d1
  const int tgt = cfg_io && narrow && cfg == 1 && svg_pick_cfg(N) == 0 ? 2 * target : target;
  int s = (tgt + tiles - 1) / tiles;
  if (s > nk / 2) s = nk / 2;
  return s < 1 ? 1 : s;
}

// blocks of 256 threads for one weight-preparation job (dense forward images use 32x32 LDS tiles)
#ifdef SV_PREP_UNITS_OVERRIDE
#define SV_PREP_UNITS SV_PREP_UNITS_OVERRIDE
#else
#define SV_PREP_UNITS 2      // re-measured (round 2): 8 -> 2 takes the weight preparation from 39 to 29 us (more workgroups in flight)
#endif
static inline int svg_prep_nblocks(const PrepJob* j) {
  int64_t units;
  if (j->packx_kw) units = ((int64_t)j->rows * j->ntaps * j->inner + 255) / 256;
  else if (!j->transpose) units = (int64_t)j->ntaps * ((j->rows + 63) / 64) * ((j->inner + 63) / 64);
  else if (j->transpose && !(j->inner & 7) && !(j->Cout & 7) && !(j->inner_off & 7) && !(j->inner_ld & 7))
    units = ((int64_t)j->rows * j->ntaps * (j->inner >> 3) + 255) / 256;          // 8 channels per thread
  else units = ((int64_t)j->rows * j->ntaps * j->inner + 255) / 256;
  return (int)((units + SV_PREP_UNITS - 1) / SV_PREP_UNITS);
}

int svg_check(const sv_conv_desc* d);
void svg_fwd_args(const sv_conv_desc* d, TapGemmArgs* a);
int svg_dgrad_classes(const sv_conv_desc* d);
void svg_dgrad_args(const sv_conv_desc* d, int cls, TapGemmArgs* a, uint8_t srctap[SV_MAX_TAPS]);
bool svg_dgrad_merged_args(const sv_conv_desc* d, const int64_t* class_off, TapGemmArgs* a);   // all 4 classes as one problem
void svg_wgrad_args(const sv_conv_desc* d, WgradArgs* a);
void svg_poly_wgrad_args(const sv_conv_desc* d, WgradArgs* a);      // main term of the polyphase weight gradient (svg_poly layers)
#define SV_POLY_WGRAD_NWG 256
void svg_wgrad_set_msplit(WgradArgs* a, int cfg, int dtype, int target_wgs);
void svg_prep_job_fwd(const sv_conv_desc* d, PrepJob* j);
void svg_prep_job_dgrad(const sv_conv_desc* d, int cls, PrepJob* j);
void svg_prep_job_polyfix(const sv_conv_desc* d, PrepJob* j);     // second forward job of a svg_poly layer (follows the main image)
// per-class polyphase (svg_polyc): forward problem of class cls = py*2 + px; its composite-weight image; the border-class image; element counts
void svg_polyc_fwd_args(const sv_conv_desc* d, int cls, TapGemmArgs* a);
void svg_prep_job_polyc(const sv_conv_desc* d, int cls, PrepJob* j);
void svg_prep_job_polyc_fix(const sv_conv_desc* d, PrepJob* j);
int64_t svg_polyc_class_elems(const sv_conv_desc* d, int cls);
int64_t svg_polyc_fix_elems(const sv_conv_desc* d);
int64_t svg_polyc_fix_ws_bytes(const sv_conv_desc* d);
// POLYPHASE INPUT GRADIENT of an upsample -> conv layer, delivered at the LOW-RES tensor (ResizeBilinearGrad o Conv2DBackpropInput o ReluGrad in one pass;
// polyd_dgrad.hip, tests/test_polyphase_math.py).  The transposed per-class polyphase conv reads dy[2(i'-ty)+py, 2(j'-tx)+px] for (class, offset) pairs: every
// hi-res offset dh = p - 2t in [-R, R] (R = 4 for k = 6, 3 for k = 4) belongs to exactly one (parity p, low-res offset t), so the main term is ONE stride-2
// conv with (2R+1)^2 taps over the hi-res dY:   dx[i',j'] = sum_{dyh,dxh} V[dyh,dxh]^T dy[2i'+dyh, 2j'+dxh],   V[dyh,dxh] = W'_(py,px)[ty,tx]
// -- 81 tap products per low-res pixel instead of 4 x 36 -- with dy zero outside the image.  That is the gradient of the conv over an UNPADDED, unclamped
// resize; the reference zero-pads the upsampled image (no gradient flows through the pad rows) and the resize clamps at the edge (the .25 weight of the
// missing neighbour folds onto the edge row).  Both corrections land on the first / last low-res row and column only:
//   rows:    dx[0, j']   += .25 sum_{q, dxh} (Vx[k+(q)][dxh] - Vx[k-(q)][dxh])^T dy[q, 2j'+dxh]          k+(q) = pad - q, k-(q) = pad - 1 - q   (q <= pad)
//            dx[h-1, j'] += .25 sum_{q, dxh} (...)^T dy[2h-1-q, 2j'+dxh]                                k+(q) = pad + q, k-(q) = pad + 1 + q   (q <= K-1-pad)
//   (Vx[ky][dxh] = sum_kx cx(px,kx,tx) w[ky,kx]: the x-only composite; indices outside the kernel drop out), the same for columns, and the four corner
//   pixels += .0625 sum_{qr,qs} (w[k+r,k+s] - w[k+r,k-s] - w[k-r,k+s] + w[k-r,k-s])^T dy[row qr, column qs].
// The main term runs on the tile kernel (S = 2); the edge terms come from polyd_edge_kernel through the epilogue's border-term path (TapGemmArgs::fix).
static inline int svg_polyd(const sv_conv_desc* d) {
  static const bool off = getenv("SV_NO_POLYD") != nullptr;
  static const bool bf = getenv("SV_POLYD_BF16") && atoi(getenv("SV_POLYD_BF16")) != 0;     // bf16: opt-in (A/B against the row-ring kernel's fused adjoint)
  if (off || (d->dtype != SV_F32 && !bf) || !d->ups_in || d->stride != 1 || d->KH != 6 || d->KW != 6 || d->ldx != d->Cin) return 0;
  if (d->H < 32 || d->W < 32 || (d->H & (d->H - 1)) || (d->W & (d->W - 1))) return 0;      // (the edge kernel works on 16-pixel fragments of the low-res lines)
  // (the edge kernel's instantiations: d4 = 64 -> 32, its 32-channel variant, the head 32 -> 6)
  return (d->Cout == 32 && (d->Cin == 64 || d->Cin == 32)) || (d->Cout <= 8 && d->Cin == 32);
}
// hi-res offsets of the stride-2 form: [-R, R]; k = 6: R = 4 (81 taps), k = 4: R = 3
static inline __host__ __device__ int svg_polyd_radius(int K) {
  int r = 0;
  for (int p = 0; p < 2; ++p) {
    int t0;
    const int n = svg_polyc_taps(K, p, &t0), hi = p - 2 * t0, lo = p - 2 * (t0 + n - 1);
    r = hi > r ? hi : r;
    r = -lo > r ? -lo : r;
  }
  return r;
}
// (parity, low-res offset) of hi-res offset dh: p = dh mod 2, t = (p - dh) / 2; false when that parity has no such offset
static inline __host__ __device__ bool svg_polyd_hitap(int K, int dh, int* p, int* t) {
  const int pp = dh & 1, tt = (pp - dh) / 2;
  int t0;
  const int n = svg_polyc_taps(K, pp, &t0);
  *p = pp; *t = tt;
  return tt >= t0 && tt < t0 + n;
}
// kernel rows k+ / k- of distance q from a low (first row / column) or high (last) edge; -1: outside the kernel
static inline __host__ __device__ void svg_polyd_kpm(int K, int hi_edge, int q, int* kp, int* km) {
  const int pad = (K - 1) / 2;
  int a = hi_edge ? pad + q : pad - q, b = hi_edge ? pad + 1 + q : pad - 1 - q;
  *kp = (a >= 0 && a < K) ? a : -1;
  *km = (b >= 0 && b < K) ? b : -1;
}
static inline int svg_polyd_cop(const sv_conv_desc* d) { const int g = svg_gdy(d), m = d->dtype == SV_BF16 ? 32 : 16; return g < m ? m : g; }   // dY channels per edge operand (padded to one MFMA group)
void svg_polyd_args(const sv_conv_desc* d, TapGemmArgs* a);
void svg_prep_job_polyd(const sv_conv_desc* d, int which, PrepJob* j);          // which: 0 main, 1 edges, 2 corners
int64_t svg_polyd_elems(const sv_conv_desc* d, int which);                      // (128-element aligned)
int64_t svg_polyd_ws_bytes(const sv_conv_desc* d);                              // row terms [B][2][w][Cin] + column terms [B][h][2][Cin], fp32
int svk_polyd_dgrad_multi(const sv_conv_desc* d, int n, const void* const* dy, const void* const* w_polyd, const void* const* mask_lo, void* const* dx_lo,
                          void* const* edgews, hipStream_t st);
// polyphase weight gradient at fp32 (polyc_wgrad.hip): 0 none, 1 per class (svg_polyc), 2 merged head (svg_poly); workspace floats per problem; class problem
int svg_polyc_wgrad_form(const sv_conv_desc* d);
int64_t svk_polyc_wgrad_ws_floats(const sv_conv_desc* d);
void svg_polyc_wgrad_args(const sv_conv_desc* d, int cls, WgradArgs* a);
int svk_polyc_wgrad_multi(const sv_conv_desc* d, int n, const void* const* x_lo, const void* const* dy, float* const* dW, float* const* dbias,
                          float* const* slab_ws, int64_t slab_bytes, float* const* pw, hipStream_t st);
// n <= 2 svg_polyc layers of one geometry (the twin networks): border kernel + all class problems in one launch (conv_api.hip)
bool svk_polyc_fwd_plannable(const sv_conv_desc* d);
int svk_polyc_fwd_multi(const sv_conv_desc* d, int n, const void* const* x, const void* const* w_fwd, const float* const* bias, void* const* y,
                        void* const* fixws, hipStream_t st);
int64_t svg_wprep_elems_class(const sv_conv_desc* d, int for_dgrad, int cls);
