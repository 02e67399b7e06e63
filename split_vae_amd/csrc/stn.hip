// SPLIT-SPAIR's spatial transformer (spair/utils.py: STN.call :119-200, STN.bilinear_sampler :202-272,
// STN.get_pixel_value :274-330), forward and the gradient tape.gradient takes through it, fp32.
//
//   forward form (inverse = 0): img [B,H,W,C] (the input image)            -> out [B,B',Ho,Wo,C]   (one glimpse per cell)
//   inverse form (inverse = 1): img [B,B',H,W,C] (the objects' renderings)  -> out [B,B',Ho,Wo,C]   (each pasted on its canvas)
// z_where [B,B',4] = (sx, sy, tx, ty) pre-activations of the Hc x Wc cells (B' = Hc*Wc):
//   sx = .5 sigmoid(z0), sy = .5 sigmoid(z1), tx = .5 tanh(z2) + bias_tx(cell), ty = .5 tanh(z3) + bias_ty(cell)    (:140-143)
//   bias = (2 - r) * i / (n - 1) - (1 - r / 2), r = (2*12)/48                                                        (:96-112)
//   inverse: tx = -tx / (sx + 1e-5), ty = -ty / (sy + 1e-5), sx = 1 / (sx + 1e-5), sy = 1 / (sy + 1e-5)              (:158-162)
// Sampling point of output pixel (oy, ox): xn = sx * gx + tx, yn = sy * gy + ty with gx = linspace(-1, 1, Wo)[ox];
// x = .5 (xn + 1)(W - 1); x0 = floor(x), x1 = x0 + 1, both clipped to [0, W-1] AFTER x1 = x0 + 1, weights from the clipped
// corners (so a point on the last column gets weight 0 from both: the reference's border behaviour, kept).
// obj_bbox_mask [B,B',4] = (bty - sy/2, btx - sx/2, bty + sy/2, btx + sx/2), bt = (t + 1)/2, from the NON-inverted values.
//
// One thread per output pixel, all C channels (C = 3 image / 4 object channels: the gather is HBM- and latency-bound).
// Backward: g_z_where is reduced per (b, cell) in the workgroup (one workgroup per cell: wave shuffles + LDS, fixed order) --
// floor / clip carry no gradient, exactly as tf.floor / tf.clip_by_value at interior points.  g_img (optional: the glimpse
// STN reads the input image, which takes no gradient) is a scatter-add:
//   inverse form, object <= SV_STN_LDS_FLOATS: every (b, cell) workgroup owns its object's gradient -- accumulated in LDS (ds_add_f32) and
//     written out once with plain coalesced stores (the buffer need not be zeroed);
//   otherwise: fp32 atomics into the zeroed global buffer (the 16 cells of an image overlap).
// Taps with zero weight (the 48x48 canvas pixels outside the pasted 32x32 object: both corners clipped) scatter nothing.
#include "common.hip.h"
#include "kernels.h"

namespace {

struct StnCell { float sx, sy, tx, ty, s0, s1, th2, th3, sxr, syr, txr, tyr; };   // transformed, activations, raw (non-inverted)

__device__ __forceinline__ StnCell stn_cell(const float* __restrict__ zw, int cell, int Hc, int Wc, int inverse) {
  StnCell c;
  const float r = (2.0f * 12) / 48;
  const int ci = cell / Wc, cj = cell - ci * Wc;
  const float by = (2.f - r) * (float)ci / (float)(Hc - 1) - (1.f - 0.5f * r);
  const float bx = (2.f - r) * (float)cj / (float)(Wc - 1) - (1.f - 0.5f * r);
  c.s0 = sigmoid_f(zw[0]); c.s1 = sigmoid_f(zw[1]); c.th2 = tanhf(zw[2]); c.th3 = tanhf(zw[3]);
  c.sxr = 0.5f * c.s0; c.syr = 0.5f * c.s1; c.txr = 0.5f * c.th2 + bx; c.tyr = 0.5f * c.th3 + by;
  if (inverse) {
    c.tx = -c.txr / (c.sxr + 1e-5f); c.ty = -c.tyr / (c.syr + 1e-5f);
    c.sx = 1.f / (c.sxr + 1e-5f); c.sy = 1.f / (c.syr + 1e-5f);
  } else { c.sx = c.sxr; c.sy = c.syr; c.tx = c.txr; c.ty = c.tyr; }
  return c;
}

struct StnTap { int x0, x1, y0, y1; float wx0, wx1, wy0, wy1; };   // wx0 = (x1 - x), wx1 = (x - x0) from the CLIPPED corners
__device__ __forceinline__ StnTap stn_tap(float xn, float yn, int H, int W) {
  StnTap t;
  const float x = 0.5f * (xn + 1.0f) * (float)(W - 1), y = 0.5f * (yn + 1.0f) * (float)(H - 1);
  float x0 = floorf(x), y0 = floorf(y), x1 = x0 + 1.f, y1 = y0 + 1.f;
  x0 = fminf(fmaxf(x0, 0.f), (float)(W - 1)); x1 = fminf(fmaxf(x1, 0.f), (float)(W - 1));
  y0 = fminf(fmaxf(y0, 0.f), (float)(H - 1)); y1 = fminf(fmaxf(y1, 0.f), (float)(H - 1));
  t.wx0 = x1 - x; t.wx1 = x - x0; t.wy0 = y1 - y; t.wy1 = y - y0;
  // both corners clipped onto one pixel (a point outside the image, or on its last row / column): the two terms are
  // w * I and -w * I of the SAME pixel with |w| up to the distance from the image -- identically zero; dropping them
  // instead of cancelling them in fp32 (forward sum and the atomics of the backward scatter) is exact and noise-free
  if (x0 == x1) t.wx0 = t.wx1 = 0.f;
  if (y0 == y1) t.wy0 = t.wy1 = 0.f;
  t.x0 = (int)x0; t.x1 = (int)x1; t.y0 = (int)y0; t.y1 = (int)y1;
  return t;
}
__device__ __forceinline__ float lin(int i, int n) { return n > 1 ? -1.f + 2.f * (float)i / (float)(n - 1) : -1.f; }   // np.linspace(-1, 1, n)[i]

#define SV_STN_LDS_FLOATS 8192                                     // 32 KB: an object_size 32 rgb+alpha rendering is 4096 floats

template <bool BWD>
__global__ __launch_bounds__(256) void stn_kernel(const float* __restrict__ img, const float* __restrict__ z_where,
                                                  float* __restrict__ out, float* __restrict__ bbox,
                                                  const float* __restrict__ g_out, float* __restrict__ g_img,
                                                  float* __restrict__ g_z, int Bp, int Hc, int Wc, int H, int W, int C,
                                                  int Ho, int Wo, int inverse, int lds_acc) {
  extern __shared__ float sacc[];                              // BWD, lds_acc: this object's gradient [H*W*C]
  const int cell = blockIdx.x, b = blockIdx.y;                 // one workgroup per (image, cell)
  if (BWD && lds_acc) {
    for (int i = threadIdx.x; i < H * W * C; i += 256) sacc[i] = 0.f;
    __syncthreads();
  }
  const StnCell c = stn_cell(z_where + ((int64_t)b * Bp + cell) * 4, cell, Hc, Wc, inverse);
  const float* ib = img + ((int64_t)b * (inverse ? Bp : 1) + (inverse ? cell : 0)) * H * W * C;
  float* gib = (BWD && g_img) ? g_img + ((int64_t)b * (inverse ? Bp : 1) + (inverse ? cell : 0)) * H * W * C : nullptr;
  const int64_t ob = ((int64_t)b * Bp + cell) * Ho * Wo * C;
  if (!BWD && bbox && threadIdx.x == 0) {
    float* bb = bbox + ((int64_t)b * Bp + cell) * 4;
    const float bty = (c.tyr + 1.f) * 0.5f, btx = (c.txr + 1.f) * 0.5f;
    bb[0] = bty - c.syr * 0.5f; bb[1] = btx - c.sxr * 0.5f; bb[2] = bty + c.syr * 0.5f; bb[3] = btx + c.sxr * 0.5f;
  }
  float gsx = 0.f, gsy = 0.f, gtx = 0.f, gty = 0.f;            // d loss / d (transformed sx, sy, tx, ty)
  for (int p = threadIdx.x; p < Ho * Wo; p += 256) {
    const int oy = p / Wo, ox = p - oy * Wo;
    const float gx = lin(ox, Wo), gy = lin(oy, Ho);
    const StnTap t = stn_tap(c.sx * gx + c.tx, c.sy * gy + c.ty, H, W);
    const float wa = t.wx0 * t.wy0, wb = t.wx0 * t.wy1, wc = t.wx1 * t.wy0, wd = t.wx1 * t.wy1;
    const float* pa = ib + ((int64_t)t.y0 * W + t.x0) * C;
    const float* pb = ib + ((int64_t)t.y1 * W + t.x0) * C;
    const float* pc = ib + ((int64_t)t.y0 * W + t.x1) * C;
    const float* pd = ib + ((int64_t)t.y1 * W + t.x1) * C;
    if (!BWD) {
      for (int k = 0; k < C; ++k) out[ob + (int64_t)p * C + k] = ((wa * pa[k] + wb * pb[k]) + wc * pc[k]) + wd * pd[k];   // tf.add_n order
    } else {
      float dx = 0.f, dy = 0.f;                                // d loss / d x, d y (pixel coordinates)
      for (int k = 0; k < C; ++k) {
        const float g = g_out[ob + (int64_t)p * C + k];
        const float Ia = pa[k], Ib = pb[k], Ic = pc[k], Id = pd[k];
        dx += g * (t.wy0 * (Ic - Ia) + t.wy1 * (Id - Ib));
        dy += g * (t.wx0 * (Ib - Ia) + t.wx1 * (Id - Ic));
        if (!gib) continue;
        if (lds_acc) {
          if (wa != 0.f) atomicAdd(sacc + (t.y0 * W + t.x0) * C + k, g * wa);
          if (wb != 0.f) atomicAdd(sacc + (t.y1 * W + t.x0) * C + k, g * wb);
          if (wc != 0.f) atomicAdd(sacc + (t.y0 * W + t.x1) * C + k, g * wc);
          if (wd != 0.f) atomicAdd(sacc + (t.y1 * W + t.x1) * C + k, g * wd);
        } else {
          if (wa != 0.f) atomicAdd(gib + ((int64_t)t.y0 * W + t.x0) * C + k, g * wa);
          if (wb != 0.f) atomicAdd(gib + ((int64_t)t.y1 * W + t.x0) * C + k, g * wb);
          if (wc != 0.f) atomicAdd(gib + ((int64_t)t.y0 * W + t.x1) * C + k, g * wc);
          if (wd != 0.f) atomicAdd(gib + ((int64_t)t.y1 * W + t.x1) * C + k, g * wd);
        }
      }
      const float dxn = dx * 0.5f * (float)(W - 1), dyn = dy * 0.5f * (float)(H - 1);
      gsx += dxn * gx; gtx += dxn; gsy += dyn * gy; gty += dyn;
    }
  }
  if (BWD && lds_acc) {
    __syncthreads();
    for (int i = threadIdx.x; i < H * W * C; i += 256) gib[i] = sacc[i];
  }
  if (BWD) {
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    gsx = wave_sum(gsx); gsy = wave_sum(gsy); gtx = wave_sum(gtx); gty = wave_sum(gty);
    if (lane == 0) { red[wave][0] = gsx; red[wave][1] = gsy; red[wave][2] = gtx; red[wave][3] = gty; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float s[4];
      for (int k = 0; k < 4; ++k) s[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
      float dsxr = s[0], dsyr = s[1], dtxr = s[2], dtyr = s[3];   // -> the raw (non-inverted) sx, sy, tx, ty
      if (inverse) {
        const float ex = c.sxr + 1e-5f, ey = c.syr + 1e-5f;
        // sx' = 1/ex, tx' = -txr/ex:  d/dsxr = -sx'^2 * gsx' + txr/ex^2 * gtx' ; d/dtxr = -gtx'/ex
        dsxr = -s[0] / (ex * ex) + s[2] * c.txr / (ex * ex); dtxr = -s[2] / ex;
        dsyr = -s[1] / (ey * ey) + s[3] * c.tyr / (ey * ey); dtyr = -s[3] / ey;
      }
      float* gz = g_z + ((int64_t)b * Bp + cell) * 4;
      gz[0] = dsxr * 0.5f * c.s0 * (1.f - c.s0);
      gz[1] = dsyr * 0.5f * c.s1 * (1.f - c.s1);
      gz[2] = dtxr * 0.5f * (1.f - c.th2 * c.th2);
      gz[3] = dtyr * 0.5f * (1.f - c.th3 * c.th3);
    }
  }
}

}  // namespace

static int stn_check(int B, int Hc, int Wc, int H, int W, int C, int Ho, int Wo) {
  if (B < 1 || Hc < 2 || Wc < 2 || H < 2 || W < 2 || C < 1 || Ho < 1 || Wo < 1) return SV_E_BADARG;
  if ((int64_t)B * Hc * Wc * Ho * Wo * C >= (1LL << 40)) return SV_E_UNSUPPORTED;
  return SV_OK;
}

extern "C" int sv_stn_sample_fwd(const float* img, const float* z_where, float* out, float* bbox, int32_t B, int32_t Hc,
                                 int32_t Wc, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t inverse,
                                 void* stream) {
  if (!img || !z_where || !out) return SV_E_BADARG;
  const int rc = stn_check(B, Hc, Wc, H, W, C, Ho, Wo);
  if (rc) return rc;
  hipLaunchKernelGGL((stn_kernel<false>), dim3(Hc * Wc, B), dim3(256), 0, (hipStream_t)stream, img, z_where, out, bbox,
                     (const float*)nullptr, (float*)nullptr, (float*)nullptr, Hc * Wc, Hc, Wc, H, W, C, Ho, Wo, inverse ? 1 : 0, 0);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_stn_bwd_overwrites(int32_t H, int32_t W, int32_t C, int32_t inverse) {
  return inverse && (int64_t)H * W * C <= SV_STN_LDS_FLOATS ? 1 : 0;
}

extern "C" int sv_stn_sample_bwd(const float* img, const float* z_where, const float* g_out, float* g_img, float* g_z_where,
                                 int32_t B, int32_t Hc, int32_t Wc, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                 int32_t inverse, void* stream) {
  if (!img || !z_where || !g_out || !g_z_where) return SV_E_BADARG;
  const int rc = stn_check(B, Hc, Wc, H, W, C, Ho, Wo);
  if (rc) return rc;
  const int lds_acc = g_img && sv_stn_bwd_overwrites(H, W, C, inverse);
  hipLaunchKernelGGL((stn_kernel<true>), dim3(Hc * Wc, B), dim3(256), lds_acc ? (size_t)H * W * C * sizeof(float) : 0, (hipStream_t)stream,
                     img, z_where, (float*)nullptr, (float*)nullptr, g_out, g_img, g_z_where, Hc * Wc, Hc, Wc, H, W, C, Ho, Wo,
                     inverse ? 1 : 0, lds_acc);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
