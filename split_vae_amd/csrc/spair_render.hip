// SPLIT-SPAIR's Renderer (spair/spair.py:534-579): the depth- and presence-weighted composite of the B' = Hc*Wc objects'
// canvases over the background, forward and the gradient tape.gradient takes through it, fp32.
//
//   a_k = clip(alpha_k, 1e-8, 1)          i_k = clip(rgb_k (+ noise_k), 0, 1)          s_k = sigmoid(-z_depth_k) + 0.5
//   t_k = z_pres_k a_k (transparency)     w_k = t_k s_k (importance)
//   N = sum_k w_k    U = sum_k w_k i_k    T = sum_k t_k w_k      out = A U/(N+1e-8) + (1 - A) bg,   A = T/(N+1e-8)
// training = 0: z_pres = max(round(sigmoid(z_pres_logits)), 1e-8), no noise (the reference's test-time rendering).
// obj [B,B',H,W,C+1] (the inverse STN's output), bg [B,H,W,C], z_* [B,B'] -> out [B,H,W,C].
//
// A workgroup per 256-pixel chunk of an image (grid = chunks x B: 288 workgroups at batch 32 instead of 32), a thread per pixel
// looping the B' <= 16 objects (their 4-float records are strided by the canvas size: a latency-bound gather of B'(C+1)
// floats per pixel).  Backward: the same pass recomputes the forward sums, writes g_obj / g_bg per pixel (clip gates as
// tf.clip_by_value: gradient inside [min, max] only) and reduces g_z_pres / g_z_depth per object over the chunk in a fixed
// order (lane shuffles, then the four waves through LDS); the chunks' partial sums go to a workspace that a second tiny kernel adds
// in chunk order (deterministic).  Without a workspace (sv_spair_render_bwd) one workgroup walks the whole image.
#include "common.hip.h"
#include "kernels.h"

namespace {
constexpr int MAXBP = 16, MAXC = 4;

template <bool BWD>
__global__ __launch_bounds__(256) void spair_render_kernel(const float* __restrict__ obj, const float* __restrict__ bg,
                                                           const float* __restrict__ z_depth, const float* __restrict__ z_pres,
                                                           const float* __restrict__ z_logits, const float* __restrict__ noise,
                                                           float* __restrict__ out, const float* __restrict__ g_out,
                                                           float* __restrict__ g_obj, float* __restrict__ g_bg,
                                                           float* __restrict__ g_zp, float* __restrict__ g_zd, int Bp, int HW,
                                                           int C, int training, float* __restrict__ part) {
  const int b = blockIdx.y;
  __shared__ float s_zp[MAXBP], s_s[MAXBP], s_ds[MAXBP];
  __shared__ float red[4][2 * MAXBP];
  if (threadIdx.x < Bp) {
    const int k = threadIdx.x;
    float zp = z_pres ? z_pres[(int64_t)b * Bp + k] : 0.f;
    if (!training) zp = fmaxf(rintf(sigmoid_f(z_logits[(int64_t)b * Bp + k])), 1e-8f);    // tf.round = half-to-even = rintf
    const float sg = sigmoid_f(-z_depth[(int64_t)b * Bp + k]);
    s_zp[k] = zp; s_s[k] = sg + 0.5f; s_ds[k] = -sg * (1.f - sg);
  }
  __syncthreads();
  const int C1 = C + 1;
  const int64_t cell_stride = (int64_t)HW * C1;
  const float* ob = obj + (int64_t)b * Bp * cell_stride;
  float azp[MAXBP], azd[MAXBP];
#pragma unroll
  for (int k = 0; k < MAXBP; ++k) azp[k] = azd[k] = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float N = 0.f, T = 0.f, U[MAXC] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Bp; ++k) {
      const float* q = ob + k * cell_stride + (int64_t)p * C1;
      const float a = fminf(fmaxf(q[C], 1e-8f), 1.f);
      const float t = s_zp[k] * a, w = t * s_s[k];
      N += w; T += t * w;
      for (int c = 0; c < C; ++c) {
        float v = q[c];
        if (noise) v += noise[((int64_t)b * Bp + k) * HW * C + (int64_t)p * C + c];
        U[c] += w * fminf(fmaxf(v, 0.f), 1.f);
      }
    }
    const float D = N + 1e-8f, A = T / D;
    const float* bp = bg + ((int64_t)b * HW + p) * C;
    if (!BWD) {
      for (int c = 0; c < C; ++c) out[((int64_t)b * HW + p) * C + c] = A * (U[c] / D) + (1.f - A) * bp[c];
      continue;
    }
    const float* gp = g_out + ((int64_t)b * HW + p) * C;
    float dA = 0.f, dU[MAXC], dD = 0.f;
    for (int c = 0; c < C; ++c) {
      const float cn = U[c] / D;
      g_bg[((int64_t)b * HW + p) * C + c] = gp[c] * (1.f - A);
      dA += gp[c] * (cn - bp[c]);
      dU[c] = gp[c] * A / D;
      dD -= gp[c] * A * U[c] / (D * D);
    }
    const float dT = dA / D;
    dD -= dA * T / (D * D);
#pragma unroll
    for (int k = 0; k < MAXBP; ++k) {                     // (static indices: azp / azd stay in registers)
      if (k >= Bp) continue;
      const float* q = ob + k * cell_stride + (int64_t)p * C1;
      float* gq = g_obj + (int64_t)b * Bp * cell_stride + k * cell_stride + (int64_t)p * C1;
      const float ar = q[C], a = fminf(fmaxf(ar, 1e-8f), 1.f);
      const float t = s_zp[k] * a, w = t * s_s[k];
      float dw = dD + dT * t;
      for (int c = 0; c < C; ++c) {
        float v = q[c];
        if (noise) v += noise[((int64_t)b * Bp + k) * HW * C + (int64_t)p * C + c];
        const float i = fminf(fmaxf(v, 0.f), 1.f);
        dw += dU[c] * i;
        gq[c] = (v >= 0.f && v <= 1.f) ? dU[c] * w : 0.f;
      }
      const float dt = dT * w + dw * s_s[k];
      gq[C] = (ar >= 1e-8f && ar <= 1.f) ? dt * s_zp[k] : 0.f;
      azp[k] += dt * a;
      azd[k] += dw * t * s_ds[k];
    }
  }
  if (BWD) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < MAXBP; ++k) {
      const float a = wave_sum(azp[k]), d = wave_sum(azd[k]);
      if (lane == 0) { red[wave][k] = a; red[wave][MAXBP + k] = d; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * MAXBP) {
      const int k = threadIdx.x & (MAXBP - 1);
      const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      if (part) part[((int64_t)b * gridDim.x + blockIdx.x) * 2 * MAXBP + threadIdx.x] = v;     // chunk partials: render_finish_kernel
      else if (k < Bp) {
        if (threadIdx.x < MAXBP) { if (g_zp) g_zp[(int64_t)b * Bp + k] = training ? v : 0.f; }
        else g_zd[(int64_t)b * Bp + k] = v;
      }
    }
  }
}
// g_z_pres / g_z_depth [B,Bp] = the chunk partials added in chunk order
__global__ __launch_bounds__(64) void render_finish_kernel(const float* __restrict__ part, float* __restrict__ g_zp, float* __restrict__ g_zd,
                                                           int Bp, int chunks) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= 2 * MAXBP) return;
  const int k = t & (MAXBP - 1);
  if (k >= Bp) return;
  float v = 0.f;
  for (int c = 0; c < chunks; ++c) v += part[((int64_t)b * chunks + c) * 2 * MAXBP + t];
  if (t < MAXBP) g_zp[(int64_t)b * Bp + k] = v;
  else g_zd[(int64_t)b * Bp + k] = v;
}
}  // namespace

extern "C" int sv_spair_render_fwd(const float* obj, const float* bg, const float* z_depth, const float* z_pres,
                                   const float* z_pres_logits, const float* noise, float* out, int32_t B, int32_t Bp, int32_t H,
                                   int32_t W, int32_t C, int32_t training, void* stream) {
  if (!obj || !bg || !z_depth || !out || B < 1 || Bp < 1 || Bp > MAXBP || C < 1 || C > MAXC || H < 1 || W < 1) return SV_E_BADARG;
  if ((training && !z_pres) || (!training && !z_pres_logits) || (!training && noise)) return SV_E_BADARG;
  hipLaunchKernelGGL((spair_render_kernel<false>), dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, obj, bg, z_depth, z_pres,
                     z_pres_logits, noise, out, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, Bp, H * W, C, training ? 1 : 0, (float*)nullptr);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_spair_render_bwd(const float* obj, const float* bg, const float* z_depth, const float* z_pres,
                                   const float* noise, const float* g_out, float* g_obj, float* g_bg, float* g_z_pres,
                                   float* g_z_depth, int32_t B, int32_t Bp, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!obj || !bg || !z_depth || !z_pres || !g_out || !g_obj || !g_bg || !g_z_pres || !g_z_depth) return SV_E_BADARG;
  if (B < 1 || Bp < 1 || Bp > MAXBP || C < 1 || C > MAXC || H < 1 || W < 1) return SV_E_BADARG;
  hipLaunchKernelGGL((spair_render_kernel<true>), dim3(1, B), dim3(256), 0, (hipStream_t)stream, obj, bg, z_depth, z_pres,
                     (const float*)nullptr, noise, (float*)nullptr, g_out, g_obj, g_bg, g_z_pres, g_z_depth, Bp, H * W, C, 1, (float*)nullptr);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int64_t sv_spair_render_bwd_workspace_floats(int32_t B, int32_t H, int32_t W) {
  return B < 1 || H < 1 || W < 1 ? 0 : (int64_t)B * ((H * W + 255) / 256) * 2 * MAXBP;
}

extern "C" int sv_spair_render_bwd_ws(const float* obj, const float* bg, const float* z_depth, const float* z_pres, const float* noise,
                                      const float* g_out, float* g_obj, float* g_bg, float* g_z_pres, float* g_z_depth, int32_t B,
                                      int32_t Bp, int32_t H, int32_t W, int32_t C, float* ws, int64_t ws_floats, void* stream) {
  if (!obj || !bg || !z_depth || !z_pres || !g_out || !g_obj || !g_bg || !g_z_pres || !g_z_depth) return SV_E_BADARG;
  if (B < 1 || Bp < 1 || Bp > MAXBP || C < 1 || C > MAXC || H < 1 || W < 1) return SV_E_BADARG;
  if (!ws || ws_floats < sv_spair_render_bwd_workspace_floats(B, H, W)) return SV_E_BADARG;
  const int chunks = (H * W + 255) / 256;
  hipLaunchKernelGGL((spair_render_kernel<true>), dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, obj, bg, z_depth, z_pres,
                     (const float*)nullptr, noise, (float*)nullptr, g_out, g_obj, g_bg, g_z_pres, g_z_depth, Bp, H * W, C, 1, ws);
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL(render_finish_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ws, g_z_pres, g_z_depth, Bp, chunks);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ============================================================================ sequential z_pres KL (spair/trainer.py:28-42, :45-94)
// compute_z_pres_kl_yolo_air: the cells are visited in raster order; the prior odds of cell i follow from the count
// distribution over 0..n objects conditioned on the cells switched on so far (z_pres > 0.5).  One thread per image carries
// the (n+1)-entry distribution through the n <= 16 cells -- the recurrence is serial by construction and tiny.  The prior
// depends on the SAMPLES only (comparisons, no gradient), so the tape's gradient reaches the posterior logits and the
// pre-sigmoid sample alone:  d kl / d y = 2T (e_q/(1+e_q+eps) - e_p/(1+e_p+eps)),  d kl / d q = 1 - 2 e_q/(1+e_q+eps),
// e_x = exp(-yT + x).  kl [B] = per-image sums (tf_mean_sum's inner sum); gradients are scaled by grad_scale (1/B for the mean).
namespace {
__device__ __forceinline__ float safe_log(float v) {       // tf_safe_log (:97-101)
  const float l = logf(v + 1e-8f);
  return (isnan(l) || isinf(l)) ? -100.f : l;
}
__global__ __launch_bounds__(64) void zpres_kl_kernel(const float* __restrict__ z_pres, const float* __restrict__ logits,
                                                      const float* __restrict__ pre, float* __restrict__ kl,
                                                      float* __restrict__ g_pre, float* __restrict__ g_logits, int B, int n,
                                                      float prior_prob, float T, float gscale, const float* __restrict__ prior_dev) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  if (prior_dev) prior_prob = *prior_dev;                      // hipGraph replay: the annealed prior lives in device memory
  float dist[MAXBP + 1];
  const float cpp = 1.f - prior_prob;
  float norm = 0.f;
#pragma unroll
  for (int k = 0; k <= MAXBP; ++k) {
    dist[k] = k <= n ? (1.f - cpp) * powf(cpp, (float)k) : 0.f;
    norm += dist[k];
  }
  norm = fmaxf(norm, 1e-6f);
#pragma unroll
  for (int k = 0; k <= MAXBP; ++k) dist[k] /= norm;
  float so_far = 0.f, total = 0.f;
  const float lt = logf(T + 1e-8f);
  for (int i = 0; i < n; ++i) {
    float pz = 0.f;
    const float inv = 1.f / (float)(n - i);
#pragma unroll
    for (int k = 0; k <= MAXBP; ++k) pz += dist[k] * (fmaxf((float)k - so_far, 0.f) * inv);
    const float p = safe_log(pz) - safe_log(1.f - pz);     // prior log-odds
    const float y = pre[(int64_t)b * n + i], q = logits[(int64_t)b * n + i];
    const float eq = expf(-y * T + q), ep = expf(-y * T + p);
    const float log_post = lt - y * T + q - 2.f * logf(1.f + eq + 1e-8f);
    const float log_prior = lt - y * T + p - 2.f * logf(1.f + ep + 1e-8f);
    total += log_post - log_prior;
    if (g_pre) g_pre[(int64_t)b * n + i] = gscale * 2.f * T * (eq / (1.f + eq + 1e-8f) - ep / (1.f + ep + 1e-8f));
    if (g_logits) g_logits[(int64_t)b * n + i] = gscale * (1.f - 2.f * eq / (1.f + eq + 1e-8f));
    const float s = z_pres[(int64_t)b * n + i] > 0.5f ? 1.f : 0.f;
    float nn = 0.f;
#pragma unroll
    for (int k = 0; k <= MAXBP; ++k) {
      const float pg = fmaxf((float)k - so_far, 0.f) * inv;
      dist[k] = (s * pg + (1.f - s) * (1.f - pg)) * dist[k];
      nn += dist[k];
    }
    nn = fmaxf(nn, 1e-6f);
#pragma unroll
    for (int k = 0; k <= MAXBP; ++k) dist[k] /= nn;
    so_far += s;
  }
  kl[b] = total;
}
}  // namespace

extern "C" int sv_spair_zpres_kl_dyn(const float* z_pres, const float* z_pres_logits, const float* z_pres_pre_sigmoid, float* kl,
                                     float* g_pre_sigmoid, float* g_logits, int32_t B, int32_t n_cells, float prior_prob,
                                     const float* prior_prob_dev, float temperature, float grad_scale, void* stream) {
  if (!z_pres || !z_pres_logits || !z_pres_pre_sigmoid || !kl || B < 1 || n_cells < 1 || n_cells > MAXBP) return SV_E_BADARG;
  hipLaunchKernelGGL(zpres_kl_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, z_pres, z_pres_logits,
                     z_pres_pre_sigmoid, kl, g_pre_sigmoid, g_logits, B, n_cells, prior_prob, temperature, grad_scale, prior_prob_dev);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_spair_zpres_kl(const float* z_pres, const float* z_pres_logits, const float* z_pres_pre_sigmoid, float* kl,
                                 float* g_pre_sigmoid, float* g_logits, int32_t B, int32_t n_cells, float prior_prob,
                                 float temperature, float grad_scale, void* stream) {
  return sv_spair_zpres_kl_dyn(z_pres, z_pres_logits, z_pres_pre_sigmoid, kl, g_pre_sigmoid, g_logits, B, n_cells, prior_prob, nullptr,
                               temperature, grad_scale, stream);
}
