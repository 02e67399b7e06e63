// Pointwise kernels of the SPLIT-GMVAE global encoder (Encoder(type='gmvae'), vae/model.py:48-79,
// call_gmvae :116-135) and of its loss terms (vae/trainer.py:17-18, :157-165).  Everything here is
// [B, <=1024]-sized glue between the dense / conv contractions (which run on the MFMA kernels), so
// the kernels are simple: one thread per element, one wave per row where a row reduction is needed.
// Randomness (dropout masks, Gumbel noise, eps) is either supplied by the caller (parity tests) or
// drawn from the counter-based Philox stream keyed by (seed, step, stream id, GLOBAL sample index,
// column) like the rest of the step, and written back so that the backward pass reuses it.
#include "common.hip.h"
#include "../../include/splitvae.h"

namespace {

template <typename T> __device__ __forceinline__ float ld_any(const void* p, int64_t i) { return to_f32(((const T*)p)[i]); }
__device__ __forceinline__ float ld_dt(const void* p, int dtype, int64_t i) {
  return dtype == SV_BF16 ? ld_any<bf16_t>(p, i) : ld_any<float>(p, i);
}
__device__ __forceinline__ void st_dt(void* p, int dtype, int64_t i, float v) {
  if (dtype == SV_BF16) ((bf16_t*)p)[i] = from_f32<bf16_t>(v); else ((float*)p)[i] = v;
}
__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : expm1f(v); }
// act'(pre) from the activation OUTPUT y: relu: y > 0 ; elu: y > 0 ? 1 : y + 1 (= exp(pre))
__device__ __forceinline__ float dact_from_out(float y, int act) {
  if (act == SV_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == SV_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
  return 1.f;
}
__device__ __forceinline__ float philox_unit(uint64_t seed, uint64_t step, int stream_id, uint64_t gs, uint32_t col) {
  Philox ph(seed ^ 0x6d76616547ULL);
  uint32_t c[4] = {col, (uint32_t)gs, (uint32_t)(gs >> 32) ^ (0x676d0000u + (uint32_t)stream_id), (uint32_t)step};
  ph(c);
  return u32_to_unit_open(c[0]);                       // (0, 1]
}

// x[b][c] = dropout(act(a[b][c] (+ bias))) ; columns C..ldx-1 zeroed (MFMA K padding of the next layer)
__global__ __launch_bounds__(256) void act_fwd_kernel(const void* a, int a_dtype, int lda, void* y_act, void* x, int x_dtype,
                                                      int ldx, int64_t rows, int C, int act, float rate,
                                                      const float* keep_in, float* keep_out, uint64_t seed,
                                                      uint64_t step, int stream_id, int64_t sample_offset,
                                                      int rows_per_sample) {
  const int64_t total = rows * ldx;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ldx;
    const int c = (int)(i - r * ldx);
    float v = 0.f;
    if (c < C) {
      v = ld_dt(a, a_dtype, r * lda + c);
      if (act == SV_ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == SV_ACT_ELU) v = elu_f(v);
      if (y_act) st_dt(y_act, x_dtype, r * ldx + c, v);          // pre-dropout activation (for act')
      if (rate > 0.f) {
        float keep;
        if (keep_in) keep = keep_in[r * C + c];
        else {
          const uint64_t gs = (uint64_t)(sample_offset + r / rows_per_sample);
          const uint32_t col = (uint32_t)((r % rows_per_sample) * C + c);
          keep = philox_unit(seed, step, stream_id, gs, col) > rate ? 1.f : 0.f;   // P(keep) = 1 - rate
        }
        if (keep_out) keep_out[r * C + c] = keep;
        v = v * keep * (1.f / (1.f - rate));                        // tf.nn.dropout scaling
      }
    } else if (y_act) st_dt(y_act, x_dtype, r * ldx + c, 0.f);
    st_dt(x, x_dtype, i, v);
  }
}

// ga[b][c] = (gx[b][c] * keep/(1-rate) + gx2[b][c]) * act'(y_act[b][c]) ; padding columns zeroed
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* gx, int gx_dtype, int ldg, const void* gx2, int gx2_dtype,
                                                      int ldg2, const void* y_act, int y_dtype, int ldy, int act,
                                                      float rate, const float* keep, void* ga, int ga_dtype, int ldga,
                                                      int64_t rows, int C) {
  const int64_t total = rows * ldga;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ldga;
    const int c = (int)(i - r * ldga);
    float g = 0.f;
    if (c < C) {
      g = ld_dt(gx, gx_dtype, r * ldg + c);
      if (rate > 0.f) g *= keep[r * C + c] * (1.f / (1.f - rate));
      if (gx2) g += ld_dt(gx2, gx2_dtype, r * ldg2 + c);
      if (y_act) g *= dact_from_out(ld_dt(y_act, y_dtype, r * ldy + c), act);
    }
    st_dt(ga, ga_dtype, i, g);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const void* a, const void* b, void* out, int dtype, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    st_dt(out, dtype, i, ld_dt(a, dtype, i) + ld_dt(b, dtype, i));
}

// y = softmax((logits - log(-log u)) / tau) over the K real columns (vae/model.py:121-122); one wave per row
__global__ __launch_bounds__(256) void gumbel_fwd_kernel(const float* logits, int ldl, const float* u_in, float* u_out,
                                                         float tau, float* y, void* y_lp, int lp_dtype, int ld_lp,
                                                         int B, int K, uint64_t seed, uint64_t step, int64_t sample_offset) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float t[2], mx = -3.0e38f;                         // K <= 128: two columns per lane
  for (int q = 0; q < 2; ++q) {
    const int k = lane + 64 * q;
    t[q] = -3.0e38f;
    if (k < K) {
      float u = u_in ? u_in[(int64_t)b * K + k] : philox_unit(seed, step, 7, (uint64_t)(sample_offset + b), (uint32_t)k);
      u = fminf(fmaxf(u, 1e-20f), 0.99999994f);       // keep log(-log u) finite at the ends of [0,1)
      if (u_out) u_out[(int64_t)b * K + k] = u;
      t[q] = (logits[(int64_t)b * ldl + k] - logf(-logf(u))) / tau;
    }
    mx = fmaxf(mx, t[q]);
  }
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e[2], s = 0.f;
  for (int q = 0; q < 2; ++q) { e[q] = lane + 64 * q < K ? expf(t[q] - mx) : 0.f; s += e[q]; }
  s = wave_sum(s);
  for (int q = 0; q < 2; ++q) {
    const int k = lane + 64 * q;
    if (k < K) y[(int64_t)b * K + k] = e[q] / s;
    if (k < ld_lp) st_dt(y_lp, lp_dtype, (int64_t)b * ld_lp + k, k < K ? e[q] / s : 0.f);
  }
}

// d logits = (1/tau) y (gy - <y, gy>)  +  (alpha/B) p (f - <p, f>),  p = softmax(logits), f = log(p+1e-8) + log K + p/(p+1e-8)
// (vae/trainer.py:161-165: the categorical term uses softmax(y_logits), not the Gumbel sample); also emits sum_k p(log(p+1e-8)+log K)
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(const float* gy, int ldg, const float* y, const float* logits, int ldl,
                                                         float tau, float alpha_over_B, void* g_logits, int g_dtype,
                                                         int ld_out, float* ykl, int B, int K) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float l[2], mx = -3.0e38f, yy[2], gg[2];
  for (int q = 0; q < 2; ++q) {
    const int k = lane + 64 * q;
    l[q] = k < K ? logits[(int64_t)b * ldl + k] : -3.0e38f;
    yy[q] = k < K ? y[(int64_t)b * K + k] : 0.f;
    gg[q] = (k < K && gy) ? gy[(int64_t)b * ldg + k] : 0.f;
    mx = fmaxf(mx, l[q]);
  }
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e[2], s = 0.f, dot = 0.f;
  for (int q = 0; q < 2; ++q) { e[q] = lane + 64 * q < K ? expf(l[q] - mx) : 0.f; s += e[q]; dot += yy[q] * gg[q]; }
  s = wave_sum(s);
  dot = wave_sum(dot);
  const float logK = logf((float)K);
  float p[2], f[2], pf = 0.f, kl = 0.f;
  for (int q = 0; q < 2; ++q) {
    p[q] = e[q] / s;
    const float lp = logf(p[q] + 1e-8f);
    f[q] = lp + logK + p[q] / (p[q] + 1e-8f);
    if (lane + 64 * q < K) { pf += p[q] * f[q]; kl += p[q] * (lp + logK); }
  }
  pf = wave_sum(pf);
  kl = wave_sum(kl);
  if (ykl && lane == 0) ykl[b] = kl;
  if (!g_logits) return;
  for (int q = 0; q < 2; ++q) {
    const int k = lane + 64 * q;
    if (k >= ld_out) continue;
    const float g = k < K ? yy[q] * (gg[q] - dot) / tau + alpha_over_B * p[q] * (f[q] - pf) : 0.f;
    st_dt(g_logits, g_dtype, (int64_t)b * ld_out + k, g);
  }
}

// posterior / prior heads: z_mean = a_m, z_sig = softplus(a_s), z = z_mean + z_sig*eps (vae/model.py:131-133, :9-13);
// prior mean = a_pm, prior sig = softplus(a_ps) (:124-125); kl2[b] = vae/trainer.py:17-18 summed over the row
__global__ __launch_bounds__(256) void gm_head_fwd_kernel(const float* a_m, const float* a_s, const float* a_pm, const float* a_ps,
                                                          const float* eps, float* eps_out, float* zm, float* zs, float* z,
                                                          float* pm, float* ps, void* z_lp, int lp_dtype, int ldz, int z_col,
                                                          float* kl2, int B, int L, uint64_t seed, uint64_t step,
                                                          int64_t sample_offset) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  Philox ph(seed ^ 0xe9515eedULL);                   // the Sampling stream of the LGVae step (stream id 0 = encoder_x)
  const uint64_t gs = (uint64_t)(sample_offset + b);
  float acc = 0.f;
  for (int j = lane; j < L; j += 64) {
    const int64_t i = (int64_t)b * L + j;
    const float m = a_m[i], s = softplus_f(a_s[i]), m2 = a_pm[i], s2 = softplus_f(a_ps[i]);
    float e;
    if (eps) e = eps[i];
    else {
      uint32_t c[4] = {(uint32_t)j, (uint32_t)gs, (uint32_t)(gs >> 32) ^ 0x65707300u, (uint32_t)step};
      ph(c);
      e = sqrtf(-2.f * logf(u32_to_unit_open(c[0]))) * cosf(6.283185307179586f * u32_to_unit_open(c[1]));
    }
    if (eps_out) eps_out[i] = e;
    const float zz = m + s * e;
    zm[i] = m; zs[i] = s; z[i] = zz; pm[i] = m2; ps[i] = s2;
    st_dt(z_lp, lp_dtype, (int64_t)b * ldz + z_col + j, zz);
    acc += logf(s2) - logf(s) + (s * s + (m - m2) * (m - m2)) / (2.f * s2 * s2) - 0.5f;
  }
  acc = wave_sum(acc);
  if (lane == 0) kl2[b] = acc;
}

// adjoint of the heads: dL/dz from the decoder plus c = beta/B times d kl2; through softplus via sigmoid(a) = 1 - exp(-softplus(a))
__global__ __launch_bounds__(256) void gm_head_bwd_kernel(const float* dz, int lddz, const float* zm, const float* zs, const float* pm,
                                                          const float* ps, const float* eps, float c, void* g_am, void* g_as,
                                                          void* g_apm, void* g_aps, int g_dtype, int B, int L) {
  const int64_t total = (int64_t)B * L;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / L;
    const int j = (int)(i - b * L);
    const float g = dz[b * lddz + j], m = zm[i], s = zs[i], m2 = pm[i], s2 = ps[i], d = m - m2, is2 = 1.f / (s2 * s2);
    const float gm = g + c * d * is2;
    const float gs = g * eps[i] + c * (s * is2 - 1.f / s);
    const float gm2 = -c * d * is2;
    const float gs2 = c * (1.f / s2 - (s * s + d * d) * is2 / s2);
    st_dt(g_am, g_dtype, i, gm);
    st_dt(g_as, g_dtype, i, gs * (1.f - expf(-s)));
    st_dt(g_apm, g_dtype, i, gm2);
    st_dt(g_aps, g_dtype, i, gs2 * (1.f - expf(-s2)));
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  return (unsigned)(b < 1 ? 1 : b);
}
inline bool dt_ok(int d) { return d == SV_BF16 || d == SV_F32; }

}  // namespace

extern "C" int sv_act_fwd(const void* a, int32_t a_dtype, int32_t lda, void* y_act, void* x, int32_t x_dtype, int32_t ldx,
                          int64_t rows, int32_t C, int32_t act, float drop_rate, const float* keep_in, float* keep_out,
                          uint64_t seed, uint64_t step, int32_t stream_id, int64_t sample_offset, int32_t rows_per_sample,
                          void* stream) {
  if (!a || !x || rows <= 0 || C <= 0 || lda < C || ldx < C || !dt_ok(a_dtype) || !dt_ok(x_dtype)) return SV_E_BADARG;
  if (act != SV_ACT_NONE && act != SV_ACT_RELU && act != SV_ACT_ELU) return SV_E_BADARG;
  if (drop_rate < 0.f || drop_rate >= 1.f || rows_per_sample <= 0) return SV_E_BADARG;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(rows * ldx)), dim3(256), 0, (hipStream_t)stream, a, a_dtype, lda, y_act, x,
                     x_dtype, ldx, rows, C, act, drop_rate, keep_in, keep_out, seed, step, stream_id, sample_offset,
                     rows_per_sample);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_act_bwd(const void* gx, int32_t gx_dtype, int32_t ldg, const void* gx2, int32_t gx2_dtype, int32_t ldg2,
                          const void* y_act, int32_t y_dtype, int32_t ldy, int32_t act, float drop_rate, const float* keep,
                          void* ga, int32_t ga_dtype, int32_t ldga, int64_t rows, int32_t C, void* stream) {
  if (!gx || !ga || rows <= 0 || C <= 0 || ldg < C || ldga < C || !dt_ok(gx_dtype) || !dt_ok(ga_dtype)) return SV_E_BADARG;
  if (drop_rate > 0.f && !keep) return SV_E_BADARG;
  if (gx2 && (!dt_ok(gx2_dtype) || ldg2 < C)) return SV_E_BADARG;
  if (y_act && (!dt_ok(y_dtype) || ldy < C)) return SV_E_BADARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(rows * ldga)), dim3(256), 0, (hipStream_t)stream, gx, gx_dtype, ldg, gx2,
                     gx2_dtype, ldg2, y_act, y_dtype, ldy, act, drop_rate, keep, ga, ga_dtype, ldga, rows, C);
  SV_LAUNCH_CHECK();
  return SV_OK;
}


// The five Mean metrics of train_step_lg_gm_vae / test_step_lg_gm_vae (vae/trainer.py:157-173) + the total loss from the per-image terms:
// out[0..4] = batch means of (nll_x, kl(q_x || p_y), nll_xh, kl(q_xh || N(0,1)), KL(softmax(y_logits) || uniform)),
// out[5] = out[0] + out[2] + beta (out[1] + out[3]) + alpha out[4].  One workgroup, fixed-order tree: bit-reproducible.
static __global__ __launch_bounds__(256) void gm_metrics_kernel(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4,
                                                         int B, float beta, float alpha, float* out) {
  __shared__ float red[5][256];
  const float* term[5] = {t0, t1, t2, t3, t4};
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float s = 0.f;
    for (int b = tid; b < B; b += 256) s += term[k][b];
    red[k][tid] = s;
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w)
#pragma unroll
      for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    float m[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { m[k] = red[k][0] / (float)B; out[k] = m[k]; }
    out[5] = m[0] + m[2] + beta * (m[1] + m[3]) + alpha * m[4];
  }
}

extern "C" int sv_gm_metrics(const float* nll_x, const float* kl_x, const float* nll_xh, const float* kl_xh, const float* y_kl, int32_t B,
                             float beta, float alpha, float* out6, void* stream) {
  if (!nll_x || !kl_x || !nll_xh || !kl_xh || !y_kl || !out6 || B <= 0) return SV_E_BADARG;
  hipLaunchKernelGGL(gm_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll_x, kl_x, nll_xh, kl_xh, y_kl, B, beta, alpha, out6);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_add(const void* a, const void* b, void* out, int32_t dtype, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0 || !dt_ok(dtype)) return SV_E_BADARG;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, dtype, n);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gumbel_softmax_fwd(const float* logits, int32_t ld_logits, const float* u, float* u_out, float tau, float* y,
                                     void* y_lp, int32_t lp_dtype, int32_t ld_lp, int32_t B, int32_t K, uint64_t seed,
                                     uint64_t step, int64_t sample_offset, void* stream) {
  if (!logits || !y || !y_lp || B <= 0 || K <= 0 || K > 128 || ld_lp > 128 || ld_lp < K || ld_logits < K || tau <= 0.f ||
      !dt_ok(lp_dtype))
    return SV_E_BADARG;
  hipLaunchKernelGGL(gumbel_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld_logits, u, u_out, tau, y,
                     y_lp, lp_dtype, ld_lp, B, K, seed, step, sample_offset);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gumbel_softmax_bwd(const float* gy, int32_t ldg, const float* y, const float* logits, int32_t ld_logits,
                                     float tau, float alpha_over_B, void* g_logits, int32_t g_dtype, int32_t ld_out,
                                     float* y_kl, int32_t B, int32_t K, void* stream) {
  if (!y || !logits || B <= 0 || K <= 0 || K > 128 || ld_out > 128 || ld_logits < K || tau <= 0.f) return SV_E_BADARG;
  if (g_logits && (!dt_ok(g_dtype) || ld_out < K || !gy || ldg < K)) return SV_E_BADARG;
  hipLaunchKernelGGL(gumbel_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, gy, ldg, y, logits, ld_logits, tau,
                     alpha_over_B, g_logits, g_dtype, ld_out, y_kl, B, K);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gm_head_fwd(const float* a_mean, const float* a_sig, const float* a_prior_mean, const float* a_prior_sig,
                              const float* eps, float* eps_out, float* z_mean, float* z_sig, float* z, float* prior_mean,
                              float* prior_sig, void* z_lp, int32_t lp_dtype, int32_t ldz, int32_t z_col, float* kl2,
                              int32_t B, int32_t L, uint64_t seed, uint64_t step, int64_t sample_offset, void* stream) {
  if (!a_mean || !a_sig || !a_prior_mean || !a_prior_sig || !z_mean || !z_sig || !z || !prior_mean || !prior_sig || !z_lp ||
      !kl2 || B <= 0 || L <= 0 || ldz < z_col + L || !dt_ok(lp_dtype))
    return SV_E_BADARG;
  hipLaunchKernelGGL(gm_head_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, a_mean, a_sig, a_prior_mean,
                     a_prior_sig, eps, eps_out, z_mean, z_sig, z, prior_mean, prior_sig, z_lp, lp_dtype, ldz, z_col, kl2, B, L,
                     seed, step, sample_offset);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_gm_head_bwd(const float* dz, int32_t ld_dz, const float* z_mean, const float* z_sig, const float* prior_mean,
                              const float* prior_sig, const float* eps, float kl_scale, void* g_a_mean, void* g_a_sig,
                              void* g_a_prior_mean, void* g_a_prior_sig, int32_t g_dtype, int32_t B, int32_t L, void* stream) {
  if (!dz || !z_mean || !z_sig || !prior_mean || !prior_sig || !eps || !g_a_mean || !g_a_sig || !g_a_prior_mean ||
      !g_a_prior_sig || B <= 0 || L <= 0 || ld_dz < L || !dt_ok(g_dtype))
    return SV_E_BADARG;
  hipLaunchKernelGGL(gm_head_bwd_kernel, dim3(grid_for((int64_t)B * L)), dim3(256), 0, (hipStream_t)stream, dz, ld_dz, z_mean,
                     z_sig, prior_mean, prior_sig, eps, kl_scale, g_a_mean, g_a_sig, g_a_prior_mean, g_a_prior_sig, g_dtype, B, L);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
