// SPLIT-SPAIR's per-image loss reductions (spair/trainer.py), fp32, each with the gradient tape.gradient takes through it:
//   mode 0  xent_loss :103-104          t = -(x L(p) + (1 - x) L(1 - p))                                a = label x, b = prediction p
//   mode 1  kl_divergence :13-21        t = -0.5 (1 + L(s^2) - m^2 - exp(L(s^2)))                       a = z_mean m, b = z_sig s
//   mode 2  kl_divergence_two_gauss :23-24 against a CONSTANT prior N(m2, s2) (the zoom prior of :156-157):
//                                       t = L(s2) - L(s) + (s^2 + (m - m2)^2) / (2 s2^2) - 0.5          a = m, b = s
// with L = tf_safe_log :97-101: log(v + 1e-8), NaN / inf replaced by -100 (and no gradient through a replaced element).
// tf_mean_sum :107-109 = the batch mean of the per-image sums this kernel writes: sums[b] = sum_i t[b, i].
// Gradients (optional): ga / gb [B, n] = d t / d a, d t / d b per element (mode 0: gb only -- the label takes none).
// One workgroup per image, fixed-order reduction: deterministic.  HBM-bound (8 -16 B per element), tiny at SPAIR's sizes: the
// point is ONE launch instead of the ~25 elementwise launches of the composed expression.
#include "common.hip.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float safe_log(float v, bool& ok) {
  const float lv = logf(v + 1e-8f);
  ok = !(isnan(lv) || isinf(lv));
  return ok ? lv : -100.0f;
}

template <int MODE>
__global__ __launch_bounds__(256) void spair_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ sums,
                                                         float* __restrict__ ga, float* __restrict__ gb, int n, float m2, float s2,
                                                         const float* __restrict__ m2_dev) {
  const int64_t base = (int64_t)blockIdx.x * n;
  if (MODE == 2 && m2_dev) m2 = *m2_dev;                       // hipGraph replay: the annealed prior mean lives in device memory
  bool ok2;
  const float ls2 = MODE == 2 ? safe_log(s2, ok2) : 0.f, inv2 = MODE == 2 ? 1.0f / (2.0f * s2 * s2) : 0.f;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = a[base + i], y = b[base + i];
    float t, da = 0.f, db;
    bool ok0, ok1;
    if (MODE == 0) {
      const float l0 = safe_log(y, ok0), l1 = safe_log(1.0f - y, ok1);
      t = -(x * l0 + (1.0f - x) * l1);
      db = -((ok0 ? x / (y + 1e-8f) : 0.f) - (ok1 ? (1.0f - x) / (1.0f - y + 1e-8f) : 0.f));
    } else if (MODE == 1) {
      const float lv = safe_log(y * y, ok0);
      t = -0.5f * (1.0f + lv - x * x - expf(lv));
      da = x;
      db = ok0 ? -0.5f * (2.0f * y / (y * y + 1e-8f) - 2.0f * y * expf(lv) / (y * y + 1e-8f)) : 0.f;      // d exp(L)/ds = exp(L) dL/ds
    } else {
      const float l1 = safe_log(y, ok0);
      const float d = x - m2;
      t = ls2 - l1 + (y * y + d * d) * inv2 - 0.5f;
      da = 2.0f * d * inv2;
      db = -(ok0 ? 1.0f / (y + 1e-8f) : 0.f) + 2.0f * y * inv2;
    }
    acc += t;
    if (ga) ga[base + i] = da;
    if (gb) gb[base + i] = db;
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int sv_spair_loss_dyn(int32_t mode, const float* a, const float* b, float* sums, float* ga, float* gb, int32_t B, int32_t n,
                                 float prior_mean, const float* prior_mean_dev, float prior_sig, void* stream) {
  if (!a || !b || !sums || B < 1 || n < 1 || mode < 0 || mode > 2) return SV_E_BADARG;
  if (mode == 2 && !(prior_sig > 0.f)) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL((spair_loss_kernel<0>), dim3(B), dim3(256), 0, st, a, b, sums, (float*)nullptr, gb, n, 0.f, 1.f, (const float*)nullptr);
  else if (mode == 1) hipLaunchKernelGGL((spair_loss_kernel<1>), dim3(B), dim3(256), 0, st, a, b, sums, ga, gb, n, 0.f, 1.f, (const float*)nullptr);
  else hipLaunchKernelGGL((spair_loss_kernel<2>), dim3(B), dim3(256), 0, st, a, b, sums, ga, gb, n, prior_mean, prior_sig, prior_mean_dev);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_spair_loss(int32_t mode, const float* a, const float* b, float* sums, float* ga, float* gb, int32_t B, int32_t n,
                             float prior_mean, float prior_sig, void* stream) {
  return sv_spair_loss_dyn(mode, a, b, sums, ga, gb, B, n, prior_mean, nullptr, prior_sig, stream);
}
