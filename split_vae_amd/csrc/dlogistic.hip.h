// Discretised-logistic element function, shared by dlogistic_kernel (pointwise.hip) and the fused loss epilogue of the
// polyphase decoder head (tile_conv.hip).
#pragma once
#include "common.hip.h"

// vae/trainer.py:21-38, one element.  Returns nll and d nll/d m, d nll/d log_scale.
// The kernel is VALU-bound (24 576 elements per image, several transcendentals each), so the element
// function is written on the hardware exp2/log2/rcp units (__expf/__logf, 1-2 ulp at these argument
// ranges) and shares every exponential: e_v = exp(-|v|) gives sigmoid(v), 1-sigmoid(v) AND softplus(+-v).
__device__ __forceinline__ void sig_pair(float v, float& s, float& sc, float& e) {
  e = __expf(-fabsf(v));
  const float hi = __builtin_amdgcn_rcpf(1.f + e), lo = e * hi;
  s = v >= 0.f ? hi : lo;     // sigmoid(v)
  sc = v >= 0.f ? lo : hi;    // 1 - sigmoid(v)
}
// log(1 + e) for e in [0, 1]: series below 0.01 (the hardware log of 1+e would round e away)
__device__ __forceinline__ float log1p_unit(float e) {
  const float ser = e * (1.f - e * (0.5f - e * (0.33333334f - e * 0.25f)));
  return e < 0.01f ? ser : __logf(1.f + e);
}

__device__ __forceinline__ void dll_elem(float x, float m, float ls, float& nll, float& dm, float& dls) {
  const float c = x - m;
  const float s = __expf(-ls);
  const float p = s * (c + (1.f / 255.f));
  const float q = s * (c - (1.f / 255.f));
  float sp, spc, ep, sq, sqc, eq;
  sig_pair(p, sp, spc, ep);
  sig_pair(q, sq, sqc, eq);
  // cdf_delta = sigmoid(p) - sigmoid(q) without cancellation: with d = p - q = 2*s/255 > 0,
  //   sigmoid(p) - sigmoid(q) = sigmoid(p) * (1 - sigmoid(q)) * (1 - exp(-d))
  // (every factor is computed to full relative precision; the reference's direct difference of two
  // fp32 sigmoids loses up to ~1e-3 relative in the saturated tails).  om = 1 - exp(-d): alternating
  // series below 0.25 (truncation < 4e-10 relative), hardware exp above.
  const float d = s * (2.f / 255.f);
  const float ser = d * (1.f - d * (0.5f - d * (0.16666667f - d * (0.041666668f - d * (8.3333338e-3f - d * (1.3888889e-3f - d * 1.9841270e-4f))))));
  // (the hardware exp only where a lane of the wave needs it: wave-uniform branches skip quarter-rate instructions whose results a
  //  select would throw away -- the element function is VALU time one to one in the head's fused epilogue)
  float om = ser;
  if (__any(!(d < 0.25f))) om = d < 0.25f ? ser : 1.f - __expf(-d);
  const float delta = sp * sqc * om;
  // the three tf.where branches that do not diverge (edges are common: every saturated pixel):
  //   x < -0.999: log_cdf_plus = -softplus(-p) ; x > 0.999: log_one_minus_cdf_min = -softplus(q)
  const bool lo = x < -0.999f, hi = x > 0.999f;
  float nll_lo = 0.f, nll_hi = 0.f;                      // saturated pixels only (level 0 / 255): rare outside flat image regions
  if (__any(lo)) nll_lo = fmaxf(-p, 0.f) + log1p_unit(ep);
  if (__any(hi)) nll_hi = fmaxf(q, 0.f) + log1p_unit(eq);
  // d/dm log(delta) = s*(sigmoid(q) - (1-sigmoid(p))) ; d/dls = -p(1-sig(p)) + q sig(q) - d/expm1(d), d/expm1(d) = d*exp(-d)/om
  const float nll_mid = -__logf(delta);                // max(delta,1e-12) == delta on this branch
  const float dls_mid = p * spc - q * sq + d * (1.f - om) * __builtin_amdgcn_rcpf(om);
  nll = lo ? nll_lo : hi ? nll_hi : nll_mid;
  dm = lo ? s * spc : hi ? -s * sq : s * (spc - sq);
  dls = lo ? p * spc : hi ? -q * sq : dls_mid;
  if (!lo && !hi && !(delta > 1e-5f)) {                // rare: log_pdf_mid - log(127.5)
    const float mid = s * c;
    float sm, smc, em;
    sig_pair(mid, sm, smc, em);
    const float k = smc - sm;   // 1 - 2 sigmoid(mid)
    nll = -(mid - ls - 2.f * (fmaxf(mid, 0.f) + log1p_unit(em))) + 4.8481163645436525f;  // log(127.5)
    dm = s * k;
    dls = mid * k + 1.f;
  }
}

