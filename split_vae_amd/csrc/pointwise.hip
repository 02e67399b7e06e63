// HBM-bound kernels of the SPLIT-VAE step: patch scramble, ELBO terms, reparameterisation/KL,
// bilinear 2x (+adjoint), Keras-Adam.  One pass over the data each, 64-lane wavefront reductions.
#include "common.hip.h"
#include "kernels.h"
#include "dlogistic.hip.h"

// ============================================================================ A1 scramble
// augmentation.py:43-57.  x_aug[r*s+i, c*s+j] = x[pr*s+i, pc*s+j], (pr,pc) = divmod(perm[r*G+c], G).
// One thread per destination pixel; both halves of the 6-channel pixel are written by the same
// thread so the 24-B output pixel is produced in one place.
__global__ __launch_bounds__(256) void scramble_kernel(const float* __restrict__ x,
                                                       const int32_t* __restrict__ perm,
                                                       float* __restrict__ out, int B, int H, int W,
                                                       int s, int G) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int xw = (int)(idx % W);
    const int64_t t = idx / W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int r = y / s, i = y - r * s, c = xw / s, j = xw - c * s;
    const int p = perm[(int64_t)b * G * G + r * G + c];
    const int pr = p / G, pc = p - pr * G;
    const float* src0 = x + idx * 3;
    const float* src1 = x + (((int64_t)b * H + pr * s + i) * W + pc * s + j) * 3;
    float* dst = out + idx * 6;
    const float a0 = src0[0], a1 = src0[1], a2 = src0[2];
    const float b0 = src1[0], b1 = src1[1], b2 = src1[2];
    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = b0; dst[4] = b1; dst[5] = b2;
  }
}

// The same gather that also writes what the step's split_pad would derive from images6: the two 8-channel (zero padded) NHWC
// tensors the first encoder layers read, in the plan's contraction dtype -- images6 is then read once less and split_pad drops out.
template <typename T>
__global__ __launch_bounds__(256) void scramble_staged_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm,
                                                              float* __restrict__ out, T* __restrict__ x8, T* __restrict__ xh8,
                                                              int B, int H, int W, int s, int G) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int xw = (int)(idx % W);
    const int64_t t = idx / W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int r = y / s, i = y - r * s, c = xw / s, j = xw - c * s;
    const int p = perm[(int64_t)b * G * G + r * G + c];
    const int pr = p / G, pc = p - pr * G;
    const float* src0 = x + idx * 3;
    const float* src1 = x + (((int64_t)b * H + pr * s + i) * W + pc * s + j) * 3;
    float* dst = out + idx * 6;
    const float a0 = src0[0], a1 = src0[1], a2 = src0[2];
    const float b0 = src1[0], b1 = src1[1], b2 = src1[2];
    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = b0; dst[4] = b1; dst[5] = b2;
    T u[8], w[8];
    u[0] = from_f32<T>(a0); u[1] = from_f32<T>(a1); u[2] = from_f32<T>(a2);
    w[0] = from_f32<T>(b0); w[1] = from_f32<T>(b1); w[2] = from_f32<T>(b2);
#pragma unroll
    for (int e = 3; e < 8; ++e) { u[e] = from_f32<T>(0.f); w[e] = from_f32<T>(0.f); }
    if constexpr (sizeof(T) == 2) {
      *(uint4*)(x8 + idx * 8) = *(uint4*)u;
      *(uint4*)(xh8 + idx * 8) = *(uint4*)w;
    } else {
      *(uint4*)(x8 + idx * 8) = *(uint4*)u; *(uint4*)(x8 + idx * 8 + 4) = *(uint4*)(u + 4);
      *(uint4*)(xh8 + idx * 8) = *(uint4*)w; *(uint4*)(xh8 + idx * 8 + 4) = *(uint4*)(w + 4);
    }
  }
}

extern "C" int sv_scramble_gather_staged(const float* x, const int32_t* perm, float* images6, void* x8, void* xh8, int32_t dtype,
                                         int32_t B, int32_t H, int32_t W, int32_t patch, void* stream) {
  if (!x || !perm || !images6 || !x8 || !xh8 || B <= 0 || H <= 0 || W <= 0 || patch <= 0) return SV_E_BADARG;
  if (dtype != SV_BF16 && dtype != SV_F32) return SV_E_BADARG;
  if (H != W || H % patch) return SV_E_UNSUPPORTED;
  const int64_t total = (int64_t)B * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 16) grid = 256 * 16;
  if (dtype == SV_BF16)
    hipLaunchKernelGGL((scramble_staged_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, perm, images6, (bf16_t*)x8,
                       (bf16_t*)xh8, B, H, W, patch, W / patch);
  else
    hipLaunchKernelGGL((scramble_staged_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, perm, images6, (float*)x8,
                       (float*)xh8, B, H, W, patch, W / patch);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_scramble_gather(const float* x, const int32_t* perm, float* images6, int32_t B,
                                  int32_t H, int32_t W, int32_t patch, void* stream) {
  if (!x || !perm || !images6 || B <= 0 || H <= 0 || W <= 0 || patch <= 0) return SV_E_BADARG;
  if (H != W || H % patch) return SV_E_UNSUPPORTED;   // augmentation.py:44-46 assumes square, s | H
  const int64_t total = (int64_t)B * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(scramble_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, perm,
                     images6, B, H, W, patch, W / patch);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// tf.random.shuffle stand-in: sort Philox keys (ties impossible: index in the low word).
__global__ __launch_bounds__(256) void random_perm_kernel(int32_t* __restrict__ perm, int n, int npow2,
                                                          uint64_t seed, uint64_t step,
                                                          int64_t sample_offset) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint64_t* keys = (uint64_t*)smem_raw;
  const int b = blockIdx.x;
  const uint64_t gs = (uint64_t)(sample_offset + b);
  Philox ph(seed ^ 0x5ca1ab1e5eedULL);
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    uint64_t k = ~0ULL;
    if (i < n) {
      uint32_t c[4] = {(uint32_t)i, (uint32_t)gs, (uint32_t)(gs >> 32) ^ 0x7065726du, (uint32_t)step};
      ph(c);
      k = ((uint64_t)c[0] << 32) | (uint32_t)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = keys[i], c = keys[ixj];
          const bool up = ((i & k) == 0);
          if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) perm[(int64_t)b * n + i] = (int32_t)(uint32_t)keys[i];
}

extern "C" int sv_random_perm(int32_t* perm, int32_t B, int32_t n_patch, uint64_t seed, uint64_t step,
                              int64_t sample_offset, void* stream) {
  if (!perm || B <= 0 || n_patch <= 0) return SV_E_BADARG;
  if (n_patch > 4096) return SV_E_UNSUPPORTED;
  int npow2 = 1;
  while (npow2 < n_patch) npow2 <<= 1;
  hipLaunchKernelGGL(random_perm_kernel, dim3(B), dim3(256), npow2 * sizeof(uint64_t),
                     (hipStream_t)stream, perm, n_patch, npow2, seed, step, sample_offset);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ============================================================================ A6 discretised logistic
// (the element function lives in dlogistic.hip.h: the polyphase decoder head evaluates it in its epilogue too)
// TWIN: one thread handles the x AND the x-hat reconstruction of its pixel (images6 holds x | x-hat in one 24-B record:
// a launch per network -- or a network per blockIdx.z -- fetches every record twice, 1.23x the algorithmic bytes of the
// whole kernel by the PMC counters); the second network's buffers lie one stride (zs_*) further.
template <typename TG, bool GRAD, bool TWIN>
__global__ __launch_bounds__(256) void dlogistic_kernel(const float* __restrict__ images6, int ch_off,
                                                        const float* __restrict__ out6,
                                                        TG* __restrict__ grad, float gscale,
                                                        float* __restrict__ partial, int HW,
                                                        int pix_per_block, int64_t zs_out, int64_t zs_grad,
                                                        int64_t zs_part) {
  constexpr int NN = TWIN ? 2 : 1;
  if (!TWIN) {
    // blockIdx.z = network (stand-alone use): channel triple 3z of images6, buffers one net stride apart
    ch_off += 3 * blockIdx.z;
    out6 += blockIdx.z * zs_out;
    if (GRAD) grad += blockIdx.z * zs_grad;
    partial += blockIdx.z * zs_part;
  }
  const int part = blockIdx.x, b = blockIdx.y, P = gridDim.x;
  const int64_t base = (int64_t)b * HW;
  const int p0 = part * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  float acc[NN];
#pragma unroll
  for (int z = 0; z < NN; ++z) acc[z] = 0.f;
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const float* xi = images6 + (base + p) * 6;
    float xv[6];
    if (TWIN) {
      const float2 a = *(const float2*)(xi), c = *(const float2*)(xi + 2), e = *(const float2*)(xi + 4);
      xv[0] = a.x; xv[1] = a.y; xv[2] = c.x; xv[3] = c.y; xv[4] = e.x; xv[5] = e.y;
    } else {
      xv[0] = xi[ch_off]; xv[1] = xi[ch_off + 1]; xv[2] = xi[ch_off + 2];
    }
#pragma unroll
    for (int z = 0; z < NN; ++z) {
      const float* oi = out6 + z * zs_out + (base + p) * 6;
      const float2 o01 = *(const float2*)(oi), o23 = *(const float2*)(oi + 2), o45 = *(const float2*)(oi + 4);
      const float x0 = xv[3 * z], x1 = xv[3 * z + 1], x2 = xv[3 * z + 2];
      float n0, n1, n2, dm0, dm1, dm2, dl0, dl1, dl2;
      dll_elem(x0, o01.x, o23.y, n0, dm0, dl0);
      dll_elem(x1, o01.y, o45.x, n1, dm1, dl1);
      dll_elem(x2, o23.x, o45.y, n2, dm2, dl2);
      acc[z] += (n0 + n1) + n2;
      if (GRAD) {
        TG* gp = grad + z * zs_grad + (base + p) * 8;
        if constexpr (sizeof(TG) == 2) {
          bf16x8 v;
          v[0] = (bf16_t)(dm0 * gscale); v[1] = (bf16_t)(dm1 * gscale); v[2] = (bf16_t)(dm2 * gscale);
          v[3] = (bf16_t)(dl0 * gscale); v[4] = (bf16_t)(dl1 * gscale); v[5] = (bf16_t)(dl2 * gscale);
          v[6] = (bf16_t)0.f; v[7] = (bf16_t)0.f;
          *(bf16x8*)gp = v;
        } else {
          float4 a = make_float4(dm0 * gscale, dm1 * gscale, dm2 * gscale, dl0 * gscale);
          float4 c = make_float4(dl1 * gscale, dl2 * gscale, 0.f, 0.f);
          *(float4*)gp = a;
          *(float4*)(gp + 4) = c;
        }
      }
    }
  }
  // per-image reduction: 64-lane shuffle, then 4 waves through LDS (deterministic order)
  __shared__ float red[NN][4];
#pragma unroll
  for (int z = 0; z < NN; ++z) {
    const float a = wave_sum(acc[z]);
    if ((threadIdx.x & 63) == 0) red[z][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < NN) {
    const int z = threadIdx.x;
    partial[z * zs_part + (int64_t)b * P + part] = (red[z][0] + red[z][1]) + (red[z][2] + red[z][3]);
  }
}

__global__ void rowsum_partials_kernel(const float* __restrict__ partial, float* __restrict__ nll, int B, int P,
                                       int64_t zs_part, int64_t zs_nll) {
  partial += blockIdx.y * zs_part;
  nll += blockIdx.y * zs_nll;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int i = 0; i < P; ++i) s += partial[(int64_t)b * P + i];
  nll[b] = s;
}

static inline int dll_parts(int HW) { return HW > 1024 ? HW / 1024 : 1; }

int svk_nll_rowsum(const float* partial_ws, float* nll, int B, int P, int64_t zs_part, int64_t zs_nll, int nets, hipStream_t st) {
  hipLaunchKernelGGL(rowsum_partials_kernel, dim3((B + 255) / 256, nets), dim3(256), 0, st, partial_ws, nll, B, P, zs_part, zs_nll);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int64_t sv_dlogistic_nll_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  return (int64_t)B * dll_parts(H * W) * sizeof(float);
}

// nets = 1: plain call.  nets = 2: the x and x-hat reconstructions in one launch (channel triples
// ch_off and ch_off+3 of images6; out6 / nll / grad / partial_ws of the second net one stride further)
int svk_dlogistic_nll_multi(const float* images6, int ch_off, const float* out6, int64_t zs_out, float* nll,
                            int64_t zs_nll, void* grad, int64_t zs_grad, int grad_dtype, float grad_scale, int B,
                            int H, int W, float* partial_ws, int64_t zs_part, int nets, hipStream_t st) {
  const int HW = H * W, P = dll_parts(HW);
  const int ppb = (HW + P - 1) / P;
  const bool twin = nets == 2 && ch_off == 0;             // x | x-hat of one images6 record in one thread
  dim3 grid(P, B, twin ? 1 : nets), block(256);
#define SV_DLL_LAUNCH(TG_, GRAD_, g_, sc_)                                                                                  \
  do {                                                                                                                      \
    if (twin) hipLaunchKernelGGL((dlogistic_kernel<TG_, GRAD_, true>), grid, block, 0, st, images6, ch_off, out6, g_, sc_,  \
                                 partial_ws, HW, ppb, zs_out, zs_grad, zs_part);                                            \
    else hipLaunchKernelGGL((dlogistic_kernel<TG_, GRAD_, false>), grid, block, 0, st, images6, ch_off, out6, g_, sc_,      \
                            partial_ws, HW, ppb, zs_out, zs_grad, zs_part);                                                 \
  } while (0)
  if (!grad) SV_DLL_LAUNCH(float, false, (float*)nullptr, 0.f);
  else if (grad_dtype == SV_BF16) SV_DLL_LAUNCH(bf16_t, true, (bf16_t*)grad, grad_scale);
  else if (grad_dtype == SV_F32) SV_DLL_LAUNCH(float, true, (float*)grad, grad_scale);
#undef SV_DLL_LAUNCH
  else
    return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL(rowsum_partials_kernel, dim3((B + 255) / 256, nets), dim3(256), 0, st, partial_ws, nll, B, P,
                     zs_part, zs_nll);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_dlogistic_nll(const float* images6, int32_t ch_off, const float* out6, float* nll,
                                void* grad, int32_t grad_dtype, float grad_scale, int32_t B,
                                int32_t H, int32_t W, float* partial_ws, void* stream) {
  if (!images6 || !out6 || !nll || !partial_ws || B <= 0 || H <= 0 || W <= 0) return SV_E_BADARG;
  if (ch_off != 0 && ch_off != 3) return SV_E_BADARG;
  return svk_dlogistic_nll_multi(images6, ch_off, out6, 0, nll, 0, grad, 0, grad_dtype, grad_scale, B, H, W, partial_ws,
                                 0, 1, (hipStream_t)stream);
}

// ============================================================================ A4 + A7 reparam / KL
struct ReparamFwdArgs {
  const float *pre, *bias_mean, *bias_sd, *eps;
  float *eps_out, *z_mean, *z_sig, *z;
  void* z_lp;
  float* kl;
  int ldz, z_col, L, stream_id;
  int S; int64_t slab_stride;      // S > 0: `pre` is the first of S K-slice slabs of the heads' GEMM (latent_gemm.hip), summed here in slice order
};
template <typename TZ>
__device__ __forceinline__ void reparam_kl_fwd_body(const ReparamFwdArgs& g, int B, uint64_t seed, uint64_t step, int64_t sample_offset) {
  const float* __restrict__ pre = g.pre; const float* __restrict__ bias_mean = g.bias_mean; const float* __restrict__ bias_sd = g.bias_sd;
  const float* __restrict__ eps = g.eps;
  float* __restrict__ eps_out = g.eps_out; float* __restrict__ z_mean = g.z_mean; float* __restrict__ z_sig = g.z_sig;
  float* __restrict__ z = g.z; TZ* __restrict__ z_lp = (TZ*)g.z_lp; float* __restrict__ kl = g.kl;
  const int ldz = g.ldz, z_col = g.z_col, L = g.L, stream_id = g.stream_id;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  Philox ph(seed ^ 0xe9515eedULL);
  const uint64_t gs = (uint64_t)(sample_offset + b);
  float acc = 0.f;
  for (int j = lane; j < L; j += 64) {
    float pm, ps;
    if (g.S > 0) {                                         // as nt_slab_reduce_kernel: 0 + slab 0 + slab 1 + ... (bitwise the two-launch result)
      pm = 0.f; ps = 0.f;
      const float* __restrict__ q = pre + (int64_t)b * 2 * L + j;
#pragma unroll 8
      for (int k = 0; k < g.S; ++k) { pm += q[(int64_t)k * g.slab_stride]; ps += q[(int64_t)k * g.slab_stride + L]; }
    } else {
      pm = pre[(int64_t)b * 2 * L + j]; ps = pre[(int64_t)b * 2 * L + L + j];
    }
    const float mu = pm + bias_mean[j];
    const float sg = softplus_f(ps + bias_sd[j]);
    float e;
    if (eps) {
      e = eps[(int64_t)b * L + j];
    } else {
      uint32_t c[4] = {(uint32_t)j, (uint32_t)gs, (uint32_t)(gs >> 32) ^ (0x65707300u + (uint32_t)stream_id),
                       (uint32_t)step};
      ph(c);
      const float u1 = u32_to_unit_open(c[0]), u2 = u32_to_unit_open(c[1]);
      e = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);   // Box-Muller
    }
    if (eps_out) eps_out[(int64_t)b * L + j] = e;
    const float zz = mu + sg * e;                        // vae/model.py:13
    z_mean[(int64_t)b * L + j] = mu;
    z_sig[(int64_t)b * L + j] = sg;
    z[(int64_t)b * L + j] = zz;
    z_lp[(int64_t)b * ldz + z_col + j] = from_f32<TZ>(zz);
    const float lv = logf(sg * sg);                      // vae/trainer.py:12
    acc += 1.f + lv - mu * mu - expf(lv);
  }
  acc = wave_sum(acc);
  if (lane == 0) kl[b] = -0.5f * acc;
}

template <typename TZ>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(
    const float* __restrict__ pre, const float* __restrict__ bias_mean, const float* __restrict__ bias_sd,
    const float* __restrict__ eps, float* __restrict__ eps_out, float* __restrict__ z_mean, float* __restrict__ z_sig,
    float* __restrict__ z, TZ* __restrict__ z_lp, int ldz, int z_col, float* __restrict__ kl, int B,
    int L, uint64_t seed, uint64_t step, int stream_id, int64_t sample_offset, const SvDynArgs* __restrict__ dyn) {
  if (dyn) { seed = dyn->seed; step = dyn->step; sample_offset = dyn->sample_offset; }   // captured step (graph replay)
  const ReparamFwdArgs g = {pre, bias_mean, bias_sd, eps, eps_out, z_mean, z_sig, z, (void*)z_lp, kl, ldz, z_col, L, stream_id, 0, 0};
  reparam_kl_fwd_body<TZ>(g, B, seed, step, sample_offset);
}

// the x and x-hat heads in one launch (blockIdx.y picks the network): one dependent launch less on the critical path
struct ReparamFwdTwin { ReparamFwdArgs a[2]; };
template <typename TZ>
__global__ __launch_bounds__(256) void reparam_kl_fwd_twin_kernel(const ReparamFwdTwin t, int B, uint64_t seed, uint64_t step,
                                                                  int64_t sample_offset, const SvDynArgs* __restrict__ dyn) {
  if (dyn) { seed = dyn->seed; step = dyn->step; sample_offset = dyn->sample_offset; }
  reparam_kl_fwd_body<TZ>(t.a[blockIdx.y], B, seed, step, sample_offset);
}

int svk_reparam_kl_fwd_twin(const float* const* pre, const float* const* bias_mean, const float* const* bias_sd,
                            const float* const* eps, float* const* eps_out, float* const* z_mean, float* const* z_sig,
                            float* const* z, void* z_lp, int z_dtype, int ldz, const int* z_col, float* const* kl, int B,
                            const int* L, uint64_t seed, uint64_t step, int64_t sample_offset, hipStream_t st, const SvDynArgs* dyn,
                            const int* S, const int64_t* slab_stride) {
  ReparamFwdTwin t;
  for (int e = 0; e < 2; ++e)
    t.a[e] = {pre[e], bias_mean[e], bias_sd[e], eps[e], eps_out[e], z_mean[e], z_sig[e], z[e], z_lp, kl[e], ldz, z_col[e], L[e], e,
              S ? S[e] : 0, S ? slab_stride[e] : 0};
  dim3 grid((B + 3) / 4, 2), block(256);
  if (z_dtype == SV_BF16) hipLaunchKernelGGL((reparam_kl_fwd_twin_kernel<bf16_t>), grid, block, 0, st, t, B, seed, step, sample_offset, dyn);
  else if (z_dtype == SV_F32) hipLaunchKernelGGL((reparam_kl_fwd_twin_kernel<float>), grid, block, 0, st, t, B, seed, step, sample_offset, dyn);
  else return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

int svk_reparam_kl_fwd2(const float* pre, const float* bias_mean, const float* bias_sd, const float* eps,
                        float* eps_out, float* z_mean, float* z_sig, float* z, void* z_lp, int z_dtype, int ldz,
                        int z_col, float* kl, int B, int L, uint64_t seed, uint64_t step, int stream_id,
                        int64_t sample_offset, hipStream_t st, const SvDynArgs* dyn) {
  if (!pre || !bias_mean || !bias_sd || !z_mean || !z_sig || !z || !z_lp || !kl || B <= 0 || L <= 0) return SV_E_BADARG;
  dim3 grid((B + 3) / 4), block(256);
  if (z_dtype == SV_BF16)
    hipLaunchKernelGGL((reparam_kl_fwd_kernel<bf16_t>), grid, block, 0, st, pre, bias_mean, bias_sd, eps, eps_out,
                       z_mean, z_sig, z, (bf16_t*)z_lp, ldz, z_col, kl, B, L, seed, step, stream_id, sample_offset, dyn);
  else if (z_dtype == SV_F32)
    hipLaunchKernelGGL((reparam_kl_fwd_kernel<float>), grid, block, 0, st, pre, bias_mean, bias_sd, eps, eps_out,
                       z_mean, z_sig, z, (float*)z_lp, ldz, z_col, kl, B, L, seed, step, stream_id, sample_offset, dyn);
  else
    return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_reparam_kl_fwd(const float* pre, const float* bias, const float* eps, float* eps_out,
                                 float* z_mean, float* z_sig, float* z, void* z_lp, int32_t z_dtype,
                                 int32_t ldz, int32_t z_col, float* kl, int32_t B, int32_t L,
                                 uint64_t seed, uint64_t step, int32_t stream_id,
                                 int64_t sample_offset, void* stream) {
  if (!bias) return SV_E_BADARG;
  return svk_reparam_kl_fwd2(pre, bias, bias + L, eps, eps_out, z_mean, z_sig, z, z_lp, z_dtype, ldz, z_col, kl, B,
                             L, seed, step, stream_id, sample_offset, (hipStream_t)stream);
}

template <typename TG>
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(
    const float* __restrict__ dz, int ld_dz, const float* __restrict__ dz2, int ld_dz2,
    const float* __restrict__ z_mean, const float* __restrict__ z_sig, const float* __restrict__ eps,
    float kl_scale, TG* __restrict__ g_pre, int B, int L) {
  const int64_t total = (int64_t)B * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), j = (int)(i - (int64_t)b * L);
    float g = dz[(int64_t)b * ld_dz + j];
    if (dz2) g += dz2[(int64_t)b * ld_dz2 + j];
    const float mu = z_mean[i], sg = z_sig[i], e = eps[i];
    const float dmu = g + kl_scale * mu;
    const float dsg = g * e + kl_scale * (sg - 1.f / sg);
    const float dpre = dsg * (1.f - expf(-sg));          // softplus'(pre) = 1 - exp(-softplus(pre))
    g_pre[(int64_t)b * 2 * L + j] = from_f32<TG>(dmu);
    g_pre[(int64_t)b * 2 * L + L + j] = from_f32<TG>(dpre);
  }
}

struct ReparamBwdArgs {
  const float *dz, *dz2, *z_mean, *z_sig, *eps; void* g_pre; int ld_dz, ld_dz2, L;
  int S, S2; int64_t stride, stride2;      // S > 0: dz / dz2 are the first of S / S2 K-slice slabs of d1's input gradient, summed here in slice order
};
struct ReparamBwdTwin { ReparamBwdArgs a[2]; };
template <typename TG>
__global__ __launch_bounds__(256) void reparam_kl_bwd_twin_kernel(const ReparamBwdTwin t, float kl_scale, int B) {
  const ReparamBwdArgs& g = t.a[blockIdx.y];
  const int L = g.L;
  const int64_t total = (int64_t)B * L;
  TG* __restrict__ g_pre = (TG*)g.g_pre;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), j = (int)(i - (int64_t)b * L);
    float gg;
    if (g.S > 0) {                                         // as nt_slab_reduce_kernel (bitwise the two-launch result)
      gg = 0.f;
      const float* __restrict__ q = g.dz + (int64_t)b * g.ld_dz + j;
#pragma unroll 8
      for (int k = 0; k < g.S; ++k) gg += q[(int64_t)k * g.stride];
      if (g.dz2) {
        float g2 = 0.f;
        const float* __restrict__ q2 = g.dz2 + (int64_t)b * g.ld_dz2 + j;
#pragma unroll 8
        for (int k = 0; k < g.S2; ++k) g2 += q2[(int64_t)k * g.stride2];
        gg += g2;
      }
    } else {
      gg = g.dz[(int64_t)b * g.ld_dz + j];
      if (g.dz2) gg += g.dz2[(int64_t)b * g.ld_dz2 + j];
    }
    const float mu = g.z_mean[i], sg = g.z_sig[i], e = g.eps[i];
    const float dmu = gg + kl_scale * mu;
    const float dsg = gg * e + kl_scale * (sg - 1.f / sg);
    const float dpre = dsg * (1.f - expf(-sg));          // as reparam_kl_bwd_kernel
    g_pre[(int64_t)b * 2 * L + j] = from_f32<TG>(dmu);
    g_pre[(int64_t)b * 2 * L + L + j] = from_f32<TG>(dpre);
  }
}
int svk_reparam_kl_bwd_twin(const float* const* dz, const int* ld_dz, const float* const* dz2, const int* ld_dz2,
                            const float* const* z_mean, const float* const* z_sig, const float* const* eps, float kl_scale,
                            void* const* g_pre, int g_dtype, int B, const int* L, hipStream_t st,
                            const int* S, const int64_t* stride, const int* S2, const int64_t* stride2) {
  ReparamBwdTwin t;
  int Lmax = 0;
  for (int e = 0; e < 2; ++e) {
    t.a[e] = {dz[e], dz2[e], z_mean[e], z_sig[e], eps[e], g_pre[e], ld_dz[e], ld_dz2[e], L[e],
              S ? S[e] : 0, S ? S2[e] : 0, S ? stride[e] : 0, S ? stride2[e] : 0};
    Lmax = L[e] > Lmax ? L[e] : Lmax;
  }
  dim3 grid((unsigned)(((int64_t)B * Lmax + 255) / 256), 2), block(256);
  if (g_dtype == SV_BF16) hipLaunchKernelGGL((reparam_kl_bwd_twin_kernel<bf16_t>), grid, block, 0, st, t, kl_scale, B);
  else if (g_dtype == SV_F32) hipLaunchKernelGGL((reparam_kl_bwd_twin_kernel<float>), grid, block, 0, st, t, kl_scale, B);
  else return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_reparam_kl_bwd(const float* dz, int32_t ld_dz, const float* dz2, int32_t ld_dz2,
                                 const float* z_mean, const float* z_sig, const float* eps,
                                 float kl_scale, void* g_pre, int32_t g_dtype, int32_t B, int32_t L,
                                 void* stream) {
  if (!dz || !z_mean || !z_sig || !eps || !g_pre || B <= 0 || L <= 0) return SV_E_BADARG;
  const int64_t total = (int64_t)B * L;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (g_dtype == SV_BF16)
    hipLaunchKernelGGL((reparam_kl_bwd_kernel<bf16_t>), grid, block, 0, st, dz, ld_dz, dz2, ld_dz2,
                       z_mean, z_sig, eps, kl_scale, (bf16_t*)g_pre, B, L);
  else if (g_dtype == SV_F32)
    hipLaunchKernelGGL((reparam_kl_bwd_kernel<float>), grid, block, 0, st, dz, ld_dz, dz2, ld_dz2,
                       z_mean, z_sig, eps, kl_scale, (float*)g_pre, B, L);
  else
    return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ============================================================================ K14 Keras Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n4, int64_t n, float alpha, float omb1,
                                                   float omb2, float eps, float gscale,
                                                   const SvDynArgs* __restrict__ dyn) {
  if (dyn) alpha = dyn->adam_alpha;   // captured step (graph replay): the bias-corrected rate of THIS iteration
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = ((float4*)p)[i], gg = ((const float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
#define SV_ADAM1(c)                               \
  {                                               \
    const float gc = gg.c * gscale;               \
    mm.c = mm.c + (gc - mm.c) * omb1;             \
    vv.c = vv.c + (gc * gc - vv.c) * omb2;        \
    pp.c = pp.c - alpha * mm.c / (sqrtf(vv.c) + eps); \
  }
    SV_ADAM1(x) SV_ADAM1(y) SV_ADAM1(z) SV_ADAM1(w)
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gc = g[i] * gscale;
      float mi = m[i], vi = v[i];
      mi = mi + (gc - mi) * omb1;
      vi = vi + (gc * gc - vi) * omb2;
      p[i] = p[i] - alpha * mi / (sqrtf(vi) + eps);
      m[i] = mi; v[i] = vi;
    }
  }
}

double svk_adam_alpha(float lr, float beta1, float beta2, int64_t t) {
  return (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
}

int svk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int64_t t, float grad_scale, const SvDynArgs* dyn, hipStream_t st) {
  if (!p || !g || !m || !v || n <= 0 || t <= 0) return SV_E_BADARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return SV_E_BADARG;
  const double alpha = svk_adam_alpha(lr, beta1, beta2, t);
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v,
                     n4, n, (float)alpha, 1.f - beta1, 1.f - beta2, eps, grad_scale, dyn);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                            float beta1, float beta2, float eps, int64_t t, float grad_scale,
                            void* stream) {
  return svk_adam_step(p, g, m, v, n, lr, beta1, beta2, eps, t, grad_scale, nullptr, (hipStream_t)stream);
}

// ---- Adam with Keras `clipnorm` (spair/main.py:109: Adam(..., clipnorm=1.0)): every gradient TENSOR is scaled by
// clipnorm / max(||g||_2, clipnorm) (tf.clip_by_norm) before the update.  Pass 1: 256 partial sums of squares per tensor
// (fixed strides); pass 2: a workgroup column per tensor adds them in a fixed order and applies the update -- deterministic.
#define SV_CLIP_PARTS 256
// element range [lo, hi) of a tensor split into an unaligned head, a float4 body and a tail (the flat buffers are 16-byte aligned, so
// the split is the same for p, g, m and v)
__device__ __forceinline__ void clip_split(int64_t lo, int64_t hi, int64_t& a, int64_t& b) {
  a = min((lo + 3) & ~(int64_t)3, hi);
  b = max(a, hi & ~(int64_t)3);
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, const int64_t* __restrict__ off,
                                                    float* __restrict__ parts, float gscale) {
  const int t = blockIdx.y;
  const int64_t lo = off[t], hi = off[t + 1];
  int64_t a, b;
  clip_split(lo, hi, a, b);
  float s = 0.f;
  const float4* g4 = (const float4*)(g + a);
  const int64_t n4 = (b - a) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)SV_CLIP_PARTS * 256) {
    const float4 v = g4[i];
    const float x = v.x * gscale, y = v.y * gscale, z = v.z * gscale, w = v.w * gscale;
    s += (x * x + y * y) + (z * z + w * w);
  }
  if (blockIdx.x == 0) {                                   // head and tail (at most 3 elements each)
    for (int64_t i = lo + threadIdx.x; i < a; i += 256) { const float v = g[i] * gscale; s += v * v; }
    for (int64_t i = b + threadIdx.x; i < hi; i += 256) { const float v = g[i] * gscale; s += v * v; }
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[(int64_t)t * SV_CLIP_PARTS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const int64_t* __restrict__ off,
                                                        const float* __restrict__ parts, float clipnorm, float alpha,
                                                        float omb1, float omb2, float eps, float gscale,
                                                        const float* __restrict__ alpha_dev) {
  const int t = blockIdx.y;
  if (alpha_dev) alpha = *alpha_dev;                           // hipGraph replay: the bias-corrected step size of iteration t
  const int64_t lo = off[t], hi = off[t + 1];
  if (lo + (int64_t)blockIdx.x * 1024 >= hi && blockIdx.x) return;   // nothing for this workgroup (small tensors): skip the partial sums too
  float ss = 0.f;
  for (int k = 0; k < SV_CLIP_PARTS; ++k) ss += parts[(int64_t)t * SV_CLIP_PARTS + k];   // fixed order: deterministic
  // (clipnorm >= 1e37: "no clipping" -- the plain Keras-Adam update whatever the norm is; as a ratio an infinite norm would zero the tensor and hide it)
  const float sc = clipnorm >= 1e37f ? gscale : gscale * clipnorm / fmaxf(sqrtf(ss), clipnorm);
  int64_t a, b;
  clip_split(lo, hi, a, b);
  auto upd = [&](float gi, float& pi, float& mi, float& vi) {
    const float gc = gi * sc;
    mi = mi + (gc - mi) * omb1;
    vi = vi + (gc * gc - vi) * omb2;
    pi = pi - alpha * mi / (sqrtf(vi) + eps);
  };
  const int64_t n4 = (b - a) >> 2;
  float4* p4 = (float4*)(p + a); float4* m4 = (float4*)(m + a); float4* v4 = (float4*)(v + a);
  const float4* g4 = (const float4*)(g + a);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    upd(gg.x, pp.x, mm.x, vv.x); upd(gg.y, pp.y, mm.y, vv.y); upd(gg.z, pp.z, mm.z, vv.z); upd(gg.w, pp.w, mm.w, vv.w);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = lo + threadIdx.x; i < a; i += 256) { float pi = p[i], mi = m[i], vi = v[i]; upd(g[i], pi, mi, vi); p[i] = pi; m[i] = mi; v[i] = vi; }
    for (int64_t i = b + threadIdx.x; i < hi; i += 256) { float pi = p[i], mi = m[i], vi = v[i]; upd(g[i], pi, mi, vi); p[i] = pi; m[i] = mi; v[i] = vi; }
  }
}

// ---- the same over SEPARATE gradient tensors (what an autograd engine hands back): their addresses travel by value in the kernel
// arguments (<= SV_CLIP_MAX_TENSORS), so no flat copy of the gradients is made (a torch.cat of 31.9 M floats was 98 us of the 3.6 ms
// SPLIT-SPAIR step) and a captured hipGraph keeps working as long as the tensors keep their addresses.
#define SV_CLIP_MAX_TENSORS 128
struct GradPtrs { const float* g[SV_CLIP_MAX_TENSORS]; };
__global__ __launch_bounds__(256) void sumsq_ptrs_kernel(const GradPtrs P, const int64_t* __restrict__ off, float* __restrict__ parts, float gscale) {
  const int t = blockIdx.y;
  const int64_t n = off[t + 1] - off[t];
  const float* __restrict__ g = P.g[t];                       // 16-byte aligned (its own allocation)
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)SV_CLIP_PARTS * 256) {
    const float4 v = ((const float4*)g)[i];
    const float x = v.x * gscale, y = v.y * gscale, z = v.z * gscale, w = v.w * gscale;
    s += (x * x + y * y) + (z * z + w * w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) { const float v = g[i] * gscale; s += v * v; }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[(int64_t)t * SV_CLIP_PARTS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void adam_clip_ptrs_kernel(float* __restrict__ p, const GradPtrs P, float* __restrict__ m, float* __restrict__ v,
                                                             const int64_t* __restrict__ off, const float* __restrict__ parts, float clipnorm,
                                                             float alpha, float omb1, float omb2, float eps, float gscale,
                                                             const float* __restrict__ alpha_dev) {
  const int t = blockIdx.y;
  if (alpha_dev) alpha = *alpha_dev;
  const int64_t lo = off[t], hi = off[t + 1];
  if (lo + (int64_t)blockIdx.x * 1024 >= hi && blockIdx.x) return;
  float ss = 0.f;
  for (int k = 0; k < SV_CLIP_PARTS; ++k) ss += parts[(int64_t)t * SV_CLIP_PARTS + k];
  // (clipnorm >= 1e37: "no clipping" -- the plain Keras-Adam update whatever the norm is; as a ratio an infinite norm would zero the tensor and hide it)
  const float sc = clipnorm >= 1e37f ? gscale : gscale * clipnorm / fmaxf(sqrtf(ss), clipnorm);
  const float* __restrict__ g = P.g[t] - lo;                  // g[i], i in [lo, hi): the flat index of the variable buffers
  int64_t a, b;
  clip_split(lo, hi, a, b);
  auto upd = [&](float gi, float& pi, float& mi, float& vi) {
    const float gc = gi * sc;
    mi = mi + (gc - mi) * omb1;
    vi = vi + (gc * gc - vi) * omb2;
    pi = pi - alpha * mi / (sqrtf(vi) + eps);
  };
  const int64_t n4 = (b - a) >> 2;
  float4* p4 = (float4*)(p + a); float4* m4 = (float4*)(m + a); float4* v4 = (float4*)(v + a);
  const bool galigned = ((a - lo) & 3) == 0;                   // the gradient tensor starts at its own 16-byte boundary
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i], gg;
    const float* gi = g + a + 4 * i;
    if (galigned) gg = *(const float4*)gi;
    else gg = make_float4(gi[0], gi[1], gi[2], gi[3]);
    upd(gg.x, pp.x, mm.x, vv.x); upd(gg.y, pp.y, mm.y, vv.y); upd(gg.z, pp.z, mm.z, vv.z); upd(gg.w, pp.w, mm.w, vv.w);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = lo + threadIdx.x; i < a; i += 256) { float pi = p[i], mi = m[i], vi = v[i]; upd(g[i], pi, mi, vi); p[i] = pi; m[i] = mi; v[i] = vi; }
    for (int64_t i = b + threadIdx.x; i < hi; i += 256) { float pi = p[i], mi = m[i], vi = v[i]; upd(g[i], pi, mi, vi); p[i] = pi; m[i] = mi; v[i] = vi; }
  }
}

extern "C" int sv_adam_step_clipnorm_ptrs(float* p, const float* const* grads, float* m, float* v, const int64_t* tensor_off,
                                          int32_t n_tensors, float* norm_ws, float clipnorm, float lr, float beta1, float beta2, float eps,
                                          int64_t t, const float* alpha_dev, float grad_scale, void* stream) {
  if (!p || !grads || !m || !v || !tensor_off || !norm_ws || n_tensors < 1 || n_tensors > SV_CLIP_MAX_TENSORS || t <= 0 || !(clipnorm > 0.f))
    return SV_E_BADARG;
  GradPtrs P;
  for (int i = 0; i < SV_CLIP_MAX_TENSORS; ++i) P.g[i] = i < n_tensors ? grads[i] : nullptr;
  for (int i = 0; i < n_tensors; ++i)
    if (!P.g[i] || ((uintptr_t)P.g[i] & 15)) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const double alpha = svk_adam_alpha(lr, beta1, beta2, t);
  hipLaunchKernelGGL(sumsq_ptrs_kernel, dim3(SV_CLIP_PARTS, n_tensors), dim3(256), 0, st, P, tensor_off, norm_ws, grad_scale);
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_clip_ptrs_kernel, dim3(256, n_tensors), dim3(256), 0, st, p, P, m, v, tensor_off, norm_ws, clipnorm, (float)alpha,
                     1.f - beta1, 1.f - beta2, eps, grad_scale, alpha_dev);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" float sv_adam_alpha(float lr, float beta1, float beta2, int64_t t) { return (float)svk_adam_alpha(lr, beta1, beta2, t); }

extern "C" int sv_adam_step_clipnorm(float* p, const float* g, float* m, float* v, const int64_t* tensor_off, int32_t n_tensors,
                                     float* norm_ws, float clipnorm, float lr, float beta1, float beta2, float eps, int64_t t,
                                     float grad_scale, void* stream) {
  return sv_adam_step_clipnorm_dyn(p, g, m, v, tensor_off, n_tensors, norm_ws, clipnorm, lr, beta1, beta2, eps, t, nullptr, grad_scale, stream);
}

extern "C" int sv_adam_step_clipnorm_dyn(float* p, const float* g, float* m, float* v, const int64_t* tensor_off, int32_t n_tensors,
                                         float* norm_ws, float clipnorm, float lr, float beta1, float beta2, float eps, int64_t t,
                                         const float* alpha_dev, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !tensor_off || !norm_ws || n_tensors < 1 || t <= 0 || !(clipnorm > 0.f)) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const double alpha = svk_adam_alpha(lr, beta1, beta2, t);
  hipLaunchKernelGGL(sumsq_kernel, dim3(SV_CLIP_PARTS, n_tensors), dim3(256), 0, st, g, tensor_off, norm_ws, grad_scale);
  SV_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_clip_kernel, dim3(256, n_tensors), dim3(256), 0, st, p, g, m, v, tensor_off, norm_ws, clipnorm,
                     (float)alpha, 1.f - beta1, 1.f - beta2, eps, grad_scale, alpha_dev);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

__global__ void set_dyn_kernel(SvDynArgs* dyn, uint64_t seed, uint64_t step, int64_t sample_offset, float adam_alpha) {
  dyn->seed = seed; dyn->step = step; dyn->sample_offset = sample_offset; dyn->adam_alpha = adam_alpha; dyn->pad = 0.f;
}
int svk_set_dyn(SvDynArgs* dyn, uint64_t seed, uint64_t step, int64_t sample_offset, float adam_alpha, hipStream_t st) {
  hipLaunchKernelGGL(set_dyn_kernel, dim3(1), dim3(1), 0, st, dyn, seed, step, sample_offset, adam_alpha);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ============================================================================ K10a bilinear 2x
// tf.image.resize (half-pixel centres): out[2i] = .25 in[i-1] + .75 in[i]; out[2i+1] = .75 in[i] + .25 in[i+1]
// with clamped neighbours.  One thread per (output pixel, 16-B channel piece).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                             int B, int H, int W, int C) {
  constexpr int EPP = ElemTraits<T>::EPP;
  const int cp = C / EPP;
  const int64_t total = (int64_t)B * 2 * H * 2 * W * cp;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cp);
    int64_t t = idx / cp;
    const int ox = (int)(t % (2 * W)); t /= 2 * W;
    const int oy = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    const int iy = oy >> 1, ix = ox >> 1;
    const int y0 = (oy & 1) ? iy : max(iy - 1, 0), y1 = (oy & 1) ? min(iy + 1, H - 1) : iy;
    const int x0 = (ox & 1) ? ix : max(ix - 1, 0), x1 = (ox & 1) ? min(ix + 1, W - 1) : ix;
    const float fy = (oy & 1) ? 0.25f : 0.75f;   // weight of the second (y1) sample
    const float fx = (ox & 1) ? 0.25f : 0.75f;
    const T* base = in + (int64_t)b * H * W * C + c * EPP;
    constexpr int NP = Piece<T>::NP;
    f32x2 v00[NP], v01[NP], v10[NP], v11[NP], r[NP];
    Piece<T>::unpack(*(const uint4*)(base + ((int64_t)y0 * W + x0) * C), v00);
    Piece<T>::unpack(*(const uint4*)(base + ((int64_t)y0 * W + x1) * C), v01);
    Piece<T>::unpack(*(const uint4*)(base + ((int64_t)y1 * W + x0) * C), v10);
    Piece<T>::unpack(*(const uint4*)(base + ((int64_t)y1 * W + x1) * C), v11);
#pragma unroll
    for (int e = 0; e < NP; ++e) r[e] = lerp2(lerp2(v00[e], v01[e], fx), lerp2(v10[e], v11[e], fx), fy);   // as tile_stage.hip.h blend2x2
    const uint4 rp = Piece<T>::pack(r);
    *(uint4*)(out + (((int64_t)b * 2 * H + oy) * 2 * W + ox) * C + c * EPP) = rp;
  }
}

// adjoint: g_lo[i,j] = sum_{a,b in -1..2} wy[a] wx[b] g_hi[clamp(2i+a), clamp(2j+b)], w = (.25,.75,.75,.25);
// then the ReLU mask of the low-res producer (y_lo > 0).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ g_hi, const T* __restrict__ mask,
                                                             T* __restrict__ g_lo, int B, int H, int W, int C) {
  constexpr int EPP = ElemTraits<T>::EPP;
  const int cp = C / EPP;
  const int64_t total = (int64_t)B * H * W * cp;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cp);
    int64_t t = idx / cp;
    const int j = (int)(t % W); t /= W;
    const int i = (int)(t % H);
    const int b = (int)(t / H);
    float acc[EPP];
#pragma unroll
    for (int e = 0; e < EPP; ++e) acc[e] = 0.f;
    const T* base = g_hi + (int64_t)b * 4 * H * W * C + c * EPP;
    if constexpr (sizeof(T) == 2) {                          // bf16: column pairs through v_dot2c_f32_bf16 (adj2x_row_bf16, common.hip.h)
      const int64_t xo[4] = {(int64_t)max(2 * j - 1, 0) * C, (int64_t)2 * j * C, (int64_t)(2 * j + 1) * C, (int64_t)min(2 * j + 2, 2 * W - 1) * C};
#pragma unroll
      for (int a = -1; a <= 2; ++a) {
        const int oy = min(max(2 * i + a, 0), 2 * H - 1);
        const T* row = base + (int64_t)oy * 2 * W * C;
        adj2x_row_bf16(acc, *(const uint4*)(row + xo[0]), *(const uint4*)(row + xo[1]), *(const uint4*)(row + xo[2]), *(const uint4*)(row + xo[3]),
                       a == -1 || a == 2);
      }
    } else {
#pragma unroll
    for (int a = -1; a <= 2; ++a) {
      const int oy = min(max(2 * i + a, 0), 2 * H - 1);
      const float wy = (a == -1 || a == 2) ? 0.25f : 0.75f;
#pragma unroll
      for (int d = -1; d <= 2; ++d) {
        const int ox = min(max(2 * j + d, 0), 2 * W - 1);
        const float w = wy * ((d == -1 || d == 2) ? 0.25f : 0.75f);
        T v[EPP];
        *(uint4*)v = *(const uint4*)(base + ((int64_t)oy * 2 * W + ox) * C);
#pragma unroll
        for (int e = 0; e < EPP; ++e) acc[e] += w * to_f32(v[e]);
      }
    }
    }
    const int64_t o = (((int64_t)b * H + i) * W + j) * C + c * EPP;
    T r[EPP];
    if (mask) {
      T mv[EPP];
      *(uint4*)mv = *(const uint4*)(mask + o);
#pragma unroll
      for (int e = 0; e < EPP; ++e) r[e] = from_f32<T>(to_f32(mv[e]) > 0.f ? acc[e] : 0.f);
    } else {
#pragma unroll
      for (int e = 0; e < EPP; ++e) r[e] = from_f32<T>(acc[e]);
    }
    *(uint4*)(g_lo + o) = *(uint4*)r;
  }
}

// The same adjoint at fp32 with the hi-res window in registers: a thread walks RB consecutive low-res rows of one (image, column, 4-channel
// piece); rows 2i+1 and 2i+2 of output row i are rows 2(i+1)-1 and 2(i+1) of the next one, so every step loads two hi-res rows (8 x 16 B)
// instead of four and no hi-res row is fetched by two different waves except at a band's edge.  The plain kernel left the row overlap to
// the caches: 1.67 x the algorithmic bytes from HBM on the fp32 step's three launches (PMC, profiles/r04_i_f32_traffic.json), 2.5 TB/s.
// Same products, same summation order (a outer, d inner) as upsample2x_bwd_kernel<float>: bitwise the same result.
template <int RB>
__global__ __launch_bounds__(256) void upsample2x_bwd_rows_kernel(const float* __restrict__ g_hi, const float* __restrict__ mask,
                                                                  float* __restrict__ g_lo, int B, int H, int W, int C) {
  const int cp = C / 4, nb = H / RB;
  const int64_t total = (int64_t)B * nb * W * cp;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cp);
    int64_t t = idx / cp;
    const int j = (int)(t % W); t /= W;
    const int band = (int)(t % nb);
    const int b = (int)(t / nb);
    const float* base = g_hi + (int64_t)b * 4 * H * W * C + c * 4;
    const int64_t xo[4] = {(int64_t)max(2 * j - 1, 0) * C, (int64_t)2 * j * C, (int64_t)(2 * j + 1) * C, (int64_t)min(2 * j + 2, 2 * W - 1) * C};
    const int i0 = band * RB;
    float4 win[4][4];
    {
      const float* r0 = base + (int64_t)max(2 * i0 - 1, 0) * 2 * W * C;
      const float* r1 = base + (int64_t)(2 * i0) * 2 * W * C;
#pragma unroll
      for (int d = 0; d < 4; ++d) { win[0][d] = *(const float4*)(r0 + xo[d]); win[1][d] = *(const float4*)(r1 + xo[d]); }
    }
#pragma unroll
    for (int s = 0; s < RB; ++s) {
      const int i = i0 + s;
      const float* r2 = base + (int64_t)(2 * i + 1) * 2 * W * C;
      const float* r3 = base + (int64_t)min(2 * i + 2, 2 * H - 1) * 2 * W * C;
#pragma unroll
      for (int d = 0; d < 4; ++d) { win[2][d] = *(const float4*)(r2 + xo[d]); win[3][d] = *(const float4*)(r3 + xo[d]); }
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float wy = (a == 0 || a == 3) ? 0.25f : 0.75f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float w = wy * ((d == 0 || d == 3) ? 0.25f : 0.75f);
          const float4 v = win[a][d];
          acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w;
        }
      }
      const int64_t o = (((int64_t)b * H + i) * W + j) * C + c * 4;
      float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (mask) {
        const float4 mv = *(const float4*)(mask + o);
        r.x = mv.x > 0.f ? r.x : 0.f; r.y = mv.y > 0.f ? r.y : 0.f; r.z = mv.z > 0.f ? r.z : 0.f; r.w = mv.w > 0.f ? r.w : 0.f;
      }
      *(float4*)(g_lo + o) = r;
#pragma unroll
      for (int d = 0; d < 4; ++d) { win[0][d] = win[2][d]; win[1][d] = win[3][d]; }
    }
  }
}

static inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 256 * 32) b = 256 * 32;
  return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int sv_upsample2x_fwd(const void* in, void* out, int32_t dtype, int32_t B, int32_t H, int32_t W,
                                 int32_t C, void* stream) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SV_BF16) {
    if (C % 8) return SV_E_UNSUPPORTED;
    hipLaunchKernelGGL((upsample2x_fwd_kernel<bf16_t>), dim3(grid_for((int64_t)B * 4 * H * W * (C / 8))),
                       dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, B, H, W, C);
  } else if (dtype == SV_F32) {
    if (C % 4) return SV_E_UNSUPPORTED;
    hipLaunchKernelGGL((upsample2x_fwd_kernel<float>), dim3(grid_for((int64_t)B * 4 * H * W * (C / 4))),
                       dim3(256), 0, st, (const float*)in, (float*)out, B, H, W, C);
  } else
    return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

extern "C" int sv_upsample2x_bwd(const void* g_hi, const void* y_lo_mask, void* g_lo, int32_t dtype,
                                 int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!g_hi || !g_lo || B <= 0 || H <= 0 || W <= 0 || C <= 0) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SV_BF16) {
    if (C % 8) return SV_E_UNSUPPORTED;
    hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t>), dim3(grid_for((int64_t)B * H * W * (C / 8))),
                       dim3(256), 0, st, (const bf16_t*)g_hi, (const bf16_t*)y_lo_mask, (bf16_t*)g_lo, B, H, W, C);
  } else if (dtype == SV_F32) {
    if (C % 4) return SV_E_UNSUPPORTED;
    static const bool plain = getenv("SV_UPS_BWD_PLAIN") != nullptr;            // A/B knob: one thread per output, sixteen loads each
    const int64_t band_threads = (int64_t)B * (H / 8) * W * (C / 4);
    if (!plain && H % 8 == 0 && band_threads >= 256 * 256)                      // (small launches keep the per-output form: more threads)
      hipLaunchKernelGGL((upsample2x_bwd_rows_kernel<8>), dim3(grid_for(band_threads)), dim3(256), 0, st, (const float*)g_hi,
                         (const float*)y_lo_mask, (float*)g_lo, B, H, W, C);
    else
      hipLaunchKernelGGL((upsample2x_bwd_kernel<float>), dim3(grid_for((int64_t)B * H * W * (C / 4))),
                         dim3(256), 0, st, (const float*)g_hi, (const float*)y_lo_mask, (float*)g_lo, B, H, W, C);
  } else
    return SV_E_BADARG;
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ============================================================================ step-internal helpers
// images6 fp32 -> two 8-channel (zero padded) low-precision NHWC tensors (vae/model.py:190 split).
template <typename T>
__global__ __launch_bounds__(256) void split_pad_kernel(const float* __restrict__ images6, T* __restrict__ x8,
                                                        T* __restrict__ xh8, int64_t npix) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* s = images6 + i * 6;
    const float2 a = *(const float2*)s, b = *(const float2*)(s + 2), c = *(const float2*)(s + 4);
    T u[8], w[8];
    u[0] = from_f32<T>(a.x); u[1] = from_f32<T>(a.y); u[2] = from_f32<T>(b.x);
    w[0] = from_f32<T>(b.y); w[1] = from_f32<T>(c.x); w[2] = from_f32<T>(c.y);
#pragma unroll
    for (int e = 3; e < 8; ++e) { u[e] = from_f32<T>(0.f); w[e] = from_f32<T>(0.f); }
    if constexpr (sizeof(T) == 2) {
      *(uint4*)(x8 + i * 8) = *(uint4*)u;
      *(uint4*)(xh8 + i * 8) = *(uint4*)w;
    } else {
      *(uint4*)(x8 + i * 8) = *(uint4*)u; *(uint4*)(x8 + i * 8 + 4) = *(uint4*)(u + 4);
      *(uint4*)(xh8 + i * 8) = *(uint4*)w; *(uint4*)(xh8 + i * 8 + 4) = *(uint4*)(w + 4);
    }
  }
}

int svk_split_pad(const float* images6, void* x8, void* xh8, int dtype, int64_t npix, hipStream_t st) {
  if (dtype == SV_BF16)
    hipLaunchKernelGGL((split_pad_kernel<bf16_t>), dim3(grid_for(npix)), dim3(256), 0, st, images6,
                       (bf16_t*)x8, (bf16_t*)xh8, npix);
  else
    hipLaunchKernelGGL((split_pad_kernel<float>), dim3(grid_for(npix)), dim3(256), 0, st, images6,
                       (float*)x8, (float*)xh8, npix);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// the five scalars of vae/trainer.py:127-135 (+ total) and the running means of :140-144
// part_x / part_xh != null (the loss was evaluated in the decoder head's epilogue): the per-image NLL sums are formed here from
// the P per-tile partials of each image, in index order (what rowsum_partials_kernel does), and written to nll_x / nll_xh
__global__ __launch_bounds__(256) void finalize_losses_kernel(float* __restrict__ nll_x,
                                                              float* __restrict__ nll_xh,
                                                              const float* __restrict__ kl_x,
                                                              const float* __restrict__ kl_xh, int B,
                                                              float beta, float* __restrict__ losses,
                                                              float* __restrict__ metric_acc, int accumulate,
                                                              const float* __restrict__ part_x,
                                                              const float* __restrict__ part_xh, int P) {
  __shared__ float red[4][4];
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (part_x) {
      float sx = 0.f, sh = 0.f;
      for (int i = 0; i < P; ++i) { sx += part_x[(int64_t)b * P + i]; sh += part_xh[(int64_t)b * P + i]; }
      nll_x[b] = sx; nll_xh[b] = sh;
    }
    a[0] += nll_x[b]; a[1] += nll_xh[b]; a[2] += kl_x[b]; a[3] += kl_xh[b];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = wave_sum(a[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = a[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float s[4];
    for (int k = 0; k < 4; ++k) s[k] = ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])) / (float)B;
    const float total_kl = beta * (s[2] + s[3]);
    losses[0] = s[0];        // x_recon_loss
    losses[1] = s[2];        // x_kl_loss
    losses[2] = s[1];        // x_hat_recon_loss
    losses[3] = s[3];        // x_hat_kl_loss
    losses[4] = total_kl;    // total_kl_loss
    losses[5] = s[0] + s[1] + total_kl;
    if (accumulate) {
      for (int k = 0; k < 5; ++k) metric_acc[k] += losses[k];
      metric_acc[5] += 1.f;
    }
  }
}

int svk_finalize_losses(const float* nll_x, const float* nll_xh, const float* kl_x, const float* kl_xh,
                        int B, float beta, float* losses, float* metric_acc, int accumulate, hipStream_t st,
                        const float* part_x, const float* part_xh, int P) {
  hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(256), 0, st, (float*)nll_x, (float*)nll_xh, kl_x, kl_xh, B,
                     beta, losses, metric_acc, accumulate, part_x, part_xh, P);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
