// Native step executor ("tape") for SPLIT-SPAIR: the train step of spair/trainer.py:136-234 -- model forward (spair/spair.py), loss
// assembly, the adjoint tape.gradient builds, Adam -- as ONE native launch sequence over a caller-owned workspace.
//
// The host (split_vae_amd/spair_native.py, mirroring the reference's classes) records the model once as a list of nodes over fp32
// 2-D tensors [rows, cols | row pitch ld]; sv_tape_run then walks the list forwards and backwards from C++: no Python, no torch
// autograd, no library GEMM between the launches.  Every node has a hand-written forward and adjoint:
//   DENSE      exact-fp32 MFMA GEMMs on the Keras [in, out] kernels as they lie in the variable buffer (dense_f32.hip)
//   CONV       sv_conv2d_nhwc_{fwd,dgrad,wgrad} (the MFMA conv kernels; weight images prepared per step in one launch)
//   UPSAMPLE   sv_upsample2x_{fwd,bwd}
//   UNARY      y[:, yo:yo+n] = f(x[:, xo:xo+n]): copy (concat / slice / tile: `rep` output rows per input row), relu, sigmoid,
//              softplus(x + p0), clamp(p0, p1); in place when x and y are the same columns
//   SAMPLE     z = mean + sig * eps (spair/utils.py:19-24);  LOGITNOISE: the concrete pre-sigmoid sample (:14-17)
//   STN / RENDER / ZPRES / LOSS   stn.hip, spair_render.hip and the per-image loss sums of spair/trainer.py with their gradients
//   NOISE      Philox draws for the tensors the reference fills from tf.random (pinned by the caller in tests)
// Gradient buffers are zeroed once per step; every adjoint ADDS into its inputs' gradients (plain read-modify-write: launches are
// stream-ordered), so fan-out needs no bookkeeping; kernels that can only assign go through a scratch buffer when their target has
// other writers.  The loss nodes add their (weight / B)-scaled gradients during the forward pass: total = sum_i w_i * mean_b(loss_i),
// weights per run (the annealed beta of spair/trainer.py:165-167 moves with the step).
//
// LANES (round 6): a node may name a lane (sv_tape_node::lane, 0 = the caller's stream).  Independent branches of the model -- LG-SPAIR's x-hat / background
// encoders and decoders beside the object pipeline (spair/spair.py:84-104) -- then run on their own HIP streams: the step is ~240 launches of 5-30 us, bound by
// their latency, not by any unit of the chip.  sv_tape_finalize derives every cross-lane dependency from the nodes' tensors (read-after-write, write-after-read,
// write-after-write, and the read-modify-write of every gradient accumulation, in tape order) and sv_tape_run turns them into events: conflicting accesses keep
// the tape's order, so the lanes compute the single-stream step bit for bit (tests/test_gpu_spair_model.py).
#include <string.h>
#include <algorithm>
#include <set>
#include <vector>
#include "common.hip.h"
#include "kernels.h"
#include "conv_geom.h"

int svk_dense_f32_fwd(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int M, int K, int N, int act,
                      int allow_split, hipStream_t st);
int svk_dense_f32_fwd_splits(int M, int K, int N);
int svk_dense_f32_dgrad(const float* dy, int ldy, const float* W, float* dx, int ldx, int M, int K, int N, int accumulate, const float* y_gate,
                        hipStream_t st);
int svk_dense_f32_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dW, float* db, int M, int K, int N, int zeroed_once,
                        const float* y_gate, hipStream_t st);

#define SV_TRY(x)                 \
  do {                            \
    const int rc_ = (x);          \
    if (rc_ != SV_OK) return rc_; \
  } while (0)

namespace {

struct TT {
  int64_t rows; int cols, ld;
  int64_t off, goff;       // float offsets into the activation / gradient regions (goff < 0: no gradient)
  int root;                // views share their root's storage
  int writers;             // adjoint writers of the root's gradient
};

// ---------------------------------------------------------------------------------------------------- pointwise kernels
__device__ __forceinline__ float un_fwd(int op, float v, float p0, float p1) {
  switch (op) {
    case SV_TAPE_RELU: return fmaxf(v, 0.f);
    case SV_TAPE_SIGMOID: return sigmoid_f(v);
    case SV_TAPE_SOFTPLUS: return softplus_f(v + p0);
    case SV_TAPE_CLAMP: return fminf(fmaxf(v, p0), p1);
    case SV_TAPE_SCALE: return v * p0;
    default: return v;
  }
}
// derivative from the OUTPUT value (all five are invertible enough for that)
__device__ __forceinline__ float un_der(int op, float y, float p0, float p1) {
  switch (op) {
    case SV_TAPE_RELU: return y > 0.f ? 1.f : 0.f;
    case SV_TAPE_SIGMOID: return y * (1.f - y);
    case SV_TAPE_SOFTPLUS: return 1.f - __expf(-y);          // softplus'(a) = sigmoid(a) = 1 - exp(-softplus(a))
    case SV_TAPE_CLAMP: return (y > p0 && y < p1) ? 1.f : 0.f;
    case SV_TAPE_SCALE: return p0;
    default: return 1.f;
  }
}
__global__ __launch_bounds__(256) void unary_fwd_kernel(int op, const float* __restrict__ x, int ldx, int xo, float* __restrict__ y, int ldy,
                                                        int yo, int64_t total, int n, int rep, float p0, float p1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / n;
  const int j = (int)(i - r * n);
  y[r * ldy + yo + j] = un_fwd(op, x[(r / rep) * ldx + xo + j], p0, p1);
}
// gx[rx, xo + j] += sum_k gy[rx * rep + k, yo + j] * f'(y[...]);  inplace: g[r, yo + j] *= f'(y[r, yo + j])
__global__ __launch_bounds__(256) void unary_bwd_kernel(int op, float* __restrict__ gx, int ldx, int xo, const float* __restrict__ gy,
                                                        const float* __restrict__ y, int ldy, int yo, int64_t total_x, int n, int rep,
                                                        float p0, float p1, int inplace) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_x) return;
  const int64_t r = i / n;
  const int j = (int)(i - r * n);
  float s = 0.f;
  for (int k = 0; k < rep; ++k) {
    const int64_t o = (r * rep + k) * ldy + yo + j;
    s += gy[o] * un_der(op, y[o], p0, p1);
  }
  if (inplace) gx[r * ldx + xo + j] = s;
  else gx[r * ldx + xo + j] += s;
}
// zero fill of up to SV_TAPE_MAX_ZERO ranges in ONE launch (the outputs of the split-K Dense layers, which add their K slices with atomics): the step's first
// launch instead of one 5-us fill in front of every such layer, inside the dependent chain (15 per LG-SPAIR step)
#define SV_TAPE_MAX_ZERO 32
struct ZeroList { float* p[SV_TAPE_MAX_ZERO]; int64_t start[SV_TAPE_MAX_ZERO + 1]; int n; };       // start: prefix sums of the ranges' float4 counts
__global__ __launch_bounds__(256) void multi_zero_kernel(const ZeroList z) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= z.start[z.n]) return;
  int k = 0;
  for (int q = 1; q < z.n; ++q) k += i >= z.start[q] ? 1 : 0;
  ((float4*)z.p[k])[i - z.start[k]] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// up to 8 independent UNARY nodes in one launch (a tf.concat's column blocks, the activations of a head's column blocks)
#define SV_TAPE_MAX_PARTS 8
struct UPart { const float* x; float* y; const float* gy; float* gx; int ldx, xo, ldy, yo, n, rep, op, inplace; float p0, p1; int64_t start; };
struct UMulti { UPart p[SV_TAPE_MAX_PARTS]; int np; int64_t total; };
__global__ __launch_bounds__(256) void unary_multi_fwd_kernel(const UMulti m) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m.total) return;
  int k = 0;
#pragma unroll
  for (int q = 1; q < SV_TAPE_MAX_PARTS; ++q) k += (q < m.np && i >= m.p[q].start) ? 1 : 0;
  const UPart& u = m.p[k];
  const int64_t l = i - u.start, r = l / u.n;
  const int j = (int)(l - r * u.n);
  u.y[r * u.ldy + u.yo + j] = un_fwd(u.op, u.x[(r / u.rep) * u.ldx + u.xo + j], u.p0, u.p1);
}
__global__ __launch_bounds__(256) void unary_multi_bwd_kernel(const UMulti m) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m.total) return;
  int k = 0;
#pragma unroll
  for (int q = 1; q < SV_TAPE_MAX_PARTS; ++q) k += (q < m.np && i >= m.p[q].start) ? 1 : 0;
  const UPart& u = m.p[k];
  const int64_t l = i - u.start, r = l / u.n;
  const int j = (int)(l - r * u.n);
  float s = 0.f;
  for (int q = 0; q < u.rep; ++q) {
    const int64_t o = (r * u.rep + q) * u.ldy + u.yo + j;
    s += u.gy[o] * un_der(u.op, u.y[o], u.p0, u.p1);
  }
  if (u.inplace) u.gx[r * u.ldx + u.xo + j] = s;
  else u.gx[r * u.ldx + u.xo + j] += s;
}
__global__ __launch_bounds__(256) void sample_fwd_kernel(const float* __restrict__ m, int ldm, int mo, const float* __restrict__ s, int lds_, int so,
                                                         const float* __restrict__ e, int lde, int eo, float* __restrict__ z, int ldz, int zo,
                                                         int64_t total, int n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / n;
  const int j = (int)(i - r * n);
  z[r * ldz + zo + j] = m[r * ldm + mo + j] + s[r * lds_ + so + j] * e[r * lde + eo + j];
}
__global__ __launch_bounds__(256) void sample_bwd_kernel(float* __restrict__ gm, int ldm, int mo, float* __restrict__ gs, int lds_, int so,
                                                         const float* __restrict__ e, int lde, int eo, const float* __restrict__ gz, int ldz, int zo,
                                                         int64_t total, int n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / n;
  const int j = (int)(i - r * n);
  const float g = gz[r * ldz + zo + j];
  gm[r * ldm + mo + j] += g;
  gs[r * lds_ + so + j] += g * e[r * lde + eo + j];
}
// concrete_binary_pre_sigmoid_sample (spair/utils.py:14-17): (logits + log(u + 1e-8) - log(1 - u + 1e-8)) / tau
__global__ __launch_bounds__(256) void logitnoise_fwd_kernel(const float* __restrict__ l, int ldl, int lo, const float* __restrict__ u, int ldu, int uo,
                                                             float* __restrict__ y, int ldy, int yo, int64_t total, int n, float inv_tau) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / n;
  const int j = (int)(i - r * n);
  const float uu = u[r * ldu + uo + j];
  y[r * ldy + yo + j] = (l[r * ldl + lo + j] + (logf(uu + 1e-8f) - logf(1.0f - uu + 1e-8f))) * inv_tau;
}
// dst[r, o + j] += scale * src[r * n + j]
__global__ __launch_bounds__(256) void acc_kernel(float* __restrict__ dst, int ld, int o, const float* __restrict__ src, int64_t total, int n, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / n;
  const int j = (int)(i - r * n);
  dst[r * ld + o + j] += scale * src[i];
}
// fp32 -> bf16 (conv_dtype bf16: the spatial convolutions' operands; round to nearest even), 4 elements per thread
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = *(const float4*)(x + i * 4);
  bf16_t o[4] = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
  *(uint2*)(y + i * 4) = *(uint2*)o;
}
// Philox draws: kind 0 normal(0, std) (Box-Muller), 1 uniform (0, 1]
__global__ __launch_bounds__(256) void noise_kernel(float* __restrict__ y, int ld, int64_t total, int n, int kind, float std, uint64_t seed, uint64_t step,
                                                    uint32_t stream_id) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;       // four values per thread
  if (q * 4 >= total) return;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)step, stream_id ^ (uint32_t)(step >> 32) * 0x9E3779B9u};
  Philox ph(seed);
  ph(c);
  float v[4];
  if (kind == 0) {
    const float r0 = sqrtf(-2.0f * logf(u32_to_unit_open(c[0]))), r1 = sqrtf(-2.0f * logf(u32_to_unit_open(c[2])));
    const float a0 = 6.283185307179586f * u32_to_unit_open(c[1]), a1 = 6.283185307179586f * u32_to_unit_open(c[3]);
    v[0] = r0 * cosf(a0) * std; v[1] = r0 * sinf(a0) * std; v[2] = r1 * cosf(a1) * std; v[3] = r1 * sinf(a1) * std;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = u32_to_unit_open(c[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = q * 4 + k;
    if (i < total) y[(i / n) * ld + (i % n)] = v[k];
  }
}

// ---- per-image loss sums of spair/trainer.py over strided tensors, adding scale * d t / d input into the gradients.
// Image b owns rows [b*R, (b+1)*R) x columns [off, off + n).  tf_safe_log :97-101: log(v + 1e-8), NaN / inf -> -100 without gradient.
__device__ __forceinline__ float safe_log_t(float v, bool& ok) {
  const float lv = logf(v + 1e-8f);
  ok = !(isnan(lv) || isinf(lv));
  return ok ? lv : -100.0f;
}
template <int MODE>
__global__ __launch_bounds__(256) void tape_loss_kernel(const float* __restrict__ a, int lda, int ao, const float* __restrict__ b, int ldb, int bo,
                                                        float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ sums, int R, int n,
                                                        float m2, float s2, float scale) {
  const int64_t r0 = (int64_t)blockIdx.x * R;
  bool ok2;
  const float ls2 = MODE == 2 ? safe_log_t(s2, ok2) : 0.f, inv2 = MODE == 2 ? 1.0f / (2.0f * s2 * s2) : 0.f;
  float acc = 0.f;
  for (int i = threadIdx.x; i < R * n; i += 256) {
    const int rr = i / n, j = i - rr * n;
    const int64_t ia = (r0 + rr) * lda + ao + j, ib = (r0 + rr) * ldb + bo + j;
    const float x = a[ia], y = b[ib];
    float t, da = 0.f, db;
    bool ok0, ok1;
    if (MODE == 0) {                                           // xent_loss :103-104 (a = label, b = prediction)
      const float l0 = safe_log_t(y, ok0), l1 = safe_log_t(1.0f - y, ok1);
      t = -(x * l0 + (1.0f - x) * l1);
      db = -((ok0 ? x / (y + 1e-8f) : 0.f) - (ok1 ? (1.0f - x) / (1.0f - y + 1e-8f) : 0.f));
    } else if (MODE == 1) {                                    // kl_divergence :13-21 (a = mean, b = sig)
      const float lv = safe_log_t(y * y, ok0);
      t = -0.5f * (1.0f + lv - x * x - expf(lv));
      da = x;
      db = ok0 ? -0.5f * (2.0f * y / (y * y + 1e-8f) - 2.0f * y * expf(lv) / (y * y + 1e-8f)) : 0.f;
    } else {                                                   // kl_divergence_two_gauss :23-24 against N(m2, s2)
      const float l1 = safe_log_t(y, ok0);
      const float d = x - m2;
      t = ls2 - l1 + (y * y + d * d) * inv2 - 0.5f;
      da = 2.0f * d * inv2;
      db = -(ok0 ? 1.0f / (y + 1e-8f) : 0.f) + 2.0f * y * inv2;
    }
    acc += t;
    if (MODE != 0 && ga) ga[ia] += scale * da;
    if (gb) gb[ib] += scale * db;
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// means[i] = mean_b sums[i][b]; out[0] = total = sum_i w_i means[i]; out[1 + j] = sum_i report[j][i] means[i]; metric accumulators
#define SV_TAPE_MAX_LOSS 16
struct FinalArgs { float w[SV_TAPE_MAX_LOSS]; float rep[SV_TAPE_MAX_LOSS * SV_TAPE_MAX_LOSS]; int nl, nrep, B, accumulate; };
__global__ __launch_bounds__(64) void tape_final_kernel(const float* __restrict__ sums, float* __restrict__ out, float* __restrict__ metric, const FinalArgs f) {
  __shared__ float means[SV_TAPE_MAX_LOSS];
  const int t = threadIdx.x;
  if (t < f.nl) {
    float s = 0.f;
    for (int b = 0; b < f.B; ++b) s += sums[t * f.B + b];      // fixed order
    means[t] = s / (float)f.B;
  }
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
    for (int i = 0; i < f.nl; ++i) tot += f.w[i] * means[i];
    out[0] = tot;
    if (f.accumulate) { metric[0] += tot; metric[SV_TAPE_MAX_LOSS + 1] += 1.0f; }
  }
  if (t < f.nrep) {
    float v = 0.f;
    for (int i = 0; i < f.nl; ++i) v += f.rep[t * SV_TAPE_MAX_LOSS + i] * means[i];
    out[1 + t] = v;
    if (f.accumulate) metric[1 + t] += v;
  }
  if (t < f.nl) out[1 + SV_TAPE_MAX_LOSS + t] = means[t];
}

inline int nblk(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

struct sv_tape {
  std::vector<TT> tens;
  std::vector<sv_tape_node> nodes;
  struct Extra { sv_conv_desc cd; int64_t wf_off = 0, wd_off = 0; int64_t scratch = -1, scratch2 = -1; int split_fwd = 0; bool has_conv = false; };
  std::vector<Extra> ex;
  std::vector<PrepJob> jobs;
  int prep_blocks = 0;
  int64_t arena_elems = 0, act_floats = 0, grad_floats = 0, scratch_floats = 0;
  int64_t off_jobs = 0, off_arena = 0, off_act = 0, off_grad = 0, off_scratch = 0, off_loss = 0, off_wgrad = -1, ws_bytes = 0;      // off_wgrad: the conv weight gradients' partial-sum slabs (shared: stream-ordered)
  int n_loss = 0, B = 0, dtype = SV_F32;
  float report[SV_TAPE_MAX_LOSS * SV_TAPE_MAX_LOSS];
  int n_report = 0;
  bool finalized = false;
  char* ws = nullptr;
  std::vector<std::pair<int, int64_t>> zero_y;     // (tensor, floats) of the split-K Dense outputs: zeroed by ONE launch at the start of the forward pass
  bool zero_at_start = false;
  // lanes: unit = one node or one UNARY group (units are launched in tape order; a lane's launches are stream-ordered among themselves)
  enum { MAX_LANES = 4 };
  struct Sched {
    std::vector<std::vector<int>> waits;   // [first node of a unit] -> nodes whose event the unit's stream waits for first
    std::vector<char> rec;                 // [last node of a unit] -> record the unit's event behind it
  };
  Sched fs, bs;                            // forward / backward pass
  int nlanes = 1;
  hipStream_t lane_st[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ev_node, ev_wg;      // per node: the unit's cross-lane event; "dY is ready" for a weight gradient sent to lane 1's stream
  ~sv_tape() {
    for (int l = 1; l < MAX_LANES; ++l) {
      if (lane_st[l]) (void)hipStreamSynchronize(lane_st[l]);          // (shared: not destroyed)
      if (ev_join[l]) (void)hipEventDestroy(ev_join[l]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto e : ev_node)
      if (e) (void)hipEventDestroy(e);
    for (auto e : ev_wg)
      if (e) (void)hipEventDestroy(e);
  }

  float* act(int t) const { return (float*)(ws + off_act) + tens[t].off; }
  float* grad(int t) const { return tens[t].goff < 0 ? nullptr : (float*)(ws + off_grad) + tens[t].goff; }
  float* scr(int64_t o) const { return (float*)(ws + off_scratch) + o; }
  // LDS-tile weight gradients (wgrad_tile*.hip) flush per-workgroup slabs here and sum them in a fixed order; SV_TAPE_NO_WGRAD_WS: atomics / im2col
  void* wgrad_ws(int lane = 0) const {                      // one slab region per lane (conv weight gradients of two lanes may be in flight together)
    static const bool off = getenv("SV_TAPE_NO_WGRAD_WS") != nullptr;
    return (off || off_wgrad < 0) ? nullptr : ws + off_wgrad + (int64_t)lane * SV_WGRAD_WS_BYTES;
  }
  float* loss_sums() const { return (float*)(ws + off_loss); }                                   // [n_loss][B]
  float* loss_out() const { return loss_sums() + (int64_t)SV_TAPE_MAX_LOSS * B; }                 // total, reported[16], means[16]
  float* metric() const { return loss_out() + 2 * SV_TAPE_MAX_LOSS + 2; }                         // sums of total / reported, count
  int root(int t) const { return tens[t].root; }
};

extern "C" int sv_tape_create(sv_tape** out, int32_t batch, int32_t conv_dtype) {
  if (!out || batch < 1 || (conv_dtype != SV_F32 && conv_dtype != SV_BF16)) return SV_E_BADARG;
  sv_tape* t = new sv_tape();
  t->B = batch; t->dtype = conv_dtype;
  memset(t->report, 0, sizeof(t->report));
  *out = t;
  return SV_OK;
}
extern "C" void sv_tape_destroy(sv_tape* t) { delete t; }

extern "C" int32_t sv_tape_tensor(sv_tape* t, int64_t rows, int32_t cols, int32_t ld, int32_t need_grad) {
  if (!t || t->finalized || rows < 1 || cols < 1 || ld < cols) return SV_E_BADARG;
  TT x;
  x.rows = rows; x.cols = cols; x.ld = ld;
  const int64_t n = (rows * ld + 63) / 64 * 64;             // 256-B granules
  x.off = t->act_floats; t->act_floats += n;
  x.goff = -1;
  if (need_grad) { x.goff = t->grad_floats; t->grad_floats += n; }
  x.root = (int)t->tens.size(); x.writers = 0;
  t->tens.push_back(x);
  return x.root;
}
extern "C" int32_t sv_tape_view(sv_tape* t, int32_t src, int64_t rows, int32_t cols, int32_t ld) {
  if (!t || t->finalized || src < 0 || src >= (int)t->tens.size() || rows < 1 || cols < 1 || ld < cols) return SV_E_BADARG;
  const TT& s = t->tens[src];
  if (rows * ld > (s.rows * s.ld + 63) / 64 * 64) return SV_E_BADARG;
  TT x = s;
  x.rows = rows; x.cols = cols; x.ld = ld; x.root = s.root;
  t->tens.push_back(x);
  return (int)t->tens.size() - 1;
}

static bool tok(const sv_tape* t, int id, bool opt = false) { return (opt && id < 0) || (id >= 0 && id < (int)t->tens.size()); }

extern "C" int sv_tape_add(sv_tape* t, const sv_tape_node* nd) {
  if (!t || !nd || t->finalized) return SV_E_BADARG;
  sv_tape_node n = *nd;
  sv_tape::Extra e;
  auto T = [&](int id) -> const TT& { return t->tens[id]; };
  switch (n.kind) {
    case SV_TAPE_DENSE: {
      if (!tok(t, n.x) || !tok(t, n.y) || n.w_off < 0) return SV_E_BADARG;
      if (T(n.x).rows != T(n.y).rows || (n.act != SV_ACT_NONE && n.act != SV_ACT_RELU)) return SV_E_BADARG;
      e.split_fwd = svk_dense_f32_fwd_splits((int)T(n.x).rows, T(n.x).cols, T(n.y).cols) > 1;
      break;
    }
    case SV_TAPE_CONV: {
      if (!tok(t, n.x) || !tok(t, n.y) || n.w_off < 0) return SV_E_BADARG;
      sv_conv_desc& c = e.cd;
      memset(&c, 0, sizeof(c));
      c.B = n.B; c.H = n.H; c.W = n.W; c.Cin = n.C; c.Cout = n.Cout; c.KH = c.KW = n.k; c.stride = n.stride;
      c.act = n.act; c.dtype = t->dtype; c.ldx = T(n.x).ld; c.ldy = T(n.y).ld;
      c.y_f32 = t->dtype == SV_BF16 ? 1 : 0;        // bf16 operands, fp32 activations between the layers
      const int rc = svg_check(&c);
      if (rc) return rc;
      if (T(n.x).rows != (int64_t)n.B * n.H * n.W || T(n.y).rows != (int64_t)n.B * svg_oh(&c) * svg_ow(&c)) return SV_E_BADARG;
      if (c.ldy != svg_gdy(&c) || ilog2_exact(c.ldy) < 0 || (c.ldx & 3)) return SV_E_BADARG;    // the input-gradient kernels index dY by r8(Cout), a power of two
      e.has_conv = true;
      break;
    }
    case SV_TAPE_UNARY:
      if (!tok(t, n.x) || !tok(t, n.y) || n.n < 1 || n.rep < 1 || T(n.y).rows != T(n.x).rows * n.rep) return SV_E_BADARG;
      if (n.xo + n.n > T(n.x).ld || n.yo + n.n > T(n.y).ld || n.op < SV_TAPE_COPY || n.op > SV_TAPE_SCALE) return SV_E_BADARG;
      break;
    case SV_TAPE_SAMPLE:      // x = mean, t2 = sig, t3 = eps -> y
      if (!tok(t, n.x) || !tok(t, n.t2) || !tok(t, n.t3) || !tok(t, n.y) || n.n < 1) return SV_E_BADARG;
      break;
    case SV_TAPE_LOGITNOISE:  // x = logits, t2 = u -> y
      if (!tok(t, n.x) || !tok(t, n.t2) || !tok(t, n.y) || n.n < 1 || !(n.p0 > 0.f)) return SV_E_BADARG;
      break;
    case SV_TAPE_UPSAMPLE:    // x [B*H*W, C] -> y [B*2H*2W, C]
      if (!tok(t, n.x) || !tok(t, n.y) || T(n.x).ld != T(n.y).ld || (T(n.x).ld & 3)) return SV_E_BADARG;
      break;
    case SV_TAPE_STN:         // x = img, t2 = z_where [B*Hc*Wc, 4] -> y, t3 = bbox (optional)
      if (!tok(t, n.x) || !tok(t, n.t2) || !tok(t, n.y) || !tok(t, n.t3, true) || T(n.t2).ld != 4) return SV_E_BADARG;
      break;
    case SV_TAPE_RENDER:      // x = obj, t2 = bg (a zero tensor for model 'spair'), t3 = z_depth, t4 = z_pres, t5 = z_pres_logits, t6 = noise (-1) -> y
      if (!tok(t, n.x) || !tok(t, n.t2) || !tok(t, n.t3) || !tok(t, n.t4) || !tok(t, n.t5) || !tok(t, n.t6, true) || !tok(t, n.y)) return SV_E_BADARG;
      if (T(n.t3).ld != 1 || T(n.t4).ld != 1 || T(n.t5).ld != 1 || n.R < 1 || n.R > 16) return SV_E_BADARG;
      break;
    case SV_TAPE_ZPRES:       // x = z_pres, t2 = logits, t3 = pre_sigmoid; p0 = tau; dyn_idx -> prior_prob
      if (!tok(t, n.x) || !tok(t, n.t2) || !tok(t, n.t3) || n.loss_idx < 0 || n.loss_idx >= SV_TAPE_MAX_LOSS) return SV_E_BADARG;
      if (T(n.x).ld != 1 || T(n.t2).ld != 1 || T(n.t3).ld != 1 || n.R < 1 || n.R > 16 || n.dyn_idx < 0 || n.dyn_idx >= 8) return SV_E_BADARG;
      break;
    case SV_TAPE_LOSS:        // x = a (+xo), t2 = b (+o2); mode, R rows per image, n columns
      if (!tok(t, n.x) || !tok(t, n.t2) || n.loss_idx < 0 || n.loss_idx >= SV_TAPE_MAX_LOSS || n.mode < 0 || n.mode > 2 || n.R < 1 || n.n < 1) return SV_E_BADARG;
      break;
    case SV_TAPE_NOISE:       // y <- Philox (op 0 normal * p0, 1 uniform); skipped when the caller pins the tensor
      if (!tok(t, n.y)) return SV_E_BADARG;
      break;
    default: return SV_E_BADARG;
  }
  if (n.kind == SV_TAPE_ZPRES || n.kind == SV_TAPE_LOSS) t->n_loss = n.loss_idx + 1 > t->n_loss ? n.loss_idx + 1 : t->n_loss;
  if (n.lane < 0 || n.lane >= sv_tape::MAX_LANES) return SV_E_BADARG;
  t->nodes.push_back(n);
  t->ex.push_back(e);
  return SV_OK;
}

extern "C" int sv_tape_set_report(sv_tape* t, const float* matrix, int32_t n_report) {
  if (!t || !matrix || n_report < 0 || n_report > SV_TAPE_MAX_LOSS) return SV_E_BADARG;
  memcpy(t->report, matrix, sizeof(float) * SV_TAPE_MAX_LOSS * n_report);
  t->n_report = n_report;
  return SV_OK;
}

// units of the forward pass in launch order: [first, last] node indices (a UNARY group of one lane is one launch)
static std::vector<std::pair<int, int>> tape_units(const sv_tape* t) {
  std::vector<std::pair<int, int>> u;
  for (size_t i = 0; i < t->nodes.size();) {
    const sv_tape_node& n = t->nodes[i];
    size_t e = i + 1;
    if (n.kind == SV_TAPE_UNARY && n.group)
      while (e < t->nodes.size() && e - i < SV_TAPE_MAX_PARTS && t->nodes[e].kind == SV_TAPE_UNARY && t->nodes[e].group == n.group && t->nodes[e].lane == n.lane) ++e;
    u.push_back({(int)i, (int)e - 1});
    i = e;
  }
  return u;
}

// Resources: activation storage of a root tensor (id r), its gradient storage (NT + r), a layer's variable gradients (2 NT + k).  Conservative access sets per node
// (more edges than strictly needed, never fewer): the forward pass READS a node's input activations and WRITES its outputs; loss nodes ADD into their operands'
// gradients there; the backward pass reads activations only (never a conflict) and READ-MODIFY-WRITES the gradient of every tensor the node touches plus its own
// variables' gradients.  An edge u -> v is kept when the two units are on different lanes; a lane's own launches are ordered by its stream, so per (unit, other lane)
// only the latest producer is waited for, and not again if an earlier unit of the same lane already waited for it or a later one.
static void build_schedules(sv_tape* t) {
  const int NT = (int)t->tens.size(), N = (int)t->nodes.size();
  std::vector<int64_t> wkeys;
  auto wres = [&](int64_t w_off) {
    for (size_t k = 0; k < wkeys.size(); ++k) if (wkeys[k] == w_off) return 2 * NT + (int)k;
    wkeys.push_back(w_off);
    return 2 * NT + (int)wkeys.size() - 1;
  };
  auto acc = [&](const sv_tape_node& n, bool backward, std::vector<int>& rd, std::vector<int>& wr) {
    const int ins[6] = {n.x, n.t2, n.t3, n.t4, n.t5, n.t6};
    auto A = [&](int id) { return t->tens[id].root; };
    auto G = [&](int id) { return NT + t->tens[id].root; };
    auto hasg = [&](int id) { return id >= 0 && t->tens[id].goff >= 0; };
    if (!backward) {
      const bool stn_box = n.kind == SV_TAPE_STN && n.t3 >= 0;
      for (int k = 0; k < 6; ++k) if (ins[k] >= 0 && !(stn_box && k == 2)) rd.push_back(A(ins[k]));
      if (n.y >= 0 && n.kind != SV_TAPE_ZPRES && n.kind != SV_TAPE_LOSS) wr.push_back(A(n.y));
      if (stn_box) wr.push_back(A(n.t3));
      if (n.kind == SV_TAPE_ZPRES || n.kind == SV_TAPE_LOSS)
        for (int k = 0; k < 3; ++k) if (hasg(ins[k])) wr.push_back(G(ins[k]));
    } else {
      for (int k = 0; k < 6; ++k) if (ins[k] >= 0) rd.push_back(A(ins[k]));
      if (n.y >= 0) rd.push_back(A(n.y));
      for (int k = 0; k < 6; ++k) if (hasg(ins[k])) wr.push_back(G(ins[k]));
      if (hasg(n.y)) wr.push_back(G(n.y));
      if ((n.kind == SV_TAPE_DENSE || n.kind == SV_TAPE_CONV) && n.w_off >= 0) wr.push_back(wres(n.w_off));
    }
  };
  const std::vector<std::pair<int, int>> units = tape_units(t);
  for (int pass = 0; pass < 2; ++pass) {
    sv_tape::Sched& S = pass ? t->bs : t->fs;
    S.waits.assign(N, {});
    S.rec.assign(N, 0);
    const int NR = 2 * NT + N + 1;
    std::vector<int> last_w(NR, -1);                 // unit that last wrote the resource
    std::vector<std::vector<int>> readers(NR);       // units that read it since
    const int U = (int)units.size();
    std::vector<int> ulane(U), upos(U);              // (upos: position in this pass's launch order)
    int seen[sv_tape::MAX_LANES][sv_tape::MAX_LANES];
    for (auto& r : seen) for (auto& v : r) v = -1;
    for (int k = 0; k < U; ++k) {
      const int u = pass ? U - 1 - k : k;
      upos[u] = k;
      ulane[u] = t->nodes[units[u].first].lane;
      std::vector<int> rd, wr;
      for (int i = units[u].first; i <= units[u].second; ++i) acc(t->nodes[i], pass == 1, rd, wr);
      std::set<int> deps;
      for (int r : rd) if (last_w[r] >= 0) deps.insert(last_w[r]);
      for (int w : wr) {
        if (last_w[w] >= 0) deps.insert(last_w[w]);
        for (int q : readers[w]) deps.insert(q);
      }
      int latest[sv_tape::MAX_LANES];
      for (auto& v : latest) v = -1;
      for (int d : deps)
        if (d != u && ulane[d] != ulane[u] && (latest[ulane[d]] < 0 || upos[d] > upos[latest[ulane[d]]])) latest[ulane[d]] = d;
      for (int l = 0; l < sv_tape::MAX_LANES; ++l) {
        const int d = latest[l];
        if (d < 0 || upos[d] <= seen[ulane[u]][l]) continue;
        seen[ulane[u]][l] = upos[d];
        // the producer unit's event sits behind its last launched node: forward = the unit's last node, backward = its first
        const int evn = pass ? units[d].first : units[d].second;
        S.waits[pass ? units[u].second : units[u].first].push_back(evn);
        S.rec[evn] = 1;
      }
      for (int r : rd) readers[r].push_back(u);
      for (int w : wr) { last_w[w] = u; readers[w].clear(); }
    }
    if (getenv("SV_TAPE_LANES_DEBUG")) {              // the schedule, one line per unit with a cross-lane wait
      int nw = 0;
      for (int k = 0; k < U; ++k) {
        const int u = pass ? U - 1 - k : k, at = pass ? units[u].second : units[u].first;
        for (int d : S.waits[at]) { fprintf(stderr, "tape %s: unit %d (nodes %d..%d kind %d lane %d) waits for node %d (lane %d)\n", pass ? "bwd" : "fwd", k, units[u].first, units[u].second, t->nodes[units[u].first].kind, ulane[u], d, t->nodes[d].lane); ++nw; }
      }
      fprintf(stderr, "tape %s: %d units, %d cross-lane waits\n", pass ? "bwd" : "fwd", U, nw);
    }
  }
}

extern "C" int sv_tape_finalize(sv_tape* t) {
  if (!t || t->finalized) return SV_E_BADARG;
  auto gr = [&](int id) { return id >= 0 && t->tens[id].goff >= 0; };
  auto wr = [&](int id) { if (gr(id)) t->tens[t->root(id)].writers++; };
  // writers of every gradient (views count on their root)
  for (size_t i = 0; i < t->nodes.size(); ++i) {
    const sv_tape_node& n = t->nodes[i];
    switch (n.kind) {
      case SV_TAPE_DENSE: case SV_TAPE_CONV: case SV_TAPE_UPSAMPLE: wr(n.x); break;
      case SV_TAPE_UNARY: if (!(n.x == n.y && n.xo == n.yo)) wr(n.x); break;
      case SV_TAPE_SAMPLE: wr(n.x); wr(n.t2); break;
      case SV_TAPE_LOGITNOISE: wr(n.x); break;
      case SV_TAPE_STN: wr(n.x); wr(n.t2); break;
      case SV_TAPE_RENDER: wr(n.x); wr(n.t2); wr(n.t3); wr(n.t4); break;
      case SV_TAPE_ZPRES: wr(n.t2); wr(n.t3); break;
      case SV_TAPE_LOSS: if (n.mode != 0) wr(n.x); wr(n.t2); break;
      default: break;
    }
  }
  // conv weight images + scratch
  int64_t arena = 0;
  int blocks = 0;
  auto push = [&](PrepJob j) { j.first_block = blocks; blocks += j.nblocks; t->jobs.push_back(j); };
  for (size_t i = 0; i < t->nodes.size(); ++i) {
    const sv_tape_node& n = t->nodes[i];
    sv_tape::Extra& e = t->ex[i];
    if (n.kind == SV_TAPE_CONV) {
      PrepJob j;
      svg_prep_job_fwd(&e.cd, &j);
      j.src_off = n.w_off;
      arena = (arena + 127) / 128 * 128;
      e.wf_off = arena; j.dst_off = arena;
      arena += (int64_t)j.rows * j.ntaps * j.inner;
      push(j);
      if (gr(n.x)) {
        arena = (arena + 127) / 128 * 128;
        e.wd_off = arena;
        for (int c = 0; c < svg_dgrad_classes(&e.cd); ++c) {
          PrepJob jd;
          svg_prep_job_dgrad(&e.cd, c, &jd);
          jd.src_off = n.w_off;
          jd.dst_off = arena;
          arena += (int64_t)jd.rows * jd.ntaps * jd.inner;
          push(jd);
        }
      }
    }
    auto scratch = [&](int64_t floats) { const int64_t o = t->scratch_floats; t->scratch_floats += (floats + 63) / 64 * 64; return o; };
    if (n.kind == SV_TAPE_CONV && t->dtype == SV_BF16) {                                    // bf16 copies of x and of dY
      e.scratch = scratch((t->tens[n.x].rows * t->tens[n.x].ld + 1) / 2);
      e.scratch2 = scratch((t->tens[n.y].rows * t->tens[n.y].ld + 1) / 2);
    }
    if (n.kind == SV_TAPE_STN) e.scratch = scratch(t->tens[n.t2].rows * 4);
    if (n.kind == SV_TAPE_ZPRES) { e.scratch = scratch(t->tens[n.t2].rows); e.scratch2 = scratch(t->tens[n.t3].rows); }
    if (n.kind == SV_TAPE_RENDER) {
      const int64_t nz = t->tens[n.t3].rows;
      e.scratch = scratch(2 * ((nz + 63) / 64 * 64) + (gr(n.t2) ? 0 : (int64_t)n.B * n.H * n.W * n.C));     // g_zp, g_zd (+ a sink for g_bg)
      e.scratch2 = scratch(sv_spair_render_bwd_workspace_floats(n.B, n.H, n.W));
    }
    if (n.kind == SV_TAPE_UPSAMPLE && gr(n.x) && t->tens[t->root(n.x)].writers > 1) e.scratch = scratch(t->tens[n.x].rows * t->tens[n.x].ld);
  }
  t->arena_elems = arena + 128;
  t->prep_blocks = blocks;
  const int64_t es = t->dtype == SV_BF16 ? 2 : 4;
  auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
  int64_t o = 0;
  t->off_jobs = o; o += al((int64_t)t->jobs.size() * sizeof(PrepJob) + 256);
  t->off_arena = o; o += al(t->arena_elems * es);
  t->off_act = o; o += al(t->act_floats * 4);
  t->off_grad = o; o += al(t->grad_floats * 4);
  t->off_scratch = o; o += al(t->scratch_floats * 4 + 256);
  t->off_loss = o; o += al(((int64_t)SV_TAPE_MAX_LOSS * t->B + 4 * SV_TAPE_MAX_LOSS + 8) * 4);
  bool any_conv = false;
  for (const sv_tape::Extra& e : t->ex) any_conv = any_conv || e.has_conv;
  // split-K Dense outputs: one zero fill for all of them (storage is 256-B granular: whole float4s)
  static const bool zero_once = !(getenv("SV_TAPE_ZERO_PER_LAYER") && atoi(getenv("SV_TAPE_ZERO_PER_LAYER")) != 0);
  for (size_t i = 0; i < t->nodes.size(); ++i)
    if (t->nodes[i].kind == SV_TAPE_DENSE && t->ex[i].split_fwd) {
      const TT& y = t->tens[t->nodes[i].y];
      t->zero_y.push_back({t->nodes[i].y, (y.rows * y.ld + 3) / 4 * 4});
    }
  t->zero_at_start = zero_once && !t->zero_y.empty() && t->zero_y.size() <= SV_TAPE_MAX_ZERO;
  // ---- lanes: cross-stream dependencies of the two passes
  // SV_TAPE_LANES = the number of extra streams the tape may use: 0 everything on the caller's stream (A/B, the equivalence test), k: lanes above k fold onto lane k
  // Default 1.  Measured (profiles/r06_spair_lanes.txt; lg_spair Multi-Bird-Hard flags, 32 images, fp32): one stream 2.98 ms per step at every GPU_MAX_HW_QUEUES setting;
  // ONE extra stream for both image branches 2.63 ms at every setting (-12 %); two extra streams 2.65 ms with two hardware queues and 2.85-2.96 with three or four (a third
  // active queue costs what the overlap saves: the same finding as the SPLIT-VAE step's fourth queue).
  static const int lanes_cap = getenv("SV_TAPE_LANES") ? atoi(getenv("SV_TAPE_LANES")) : 1;
  int maxlane = 0;
  for (sv_tape_node& n : t->nodes) {
    if (n.lane > lanes_cap) n.lane = lanes_cap < 0 ? 0 : lanes_cap;
    maxlane = n.lane > maxlane ? n.lane : maxlane;
  }
  t->nlanes = maxlane + 1;
  if (any_conv) { t->off_wgrad = o; o += al(SV_WGRAD_WS_BYTES) * t->nlanes; }
  if (t->nlanes > 1) { build_schedules(t); t->ev_node.assign(t->nodes.size(), nullptr); t->ev_wg.assign(t->nodes.size(), nullptr); }
  t->ws_bytes = o;
  t->finalized = true;
  return SV_OK;
}

// The cross-lane schedule sv_tape_finalize derived, one node at a time (host logic: tests/test_abi.py checks it without a GPU).  pass 0: forward, 1: backward.
// Returns the number of nodes whose event `node`'s launch waits for (written to waits[0 .. max_waits)); *records = 1 when an event is recorded behind the node's
// launch for a later node of another lane.  A UNARY group is one launch: its waits sit on its first node in the forward pass and on its last in the backward
// pass, its event behind its last / first node.  SV_E_BADARG: not finalized, bad pass / node.  A single-lane tape has no schedule: 0 waits, *records = 0.
extern "C" int sv_tape_schedule(const sv_tape* t, int32_t pass, int32_t node, int32_t* waits, int32_t max_waits, int32_t* records) {
  if (!t || !t->finalized || pass < 0 || pass > 1 || node < 0 || node >= (int)t->nodes.size()) return SV_E_BADARG;
  if (records) *records = 0;
  if (t->nlanes <= 1) return 0;
  const sv_tape::Sched& S = pass ? t->bs : t->fs;
  if (records) *records = S.rec[node];
  const int n = (int)S.waits[node].size();
  for (int i = 0; i < n && i < max_waits && waits; ++i) waits[i] = S.waits[node][i];
  return n;
}

extern "C" int64_t sv_tape_workspace_bytes(const sv_tape* t) { return (t && t->finalized) ? t->ws_bytes : -1; }

extern "C" int sv_tape_bind(sv_tape* t, void* workspace, int64_t bytes, void* stream) {
  if (!t || !t->finalized || !workspace || ((uintptr_t)workspace & 255)) return SV_E_BADARG;
  if (bytes < t->ws_bytes) return SV_E_WORKSPACE;
  t->ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(t->ws, 0, (size_t)t->ws_bytes, st) != hipSuccess) return (int)hipGetLastError();     // pad columns stay zero for ever
  if (!t->jobs.empty() &&
      hipMemcpyAsync(t->ws + t->off_jobs, t->jobs.data(), t->jobs.size() * sizeof(PrepJob), hipMemcpyHostToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  if (hipStreamSynchronize(st) != hipSuccess) return (int)hipGetLastError();
  return SV_OK;
}

// byte offsets of a tensor (and its gradient, -1 when it has none) in the workspace; the loss block: [total, reported x 16, means x 16]
extern "C" int sv_tape_tensor_info(const sv_tape* t, int32_t id, int64_t* offset, int64_t* grad_offset) {
  if (!t || !t->finalized || id < 0 || id >= (int)t->tens.size()) return SV_E_BADARG;
  if (offset) *offset = t->off_act + t->tens[id].off * 4;
  if (grad_offset) *grad_offset = t->tens[id].goff < 0 ? -1 : t->off_grad + t->tens[id].goff * 4;
  return SV_OK;
}
extern "C" int sv_tape_loss_info(const sv_tape* t, int64_t* out_offset, int64_t* metric_offset, int32_t* n_loss) {
  if (!t || !t->finalized) return SV_E_BADARG;
  if (out_offset) *out_offset = (char*)t->loss_out() - t->ws;
  if (metric_offset) *metric_offset = (char*)t->metric() - t->ws;
  if (n_loss) *n_loss = t->n_loss;
  return SV_OK;
}

namespace {

int node_forward(sv_tape* t, size_t i, const sv_tape_run_args* a, bool with_grad, hipStream_t st) {
  const sv_tape_node& n = t->nodes[i];
  const sv_tape::Extra& e = t->ex[i];
  auto T = [&](int id) -> const TT& { return t->tens[id]; };
  switch (n.kind) {
    case SV_TAPE_DENSE: {
      const TT &x = T(n.x), &y = T(n.y);
      const float* W = a->params + n.w_off;
      const float* b = n.b_off >= 0 ? a->params + n.b_off : nullptr;
      if (e.split_fwd) {
        if (!t->zero_at_start && hipMemsetAsync(t->act(n.y), 0, (size_t)y.rows * y.ld * 4, st) != hipSuccess) return (int)hipGetLastError();
        const int rc = svk_dense_f32_fwd(t->act(n.x), x.ld, W, b, t->act(n.y), y.ld, (int)x.rows, x.cols, y.cols, SV_ACT_NONE, 1, st);
        if (rc < 0 || rc > 1) return rc;
        if (n.act == SV_ACT_RELU)
          hipLaunchKernelGGL(unary_fwd_kernel, dim3(nblk(y.rows * y.cols)), dim3(256), 0, st, (int)SV_TAPE_RELU, t->act(n.y), y.ld, 0, t->act(n.y), y.ld, 0,
                             y.rows * y.cols, y.cols, 1, 0.f, 0.f);
        SV_LAUNCH_CHECK();
        return SV_OK;
      }
      const int rc = svk_dense_f32_fwd(t->act(n.x), x.ld, W, b, t->act(n.y), y.ld, (int)x.rows, x.cols, y.cols, n.act, 0, st);
      return rc < 0 || rc > 1 ? rc : SV_OK;
    }
    case SV_TAPE_CONV: {
      const size_t es = t->dtype == SV_BF16 ? 2 : 4;
      const void* xin = t->act(n.x);
      if (t->dtype == SV_BF16) {
        const TT& x = T(n.x);
        const int64_t n4 = x.rows * x.ld / 4;
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(nblk(n4)), dim3(256), 0, st, t->act(n.x), (bf16_t*)t->scr(e.scratch), n4);
        SV_LAUNCH_CHECK();
        xin = t->scr(e.scratch);
      }
      return sv_conv2d_nhwc_fwd(&e.cd, xin, t->ws + t->off_arena + e.wf_off * es, a->params + n.b_off, t->act(n.y), st);
    }
    case SV_TAPE_UNARY: {
      const TT &x = T(n.x), &y = T(n.y);
      hipLaunchKernelGGL(unary_fwd_kernel, dim3(nblk(y.rows * n.n)), dim3(256), 0, st, n.op, t->act(n.x), x.ld, n.xo, t->act(n.y), y.ld, n.yo,
                         y.rows * n.n, n.n, n.rep, n.p0, n.p1);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_SAMPLE: {
      const TT &m = T(n.x), &s = T(n.t2), &ep = T(n.t3), &z = T(n.y);
      hipLaunchKernelGGL(sample_fwd_kernel, dim3(nblk(z.rows * n.n)), dim3(256), 0, st, t->act(n.x), m.ld, n.xo, t->act(n.t2), s.ld, n.o2, t->act(n.t3),
                         ep.ld, n.o3, t->act(n.y), z.ld, n.yo, z.rows * n.n, n.n);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_LOGITNOISE: {
      const TT &l = T(n.x), &u = T(n.t2), &y = T(n.y);
      hipLaunchKernelGGL(logitnoise_fwd_kernel, dim3(nblk(y.rows * n.n)), dim3(256), 0, st, t->act(n.x), l.ld, n.xo, t->act(n.t2), u.ld, n.o2,
                         t->act(n.y), y.ld, n.yo, y.rows * n.n, n.n, 1.0f / n.p0);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_UPSAMPLE:
      return sv_upsample2x_fwd(t->act(n.x), t->act(n.y), SV_F32, n.B, n.H, n.W, T(n.x).ld, st);
    case SV_TAPE_STN:
      return sv_stn_sample_fwd(t->act(n.x), t->act(n.t2), t->act(n.y), n.t3 >= 0 ? t->act(n.t3) : nullptr, n.B, n.Hc, n.Wc, n.H, n.W, n.C, n.Ho, n.Wo,
                               n.inverse, st);
    case SV_TAPE_RENDER:
      return sv_spair_render_fwd(t->act(n.x), t->act(n.t2), t->act(n.t3), n.training ? t->act(n.t4) : nullptr, n.training ? nullptr : t->act(n.t5),
                                 (n.training && n.t6 >= 0) ? t->act(n.t6) : nullptr, t->act(n.y), n.B, n.R, n.H, n.W, n.C, n.training, st);
    case SV_TAPE_ZPRES: {
      const float w = with_grad ? a->loss_weights[n.loss_idx] / (float)t->B : 0.f;
      SV_TRY(sv_spair_zpres_kl(t->act(n.x), t->act(n.t2), t->act(n.t3), t->loss_sums() + (int64_t)n.loss_idx * t->B, with_grad ? t->scr(e.scratch2) : nullptr,
                               with_grad ? t->scr(e.scratch) : nullptr, t->B, n.R, a->dyn[n.dyn_idx], n.p0, w, st));
      if (with_grad) {
        const TT &lg = T(n.t2), &pr = T(n.t3);
        hipLaunchKernelGGL(acc_kernel, dim3(nblk(lg.rows)), dim3(256), 0, st, t->grad(n.t2), lg.ld, 0, t->scr(e.scratch), lg.rows, 1, 1.0f);
        hipLaunchKernelGGL(acc_kernel, dim3(nblk(pr.rows)), dim3(256), 0, st, t->grad(n.t3), pr.ld, 0, t->scr(e.scratch2), pr.rows, 1, 1.0f);
        SV_LAUNCH_CHECK();
      }
      return SV_OK;
    }
    case SV_TAPE_LOSS: {
      const TT &ta = T(n.x), &tb = T(n.t2);
      const float sc = with_grad ? a->loss_weights[n.loss_idx] / (float)t->B : 0.f;
      float* ga = with_grad ? t->grad(n.x) : nullptr;
      float* gb = with_grad ? t->grad(n.t2) : nullptr;
      float* sums = t->loss_sums() + (int64_t)n.loss_idx * t->B;
      const float m2 = n.dyn_idx >= 0 ? a->dyn[n.dyn_idx] : n.p0;
      if (n.mode == 0) hipLaunchKernelGGL((tape_loss_kernel<0>), dim3(t->B), dim3(256), 0, st, t->act(n.x), ta.ld, n.xo, t->act(n.t2), tb.ld, n.o2, ga, gb, sums, n.R, n.n, 0.f, 1.f, sc);
      else if (n.mode == 1) hipLaunchKernelGGL((tape_loss_kernel<1>), dim3(t->B), dim3(256), 0, st, t->act(n.x), ta.ld, n.xo, t->act(n.t2), tb.ld, n.o2, ga, gb, sums, n.R, n.n, 0.f, 1.f, sc);
      else hipLaunchKernelGGL((tape_loss_kernel<2>), dim3(t->B), dim3(256), 0, st, t->act(n.x), ta.ld, n.xo, t->act(n.t2), tb.ld, n.o2, ga, gb, sums, n.R, n.n, m2, n.p1, sc);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_NOISE: {
      if (a->pinned_noise) return SV_OK;
      const TT& y = T(n.y);
      const int64_t total = y.rows * y.cols;
      hipLaunchKernelGGL(noise_kernel, dim3(nblk((total + 3) / 4)), dim3(256), 0, st, t->act(n.y), y.ld, total, y.cols, n.op, n.p0, a->seed, a->step,
                         (uint32_t)n.stream_id);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
  }
  return SV_E_BADARG;
}

hipStream_t lane_stream(sv_tape* t, int lane, hipStream_t st);

int node_backward(sv_tape* t, size_t i, const sv_tape_run_args* a, hipStream_t st) {
  const sv_tape_node& n = t->nodes[i];
  const sv_tape::Extra& e = t->ex[i];
  auto T = [&](int id) -> const TT& { return t->tens[id]; };
  auto multi = [&](int id) { return t->tens[t->root(id)].writers > 1; };
  switch (n.kind) {
    case SV_TAPE_DENSE: case SV_TAPE_CONV: {
      const TT &x = T(n.x), &y = T(n.y);
      float* gy = t->grad(n.y);
      if (!gy) return SV_OK;
      if (n.act == SV_ACT_RELU && n.kind == SV_TAPE_CONV) {            // this layer's ReLU: dY *= (y > 0), in place (Dense: gated on load)
        hipLaunchKernelGGL(unary_bwd_kernel, dim3(nblk(y.rows * y.cols)), dim3(256), 0, st, (int)SV_TAPE_RELU, gy, y.ld, 0, gy, t->act(n.y), y.ld, 0,
                           y.rows * y.cols, y.cols, 1, 0.f, 0.f, 1);
        SV_LAUNCH_CHECK();
      }
      float* gx = t->grad(n.x);
      // (where a lane-0 layer's weight gradient goes: see the Dense case)
      static const bool wside = !(getenv("SV_TAPE_WGRAD_SIDE") && atoi(getenv("SV_TAPE_WGRAD_SIDE")) == 0);
      auto wgrad_stream = [&](hipStream_t* ws, int* wlane) -> int {     // call when everything the weight gradient reads has been enqueued on st
        *ws = st; *wlane = t->nlanes > 1 ? n.lane : 0;
        if (!(wside && t->nlanes > 1 && n.lane == 0 && gx)) return SV_OK;
        // conv layers: at bf16 only.  Measured (profiles/r06_spair_lanes.txt, 32 images): Dense layers' weight gradients on lane 1 2.65 -> 2.49 ms (fp32) / 2.60 -> 2.38
        // (bf16); the conv layers' too: 2.52 (fp32: the 77-85 us fp32 weight gradients of the object decoder make lane 1 the longer one) / 2.35 (bf16)
        if (n.kind == SV_TAPE_CONV && t->dtype != SV_BF16) return SV_OK;
        hipStream_t s1 = lane_stream(t, 1, st);
        if (s1 == st) return SV_OK;
        if (!t->ev_wg[i] && hipEventCreateWithFlags(&t->ev_wg[i], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
        if (hipEventRecord(t->ev_wg[i], st) != hipSuccess || hipStreamWaitEvent(s1, t->ev_wg[i], 0) != hipSuccess) return (int)hipGetLastError();
        *ws = s1; *wlane = 1;                  // (lane 1's slab workspace: its own conv layers use it in the same stream's order)
        return SV_OK;
      };
      if (n.kind == SV_TAPE_DENSE) {
        const float* gate = n.act == SV_ACT_RELU ? t->act(n.y) : nullptr;
        // The weight gradient feeds only Adam: with lanes on, a lane-0 layer's goes to lane 1's stream behind an event (dY is final here: every consumer of y ran
        // its adjoint already, and nothing writes the gradient buffers again before the next step's zero fill, which follows the join) and the input-gradient chain --
        // the critical path of the adjoint, ~11 us per launch -- continues at once.  The variables' gradients are written by this launch alone.
        hipStream_t ws;
        int wl;
        SV_TRY(wgrad_stream(&ws, &wl));
        SV_TRY(svk_dense_f32_wgrad(t->act(n.x), x.ld, gy, y.ld, a->grads + n.w_off, n.b_off >= 0 ? a->grads + n.b_off : nullptr, (int)x.rows, x.cols,
                                   y.cols, 1, gate, ws));
        if (gx) SV_TRY(svk_dense_f32_dgrad(gy, y.ld, a->params + n.w_off, gx, x.ld, (int)x.rows, x.cols, y.cols, multi(n.x) ? 1 : 2, gate, st));
        return SV_OK;
      }
      const size_t es = t->dtype == SV_BF16 ? 2 : 4;
      if (t->dtype == SV_BF16) {               // bf16 operands (x was cast in the forward pass), fp32 gradients: dx is ADDED to the zeroed buffer
        const int64_t n4 = y.rows * y.ld / 4;
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(nblk(n4)), dim3(256), 0, st, gy, (bf16_t*)t->scr(e.scratch2), n4);
        SV_LAUNCH_CHECK();
        hipStream_t ws;
        int wl;
        SV_TRY(wgrad_stream(&ws, &wl));          // (behind the cast: the weight gradient reads the bf16 copy of dY)
        SV_TRY(sv_conv2d_nhwc_wgrad_ws(&e.cd, t->scr(e.scratch), t->scr(e.scratch2), a->grads + n.w_off, a->grads + n.b_off, t->wgrad_ws(wl), SV_WGRAD_WS_BYTES, ws));
        if (gx) SV_TRY(sv_conv2d_nhwc_dgrad(&e.cd, t->scr(e.scratch2), t->ws + t->off_arena + e.wd_off * es, nullptr, gx, 1, st));
        return SV_OK;
      }
      hipStream_t ws;
      int wl;
      SV_TRY(wgrad_stream(&ws, &wl));
      SV_TRY(sv_conv2d_nhwc_wgrad_ws(&e.cd, t->act(n.x), gy, a->grads + n.w_off, a->grads + n.b_off, t->wgrad_ws(wl), SV_WGRAD_WS_BYTES, ws));
      if (gx) SV_TRY(sv_conv2d_nhwc_dgrad(&e.cd, gy, t->ws + t->off_arena + e.wd_off * es, nullptr, gx, multi(n.x) ? 1 : 0, st));
      return SV_OK;
    }
    case SV_TAPE_UNARY: {
      const TT &x = T(n.x), &y = T(n.y);
      float* gy = t->grad(n.y);
      float* gx = t->grad(n.x);
      if (!gy || !gx) return SV_OK;
      const int inplace = (n.x == n.y && n.xo == n.yo) ? 1 : 0;
      hipLaunchKernelGGL(unary_bwd_kernel, dim3(nblk(x.rows * n.n)), dim3(256), 0, st, n.op, gx, x.ld, n.xo, gy, t->act(n.y), y.ld, n.yo, x.rows * n.n, n.n,
                         n.rep, n.p0, n.p1, inplace);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_SAMPLE: {
      const TT &m = T(n.x), &s = T(n.t2), &ep = T(n.t3), &z = T(n.y);
      if (!t->grad(n.y) || !t->grad(n.x) || !t->grad(n.t2)) return t->grad(n.y) ? SV_E_BADARG : SV_OK;
      hipLaunchKernelGGL(sample_bwd_kernel, dim3(nblk(z.rows * n.n)), dim3(256), 0, st, t->grad(n.x), m.ld, n.xo, t->grad(n.t2), s.ld, n.o2, t->act(n.t3),
                         ep.ld, n.o3, t->grad(n.y), z.ld, n.yo, z.rows * n.n, n.n);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_LOGITNOISE: {
      const TT &l = T(n.x), &y = T(n.y);
      if (!t->grad(n.y) || !t->grad(n.x)) return SV_OK;
      // d pre / d logits = 1 / tau: a scaled copy-add (sources with a row pitch: one launch per column block)
      hipLaunchKernelGGL(unary_bwd_kernel, dim3(nblk(l.rows * n.n)), dim3(256), 0, st, (int)SV_TAPE_SCALE, t->grad(n.x), l.ld, n.xo, t->grad(n.y), t->act(n.y),
                         y.ld, n.yo, l.rows * n.n, n.n, 1, 1.0f / n.p0, 0.f, 0);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    case SV_TAPE_UPSAMPLE: {
      if (!t->grad(n.y) || !t->grad(n.x)) return SV_OK;
      const TT& x = T(n.x);
      if (e.scratch >= 0) {
        SV_TRY(sv_upsample2x_bwd(t->grad(n.y), nullptr, t->scr(e.scratch), SV_F32, n.B, n.H, n.W, x.ld, st));
        hipLaunchKernelGGL(acc_kernel, dim3(nblk(x.rows * x.ld)), dim3(256), 0, st, t->grad(n.x), x.ld, 0, t->scr(e.scratch), x.rows * x.ld, x.ld, 1.0f);
        SV_LAUNCH_CHECK();
        return SV_OK;
      }
      return sv_upsample2x_bwd(t->grad(n.y), nullptr, t->grad(n.x), SV_F32, n.B, n.H, n.W, x.ld, st);
    }
    case SV_TAPE_STN: {
      if (!t->grad(n.y)) return SV_OK;
      const TT& zw = T(n.t2);
      SV_TRY(sv_stn_sample_bwd(t->act(n.x), t->act(n.t2), t->grad(n.y), t->grad(n.x), t->scr(e.scratch), n.B, n.Hc, n.Wc, n.H, n.W, n.C, n.Ho, n.Wo,
                               n.inverse, st));
      if (t->grad(n.t2)) {
        hipLaunchKernelGGL(acc_kernel, dim3(nblk(zw.rows * 4)), dim3(256), 0, st, t->grad(n.t2), 4, 0, t->scr(e.scratch), zw.rows * 4, 4, 1.0f);
        SV_LAUNCH_CHECK();
      }
      return SV_OK;
    }
    case SV_TAPE_RENDER: {
      if (!t->grad(n.y) || !n.training) return SV_OK;
      const TT &zd = T(n.t3), &zp = T(n.t4);
      const int64_t nz = (zd.rows + 63) / 64 * 64;
      float* g_zp = t->scr(e.scratch);
      float* g_zd = g_zp + nz;
      float* g_bg = t->grad(n.t2) ? t->grad(n.t2) : g_zd + nz;
      SV_TRY(sv_spair_render_bwd_ws(t->act(n.x), t->act(n.t2), t->act(n.t3), t->act(n.t4), n.t6 >= 0 ? t->act(n.t6) : nullptr,
                                    t->grad(n.y), t->grad(n.x), g_bg, g_zp, g_zd, n.B, n.R, n.H, n.W, n.C, t->scr(e.scratch2),
                                    sv_spair_render_bwd_workspace_floats(n.B, n.H, n.W), st));
      if (t->grad(n.t3)) hipLaunchKernelGGL(acc_kernel, dim3(nblk(zd.rows)), dim3(256), 0, st, t->grad(n.t3), zd.ld, 0, g_zd, zd.rows, 1, 1.0f);
      if (t->grad(n.t4)) hipLaunchKernelGGL(acc_kernel, dim3(nblk(zp.rows)), dim3(256), 0, st, t->grad(n.t4), zp.ld, 0, g_zp, zp.rows, 1, 1.0f);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
    default: return SV_OK;        // ZPRES / LOSS added their gradients in the forward pass; NOISE has none
  }
}

// [i, end) = a run of UNARY nodes of one group (at most SV_TAPE_MAX_PARTS)
size_t group_end(const sv_tape* t, size_t i) {
  const sv_tape_node& n = t->nodes[i];
  if (n.kind != SV_TAPE_UNARY || !n.group) return i + 1;
  size_t e = i + 1;
  while (e < t->nodes.size() && e - i < SV_TAPE_MAX_PARTS && t->nodes[e].kind == SV_TAPE_UNARY && t->nodes[e].group == n.group && t->nodes[e].lane == n.lane) ++e;
  return e;
}
size_t group_begin(const sv_tape* t, size_t i) {       // the run that ENDS at i, as the forward pass cut it (scan from the group's first node)
  const sv_tape_node& n = t->nodes[i];
  if (n.kind != SV_TAPE_UNARY || !n.group) return i;
  size_t f = i;
  while (f > 0 && t->nodes[f - 1].kind == SV_TAPE_UNARY && t->nodes[f - 1].group == n.group && t->nodes[f - 1].lane == n.lane) --f;
  size_t b = f;
  while (b + SV_TAPE_MAX_PARTS <= i) b += SV_TAPE_MAX_PARTS;
  return b;
}
int group_forward(sv_tape* t, size_t b, size_t e, hipStream_t st) {
  UMulti m;
  memset(&m, 0, sizeof(m));
  int64_t tot = 0;
  for (size_t i = b; i < e; ++i) {
    const sv_tape_node& n = t->nodes[i];
    const TT &x = t->tens[n.x], &y = t->tens[n.y];
    UPart& u = m.p[m.np++];
    u.x = t->act(n.x); u.y = t->act(n.y); u.ldx = x.ld; u.xo = n.xo; u.ldy = y.ld; u.yo = n.yo; u.n = n.n; u.rep = n.rep; u.op = n.op;
    u.p0 = n.p0; u.p1 = n.p1; u.start = tot;
    tot += y.rows * n.n;
  }
  m.total = tot;
  hipLaunchKernelGGL(unary_multi_fwd_kernel, dim3(nblk(tot)), dim3(256), 0, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
int group_backward(sv_tape* t, size_t b, size_t e, hipStream_t st) {
  UMulti m;
  memset(&m, 0, sizeof(m));
  int64_t tot = 0;
  for (size_t i = b; i < e; ++i) {
    const sv_tape_node& n = t->nodes[i];
    const TT &x = t->tens[n.x], &y = t->tens[n.y];
    if (!t->grad(n.y) || !t->grad(n.x)) continue;
    UPart& u = m.p[m.np++];
    u.gx = t->grad(n.x); u.gy = t->grad(n.y); u.y = t->act(n.y); u.ldx = x.ld; u.xo = n.xo; u.ldy = y.ld; u.yo = n.yo; u.n = n.n; u.rep = n.rep;
    u.op = n.op; u.p0 = n.p0; u.p1 = n.p1; u.inplace = (n.x == n.y && n.xo == n.yo) ? 1 : 0; u.start = tot;
    tot += x.rows * n.n;
  }
  if (!m.np) return SV_OK;
  m.total = tot;
  hipLaunchKernelGGL(unary_multi_bwd_kernel, dim3(nblk(tot)), dim3(256), 0, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ---- lanes at run time: lane 0 is the caller's stream; the others are created on first use
hipStream_t lane_stream(sv_tape* t, int lane, hipStream_t st) {
  if (lane <= 0) return st;
  if (!t->lane_st[lane]) {
    t->lane_st[lane] = sv_shared_stream(lane - 1);       // the process's shared side streams (streams.hip): a stream per tape aliased the caller's hardware queue
    if (!t->lane_st[lane]) return st;
  }
  return t->lane_st[lane];
}
int lanes_fork(sv_tape* t, hipStream_t st) {             // every lane behind what the caller's stream holds now
  if (!t->ev_fork && hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  if (hipEventRecord(t->ev_fork, st) != hipSuccess) return (int)hipGetLastError();
  for (int l = 1; l < t->nlanes; ++l) {
    hipStream_t s = lane_stream(t, l, st);
    if (s != st && hipStreamWaitEvent(s, t->ev_fork, 0) != hipSuccess) return (int)hipGetLastError();
  }
  return SV_OK;
}
int lanes_join(sv_tape* t, hipStream_t st) {             // the caller's stream behind every lane
  for (int l = 1; l < t->nlanes; ++l) {
    hipStream_t s = lane_stream(t, l, st);
    if (s == st) continue;
    if (!t->ev_join[l] && hipEventCreateWithFlags(&t->ev_join[l], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    if (hipEventRecord(t->ev_join[l], s) != hipSuccess || hipStreamWaitEvent(st, t->ev_join[l], 0) != hipSuccess) return (int)hipGetLastError();
  }
  return SV_OK;
}
int lane_waits(sv_tape* t, const std::vector<int>& w, hipStream_t s) {
  for (int nd : w)
    if (t->ev_node[nd] && hipStreamWaitEvent(s, t->ev_node[nd], 0) != hipSuccess) return (int)hipGetLastError();
  return SV_OK;
}
int lane_record(sv_tape* t, int nd, hipStream_t s) {
  if (!t->ev_node[nd] && hipEventCreateWithFlags(&t->ev_node[nd], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  return hipEventRecord(t->ev_node[nd], s) == hipSuccess ? SV_OK : (int)hipGetLastError();
}

}  // namespace

extern "C" int sv_tape_run(sv_tape* t, const sv_tape_run_args* a, void* stream) {
  if (!t || !t->finalized || !t->ws || !a || !a->params) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const bool fwd = a->phases & SV_TAPE_PHASE_FORWARD, bwd = a->phases & SV_TAPE_PHASE_BACKWARD, adam = a->phases & SV_TAPE_PHASE_ADAM;
  if ((bwd || adam) && !a->grads) return SV_E_BADARG;
  if (bwd && !fwd) return SV_E_BADARG;             // the loss nodes seed the gradients during the forward pass
  if (bwd && (!a->loss_weights || a->n_weights < t->n_loss)) return SV_E_BADARG;
  if (adam && (!a->adam_m || !a->adam_v || a->t <= 0 || a->n_params <= 0)) return SV_E_BADARG;
  if (fwd) {
    if (!t->jobs.empty())
      SV_TRY(svk_prep_weights(a->params, t->ws + t->off_arena, t->dtype, (const PrepJob*)(t->ws + t->off_jobs), (int)t->jobs.size(), t->prep_blocks, st));
    if (bwd) {
      if (t->grad_floats && hipMemsetAsync(t->ws + t->off_grad, 0, (size_t)t->grad_floats * 4, st) != hipSuccess) return (int)hipGetLastError();
      if (hipMemsetAsync(a->grads, 0, (size_t)a->n_params * 4, st) != hipSuccess) return (int)hipGetLastError();
    }
    if (t->zero_at_start) {                  // (every lane forks behind this: the previous step's readers of these tensors were joined at its end)
      ZeroList z;
      memset(&z, 0, sizeof(z));
      int64_t tot = 0;
      for (size_t k = 0; k < t->zero_y.size(); ++k) {
        z.p[k] = t->act(t->zero_y[k].first);
        z.start[k] = tot;
        tot += t->zero_y[k].second / 4;
      }
      z.n = (int)t->zero_y.size();
      z.start[z.n] = tot;
      hipLaunchKernelGGL(multi_zero_kernel, dim3(nblk(tot)), dim3(256), 0, st, z);
      SV_LAUNCH_CHECK();
    }
    const bool lanes = t->nlanes > 1;
    if (lanes) SV_TRY(lanes_fork(t, st));
    for (size_t i = 0; i < t->nodes.size(); ++i) {
      const size_t e = group_end(t, i);
      hipStream_t s = lanes ? lane_stream(t, t->nodes[i].lane, st) : st;
      if (lanes) SV_TRY(lane_waits(t, t->fs.waits[i], s));
      if (e > i + 1) { SV_TRY(group_forward(t, i, e, s)); i = e - 1; }
      else SV_TRY(node_forward(t, i, a, bwd, s));
      if (lanes && t->fs.rec[i]) SV_TRY(lane_record(t, (int)i, s));          // (i = the unit's last node by now)
    }
    if (lanes) SV_TRY(lanes_join(t, st));
    if (t->n_loss) {
      FinalArgs f;
      memset(&f, 0, sizeof(f));
      for (int k = 0; k < t->n_loss; ++k) f.w[k] = a->loss_weights ? a->loss_weights[k] : 0.f;
      memcpy(f.rep, t->report, sizeof(f.rep));
      f.nl = t->n_loss; f.nrep = t->n_report; f.B = t->B; f.accumulate = a->accumulate_metrics;
      hipLaunchKernelGGL(tape_final_kernel, dim3(1), dim3(64), 0, st, t->loss_sums(), t->loss_out(), t->metric(), f);
      SV_LAUNCH_CHECK();
    }
  }
  if (bwd) {
    const bool lanes = t->nlanes > 1;
    if (lanes) SV_TRY(lanes_fork(t, st));
    for (size_t i = t->nodes.size(); i-- > 0;) {
      const size_t b = group_begin(t, i);
      hipStream_t s = lanes ? lane_stream(t, t->nodes[i].lane, st) : st;
      if (lanes) SV_TRY(lane_waits(t, t->bs.waits[i], s));
      if (b < i) { SV_TRY(group_backward(t, b, i + 1, s)); i = b; }
      else SV_TRY(node_backward(t, i, a, s));
      if (lanes && t->bs.rec[i]) SV_TRY(lane_record(t, (int)i, s));          // (i = the unit's first node by now)
    }
    if (lanes) SV_TRY(lanes_join(t, st));
  }
  if (adam) {
    if (a->clipnorm > 0.f) {
      if (!a->tensor_off || !a->norm_ws || a->n_tensors < 1) return SV_E_BADARG;
      SV_TRY(sv_adam_step_clipnorm(a->params, a->grads, a->adam_m, a->adam_v, a->tensor_off, a->n_tensors, a->norm_ws, a->clipnorm, a->lr, a->beta1,
                                   a->beta2, a->adam_eps, a->t, 1.0f, st));
    } else {
      SV_TRY(svk_adam_step(a->params, a->grads, a->adam_m, a->adam_v, a->n_params, a->lr, a->beta1, a->beta2, a->adam_eps, a->t, 1.0f, nullptr, st));
    }
  }
  return SV_OK;
}
