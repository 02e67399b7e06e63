// Dense layers of any shape in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 fma chain).
//
// SPLIT-SPAIR's Dense layers (spair/spair.py:135-154, :185-202, :246-273, :341-366, :424-467) have fan-ins like 6912, 500, 177 or 68:
// not the power-of-two channel counts the conv kernels' K pieces want (conv_api.hip svg_check), so they get their own GEMM.
// One kernel, three operand layouts, all reading the Keras [in, out] kernel as it lies in the variable buffer (no weight images):
//   forward          y [M,N]  = x [M,K] . W [K,N] (+ bias)      A row-major (k contiguous), B row-major (n contiguous)
//   input gradient   dx [M,K] (+)= dy [M,N] . W^T               A row-major, B = W read transposed (TB)
//   weight gradient  dW [K,N] += x^T . dy, db += colsum(dy)     A = x read transposed (TA), B row-major
// 64 x 64 output tile per workgroup, 32-deep K steps, the next step's global loads issued before the current step's MFMAs; four
// waves, each 32 x 32 = 2 x 2 fragments of 16 x 16.  K can be split over blockIdx.z (the few-row layers with 28 MB kernels:
// 6912 <-> 1024 at 32 rows): partial sums then leave through fp32 atomics onto a zeroed / accumulating output.
#include "common.hip.h"
#include "kernels.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LP = 68;   // LDS rows of 68 floats: 16-B aligned, consecutive k rows 4 banks apart

// A operand: TA = 0: stored [M][lda] (k contiguous); TA = 1: stored [K][lda] (m contiguous).  Thread t stages 8 floats.
// `gate` (optional, laid out like P): the operand is dy of a ReLU layer and gate its activation output: elements with gate <= 0 read as 0
// (ReluGrad fused into the loads of the layer's own input / weight gradient: no separate masking pass over dy)
template <int T>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, const float* __restrict__ gate, int ld, int mn0, int k0, int MN, int Kend, int tid,
                                          bool vec, float (&r)[8]) {
  int64_t o;
  bool full, any;
  int lim;                                       // valid elements of the 8 along the contiguous direction
  if (T == 0) {            // rows = m (or n), k contiguous: thread -> row tid / 4, k piece (tid % 4) * 8
    const int row = mn0 + (tid >> 2), k = k0 + (tid & 3) * 8;
    o = (int64_t)row * ld + k;
    any = row < MN; lim = Kend - k; full = any && vec && lim >= 8;
  } else {                 // rows = k, m (or n) contiguous: thread -> k row tid / 8, piece (tid % 8) * 8
    const int k = k0 + (tid >> 3), c = mn0 + (tid & 7) * 8;
    o = (int64_t)k * ld + c;
    any = k < Kend; lim = MN - c; full = any && vec && lim >= 8;
  }
  const float* p = P + o;
  if (full) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    if (gate) {
      const float4 ga = *(const float4*)(gate + o), gb = *(const float4*)(gate + o + 4);
      const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = gv[i] > 0.f ? r[i] : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = any && i < lim;
      r[i] = ok ? p[i] : 0.f;
      if (gate && ok && !(gate[o + i] > 0.f)) r[i] = 0.f;
    }
  }
}
// LDS image is always [k][m]: k-major rows of LP floats
template <int T>
__device__ __forceinline__ void store_tile(float* s, int tid, const float (&r)[8]) {
  if (T == 0) {
    const int m = tid >> 2, k = (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[(k + i) * LP + m] = r[i];
  } else {
    const int k = tid >> 3, c = (tid & 7) * 8;
    *(float4*)(s + k * LP + c) = make_float4(r[0], r[1], r[2], r[3]);
    *(float4*)(s + k * LP + c + 4) = make_float4(r[4], r[5], r[6], r[7]);
  }
}

struct DenseArgs {
  const float* A; const float* B; float* C; const float* bias; float* colsum;
  int lda, ldb, ldc, M, N, K, act, splitk, atomic, vecA, vecB;
  const float* gateA; const float* gateB;        // ReLU gates of the A / B operand (laid out like it), or null
};

template <int TA, int TB>
__global__ __launch_bounds__(256) void dense_f32_kernel(const DenseArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[2][BK * LP];
  __shared__ __attribute__((aligned(16))) float sB[2][BK * LP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, zi = blockIdx.z;
  const int nks = (g.K + BK - 1) / BK;
  const int ks0 = (int)(((int64_t)nks * zi) / g.splitk), ks1 = (int)(((int64_t)nks * (zi + 1)) / g.splitk);
  const int Kend = min(g.K, ks1 * BK);
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
  float ra[8], rb[8];
  float csum = 0.f;                                  // weight gradient: column sums of dy for the bias (m-tile 0 only)
  const bool do_cs = g.colsum != nullptr && blockIdx.x == 0 && tid < BN;
  if (ks0 < ks1) {
    load_tile<TA>(g.A, g.gateA, g.lda, m0, ks0 * BK, g.M, Kend, tid, g.vecA, ra);
    load_tile<TB ? 0 : 1>(g.B, g.gateB, g.ldb, n0, ks0 * BK, g.N, Kend, tid, g.vecB, rb);
    store_tile<TA>(sA[0], tid, ra);
    store_tile<TB ? 0 : 1>(sB[0], tid, rb);
  }
  __syncthreads();
  for (int ks = ks0; ks < ks1; ++ks) {
    const int buf = (ks - ks0) & 1;
    const bool more = ks + 1 < ks1;
    if (more) {
      load_tile<TA>(g.A, g.gateA, g.lda, m0, (ks + 1) * BK, g.M, Kend, tid, g.vecA, ra);
      load_tile<TB ? 0 : 1>(g.B, g.gateB, g.ldb, n0, (ks + 1) * BK, g.N, Kend, tid, g.vecB, rb);
    }
    const float* cA = sA[buf];
    const float* cB = sB[buf];
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int k = kk * 4 + lk;
      const float a0 = cA[k * LP + wm + lr], a1 = cA[k * LP + wm + 16 + lr];
      const float b0 = cB[k * LP + wn + lr], b1 = cB[k * LP + wn + 16 + lr];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (do_cs) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) csum += cB[k * LP + tid];
    }
    if (more) {
      store_tile<TA>(sA[buf ^ 1], tid, ra);
      store_tile<TB ? 0 : 1>(sB[buf ^ 1], tid, rb);
    }
    __syncthreads();
  }
  // D: row = (lane >> 4) * 4 + reg, column = lane & 15 of each fragment
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 16 + lr;
      if (n >= g.N) continue;
      const float bv = (g.bias && zi == 0) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + lk * 4 + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bv;
        float* o = g.C + (int64_t)m * g.ldc + n;
        if (g.atomic) atomicAdd(o, v);
        else {
          if (g.act == SV_ACT_RELU) v = fmaxf(v, 0.f);
          *o = v;
        }
      }
    }
  if (do_cs && n0 + tid < g.N) atomicAdd(g.colsum + n0 + tid, csum);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int TA, int TB>
int launch(const DenseArgs& a, hipStream_t st) {
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.splitk);
  hipLaunchKernelGGL((dense_f32_kernel<TA, TB>), grid, dim3(256), 0, st, a);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// K slices so that the launch has at least ~256 workgroups while every slice keeps >= 2 K steps
int pick_splitk(int M, int N, int K) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nks = (K + BK - 1) / BK;
  if (tiles >= 128 || nks < 8) return 1;
  static const int target = getenv("SV_DENSE_SPLIT_WGS") ? atoi(getenv("SV_DENSE_SPLIT_WGS")) : 256;      // A/B knob
  static const int min_steps = getenv("SV_DENSE_SPLIT_MIN_STEPS") ? atoi(getenv("SV_DENSE_SPLIT_MIN_STEPS")) : 2;
  int s = (target + tiles - 1) / tiles;
  if (s > nks / min_steps) s = nks / min_steps;
  return s < 1 ? 1 : s;
}

}  // namespace

// y [M, ldy] = act(x [M, ldx] . W [K, N] + bias).  When K is split the partial sums are ADDED to y with atomics: the caller
// zeroes y first and applies the activation afterwards (returns 1 instead of SV_OK to say so; `allow_split` = 0 forbids it).
int svk_dense_f32_fwd(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int M, int K, int N, int act,
                      int allow_split, hipStream_t st) {
  DenseArgs a = {x, W, y, bias, nullptr, ldx, N, ldy, M, N, K, act, 1, 0, 0, 0, nullptr, nullptr};
  a.splitk = allow_split ? pick_splitk(M, N, K) : 1;
  a.atomic = a.splitk > 1;
  a.vecA = !(ldx & 3) && al16(x); a.vecB = !(N & 3) && al16(W);
  const int rc = launch<0, 0>(a, st);
  return rc ? rc : (a.atomic ? 1 : SV_OK);
}
int svk_dense_f32_fwd_splits(int M, int K, int N) { return pick_splitk(M, N, K); }

// dx [M, ldx] (+)= dy [M, ldy] . W^T.  accumulate: 0 plain stores (K never split); 1 atomic adds onto dx; 2 dx is known to be ZERO (a
// gradient buffer zeroed for this step with no earlier writer): stores when one workgroup owns an output tile, atomics when K is split.
// y_gate (optional, [M, ldy]): the layer's ReLU output -- dy is gated on load (ReluGrad fused).
int svk_dense_f32_dgrad(const float* dy, int ldy, const float* W, float* dx, int ldx, int M, int K, int N, int accumulate, const float* y_gate,
                        hipStream_t st) {
  DenseArgs a = {dy, W, dx, nullptr, nullptr, ldy, N, ldx, M, K, N, 0, 1, 0, 0, 0, y_gate, nullptr};      // contraction over the layer's N outputs
  a.splitk = accumulate ? pick_splitk(M, K, N) : 1;
  a.atomic = accumulate == 1 || a.splitk > 1;
  a.vecA = !(ldy & 3) && al16(dy) && (!y_gate || al16(y_gate)); a.vecB = !(N & 3) && al16(W);
  return launch<0, 1>(a, st);
}

// dW [K, N] += x^T . dy;  db [N] += column sums of dy (may be null), onto gradients the caller zeroed.  zeroed_once != 0: dW holds zeros and
// nothing else adds to it this step, so an unsplit launch stores instead of adding (28 MB kernels: HBM stores, not fp32 atomics).
int svk_dense_f32_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dW, float* db, int M, int K, int N, int zeroed_once,
                        const float* y_gate, hipStream_t st) {
  DenseArgs a = {x, dy, dW, nullptr, db, ldx, ldy, N, K, N, M, 0, 1, 1, 0, 0, nullptr, y_gate};           // output [K, N], contraction over the M rows
  a.splitk = pick_splitk(K, N, M);
  a.atomic = !(zeroed_once && a.splitk == 1);
  a.vecA = !(ldx & 3) && al16(x); a.vecB = !(ldy & 3) && al16(dy) && (!y_gate || al16(y_gate));
  return launch<1, 0>(a, st);
}

extern "C" int sv_dense_f32_fwd(const float* x, int32_t ldx, const float* w, const float* bias, float* y, int32_t ldy, int32_t M, int32_t K,
                                int32_t N, int32_t act, void* stream) {
  if (!x || !w || !y || M < 1 || K < 1 || N < 1 || ldx < K || ldy < N || (act != SV_ACT_NONE && act != SV_ACT_RELU)) return SV_E_BADARG;
  return svk_dense_f32_fwd(x, ldx, w, bias, y, ldy, M, K, N, act, 0, (hipStream_t)stream);
}
extern "C" int sv_dense_f32_dgrad(const float* dy, int32_t ldy, const float* w, float* dx, int32_t ldx, int32_t M, int32_t K, int32_t N,
                                  int32_t accumulate, void* stream) {
  if (!dy || !w || !dx || M < 1 || K < 1 || N < 1 || ldx < K || ldy < N) return SV_E_BADARG;
  return svk_dense_f32_dgrad(dy, ldy, w, dx, ldx, M, K, N, accumulate ? 1 : 0, nullptr, (hipStream_t)stream);
}
extern "C" int sv_dense_f32_wgrad(const float* x, int32_t ldx, const float* dy, int32_t ldy, float* dw, float* dbias, int32_t M, int32_t K,
                                  int32_t N, void* stream) {
  if (!x || !dy || !dw || M < 1 || K < 1 || N < 1 || ldx < K || ldy < N) return SV_E_BADARG;
  return svk_dense_f32_wgrad(x, ldx, dy, ldy, dw, dbias, M, K, N, 0, nullptr, (hipStream_t)stream);
}
