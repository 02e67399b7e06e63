// The latent block's GEMMs (bf16, fp32 accumulate): the encoder heads e4_mean | e4_sd (vae/model.py:41-42, :111-112), the decoders'
// d1 Dense (:152, :160) and their input gradients (tape.gradient, vae/trainer.py:137).  Plain row-major products
//     out [M, N] = A [M, K] . W^T,   A and the prepared weight image W [N, K] both K-contiguous
// with M = the batch (64 ... 512 rows) and K, N in {128, 256, 2048, 8192, ...}: 0.8 % of the step's FLOPs but, as eight dependent
// launches of the im2col kernel (two-deep register staging, 64 x 32 tiles, split-K fp32 atomics onto memset buffers), 10 % of its
// time.  Here a workgroup moves a whole 128-deep K phase of its tile -- A [BM x 128] and W [128 x 128] -- into LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, every load of the phase in flight at once), waits once, and runs the phase's
// MFMAs (16x16x32 bf16, operands swapped so a lane ends up with 4 consecutive output channels of one row); two or three such
// workgroups per CU overlap each other's phases.  LDS rows are 256 B (one bank row), so the 16-B piece p of row r lives in slot
// p ^ (r & 15): the DMA writes linearly and each lane FETCHES the piece that belongs in its slot; fragment reads un-swizzle.
// Big-K layers (heads forward, d1 input gradient: K = H/8 * W/8 * 128) split K over blockIdx.z into fp32 slabs [S][M][N] that
// nt_slab_reduce_kernel sums in slice order -- no atomics, no zero fill, run-to-run identical.
//
// fp32 (the reference's precision, vae/model.py:12): the same kernels over float operands.  An LDS row is still 256 B, so a phase is 64
// deep and a 16-B piece holds 4 floats; a lane's piece pair feeds FOUR v_mfma_f32_16x16x4_f32 (instruction e contracts element e of both
// pieces: the same permutation of K on both operands, fix_mma.hip.h), exact fp32 products and sums.
#include "common.hip.h"
#include "fix_mma.hip.h"
#include "kernels.h"

namespace {

constexpr int BN = 128, BK = 128, RB = 256;        // tile columns, K per phase (bf16; fp32: RB / 4 = 64), bytes per LDS row

template <typename T, int BM>
__global__ __launch_bounds__(256) void nt_gemm_kernel(const NtGemmMulti mg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int inst = (mg.n > 1 && (int)blockIdx.z >= mg.p[1].zbase) ? 1 : 0;
  const NtGemmProb& g = mg.p[inst];
  const int zi = (int)blockIdx.z - g.zbase;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= g.M || n0 >= g.N || zi >= g.splitk) return;
  char* sA = smem;
  char* sW = smem + BM * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PE = 16 / (int)sizeof(T), BKT = RB / (int)sizeof(T);      // elements per 16-B piece, K per phase
  const int kper = g.K / g.splitk, kbeg = zi * kper, nph = kper / BKT;
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int FM = BM / 32;                      // 16-row fragments per wave along M (waves 2 x 2: (BM / 2) x 64 each)
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 64;
  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const T* __restrict__ Ab = (const T*)g.A;
  const T* __restrict__ Wb = (const T*)g.W;
  auto issue = [&](int ph) {
    const int kk = kbeg + ph * BKT;
    // one wave-instruction = 64 x 16 B = 4 LDS rows; lane -> (row 4q + lane / 16, slot lane % 16) fetches piece slot ^ (row & 15)
#pragma unroll
    for (int q = wave; q < BM / 4; q += 4) {
      const int r = 4 * q + lg, gm = min(m0 + r, g.M - 1);           // rows past M re-read the last row (discarded in the epilogue)
      const T* src = Ab + (int64_t)gm * g.lda + kk + (lr ^ (r & 15)) * PE;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(sA + q * (4 * RB)),
                                       16, 0, 0);
    }
#pragma unroll
    for (int q = wave; q < BN / 4; q += 4) {
      const int r = 4 * q + lg;
      const T* src = Wb + (int64_t)(n0 + r) * g.ldw + kk + (lr ^ (r & 15)) * PE;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(sW + q * (4 * RB)),
                                       16, 0, 0);
    }
  };
  issue(0);
  for (int ph = 0; ph < nph; ++ph) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMAs have landed ...
    __syncthreads();                                        // ... and everybody else's
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                  // 16 pieces per row, 4 lane groups
      const int p = ks * 4 + lg;
      uint4 af[FM], wf[4];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm + i * 16 + lr;
        af[i] = *(const uint4*)(sA + r * RB + ((p ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn + j * 16 + lr;
        wf[j] = *(const uint4*)(sW + r * RB + ((p ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          FixMma<T>::run(wf[j], af[i], acc[i][j]);
    }
    if (ph + 1 < nph) {
      __syncthreads();                                      // the tile is consumed: the next phase may overwrite it
      issue(ph + 1);
    }
  }
  // D rows = channels, columns = GEMM rows: lane holds channels n0 + wn + 16 j + 4 lg + {0..3} of row m0 + wm + 16 i + lr
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm + i * 16 + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn + j * 16 + lg * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
      if (g.out_f32) {                                      // K slice -> its slab (bias and activation belong to the consumer)
        *(float4*)((float*)g.out + (int64_t)zi * g.slab_stride + (int64_t)m * g.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
        continue;
      }
      if (g.bias) {
        const float4 b = *(const float4*)(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (g.act == SV_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      const int64_t o = (int64_t)m * g.ldo + n;
      if constexpr (sizeof(T) == 4) {
        if (g.mask) {                                       // ReLU gate of the tensor this gradient lands on
          const float4 mk = *(const float4*)((const float*)g.mask + o);
          v[0] = mk.x > 0.f ? v[0] : 0.f; v[1] = mk.y > 0.f ? v[1] : 0.f; v[2] = mk.z > 0.f ? v[2] : 0.f; v[3] = mk.w > 0.f ? v[3] : 0.f;
        }
        *(float4*)((float*)g.out + o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
      bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      if (g.mask) {                                         // ReLU gate of the tensor this gradient lands on
        bf16_t mk[4];
        *(uint2*)mk = *(const uint2*)((const bf16_t*)g.mask + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (float)mk[e] > 0.f ? pk[e] : (bf16_t)0.f;
      }
      *(uint2*)((bf16_t*)g.out + o) = *(uint2*)pk;
      }
    }
  }
}

// The same tile with a RING of NS phase slots (big-K slices: several phases per workgroup, one workgroup per CU, so nothing else hides a
// phase's transfer time): phase ph + NS - 1 is issued right after the barrier that retires phase ph - 1, i.e. NS - 1 phases are in flight
// beside the MFMAs of phase ph.  The transfers are inline assembly with counted waits (the compiler would order every LDS read behind a
// DMA builtin's vmcnt(0): wgrad_roll.hip); every wave issues the same PER instructions per phase, so "all but the youngest PER * n"
// is exactly "phase ph has landed".  K-slice slabs only (bias / activation belong to the consumer).
__device__ __forceinline__ void nt_dma16(const void* base, uint32_t off, const char* lds) {     // base: wave-uniform; off: this lane's byte offset
  const uint32_t l = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}

template <typename T, int BM, int NS>
__global__ __launch_bounds__(256) void nt_gemm_ring_kernel(const NtGemmMulti mg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int inst = (mg.n > 1 && (int)blockIdx.z >= mg.p[1].zbase) ? 1 : 0;
  const NtGemmProb g = mg.p[inst];                  // (a copy: by reference the fields are re-read from the argument segment in the loop)
  const int zi = (int)blockIdx.z - g.zbase;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= g.M || n0 >= g.N || zi >= g.splitk) return;
  constexpr int SLOTB = (BM + BN) * RB;
  constexpr int PER = (BM + BN) / 16;               // DMA instructions per wave and phase
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PE = 16 / (int)sizeof(T), BKT = RB / (int)sizeof(T);
  const int kper = g.K / g.splitk, kbeg = zi * kper, nph = kper / BKT;
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int FM = BM / 32;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 64;
  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this lane's source offsets (bytes, phase 0 of the slice): row part + swizzled piece; a phase adds 2 BK bytes
  uint32_t offA[BM / 16], offW[BN / 16];
#pragma unroll
  for (int i = 0; i < BM / 16; ++i) {
    const int r = 4 * (wave + 4 * i) + lg, gm = min(m0 + r, g.M - 1);
    offA[i] = (uint32_t)(((int64_t)gm * g.lda + kbeg + (lr ^ (r & 15)) * PE) * (int)sizeof(T));
  }
#pragma unroll
  for (int i = 0; i < BN / 16; ++i) {
    const int r = 4 * (wave + 4 * i) + lg;
    offW[i] = (uint32_t)(((int64_t)(n0 + r) * g.ldw + kbeg + (lr ^ (r & 15)) * PE) * (int)sizeof(T));
  }
  auto issue = [&](int ph, int slot) {
    char* sA = smem + slot * SLOTB;
    char* sW = sA + BM * RB;
    const uint32_t kb = (uint32_t)ph * RB;
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) nt_dma16(g.A, offA[i] + kb, sA + (wave + 4 * i) * (4 * RB));
#pragma unroll
    for (int i = 0; i < BN / 16; ++i) nt_dma16(g.W, offW[i] + kb, sW + (wave + 4 * i) * (4 * RB));
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nph) issue(i, i);
  int slot = 0;
  for (int ph = 0; ph < nph; ++ph) {
    // phases issued beyond ph: min(NS - 2, nph - 1 - ph)
    const int ahead = min(NS - 2, nph - 1 - ph);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // phase ph is complete; phase ph - 1 is consumed by every wave
    if (ph + NS - 1 < nph) {
      int ns = slot + NS - 1;
      if (ns >= NS) ns -= NS;
      issue(ph + NS - 1, ns);
    }
    const char* sA = smem + slot * SLOTB;
    const char* sW = sA + BM * RB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                  // 16 pieces per row, 4 lane groups
      const int p = ks * 4 + lg;
      uint4 af[FM], wf[4];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm + i * 16 + lr;
        af[i] = *(const uint4*)(sA + r * RB + ((p ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn + j * 16 + lr;
        wf[j] = *(const uint4*)(sW + r * RB + ((p ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          FixMma<T>::run(wf[j], af[i], acc[i][j]);
    }
    if (++slot == NS) slot = 0;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm + i * 16 + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn + j * 16 + lg * 4;
      *(float4*)((float*)g.out + (int64_t)zi * g.slab_stride + (int64_t)m * g.ldo + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// out[i] = sum_s slab[s][i] in slice order (i over M * ldo floats), both problems of a launch (blockIdx.y)
__global__ __launch_bounds__(256) void nt_slab_reduce_kernel(const NtReduceMulti r) {
  const int z = blockIdx.y;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= r.count[z]) return;
  const float* __restrict__ s = r.slab[z] + i;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int k = 0; k < r.S[z]; ++k) {
    const float4 v = *(const float4*)(s + (int64_t)k * r.stride[z]);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  *(float4*)(r.out[z] + i) = a;
}

template <typename T, int BM>
int launch_nt(const NtGemmMulti& m, hipStream_t st) {
  int gx = 0, gy = 0;
  bool ring = BM == 64;                 // slab launches whose slices span several phases: the ring form
  static const bool no_ring = getenv("SV_NO_NT_RING") != nullptr;
  for (int i = 0; i < m.n; ++i) {
    gx = max(gx, (m.p[i].M + BM - 1) / BM);
    gy = max(gy, m.p[i].N / BN);
    ring = ring && m.p[i].out_f32 && m.p[i].K / m.p[i].splitk >= 2 * (RB / (int)sizeof(T));
  }
  const int gz = m.p[m.n - 1].zbase + m.p[m.n - 1].splitk;
  if constexpr (BM == 64) {
    if (ring && !no_ring) {
      constexpr int NS = 3;
      const size_t lds = (size_t)NS * (BM + BN) * RB;
      sv_ensure_dynamic_lds((const void*)nt_gemm_ring_kernel<T, BM, NS>, lds);
      hipLaunchKernelGGL((nt_gemm_ring_kernel<T, BM, NS>), dim3(gx, gy, gz), dim3(256), lds, st, m);
      SV_LAUNCH_CHECK();
      return SV_OK;
    }
  }
  const size_t lds = (size_t)(BM + BN) * RB;
  sv_ensure_dynamic_lds((const void*)nt_gemm_kernel<T, BM>, lds);
  hipLaunchKernelGGL((nt_gemm_kernel<T, BM>), dim3(gx, gy, gz), dim3(256), lds, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

}  // namespace

bool svk_nt_gemm_supported(const NtGemmProb& p) {
  const int bk = p.f32 ? RB / 4 : BK, pe = p.f32 ? 3 : 7;         // K per phase, elements per 16-B piece - 1
  if (p.M < 1 || p.N < BN || (p.N % BN) || p.K < bk || (p.K % bk) || p.splitk < 1 || (p.K % p.splitk) || ((p.K / p.splitk) % bk)) return false;
  if ((p.lda & pe) || (p.ldw & pe) || (p.ldo & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.out & 15)) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  return true;
}

// K slices for a big-K layer: enough workgroups to fill the chip (>= ~256 per launch of `nprob` problems), whole 128-deep phases
int svk_nt_gemm_pick_splitk(int M, int N, int K, int nprob) {
  if (N < BN || K < BK) return 1;                           // (not a shape this kernel takes: svk_nt_gemm_supported refuses it)
  const int bm = 64;
  const int tiles = ((M + bm - 1) / bm) * (N / BN) * (nprob < 1 ? 1 : nprob);
  int s = (256 + tiles - 1) / tiles;
  int maxs = K / BK;
  static const int cap = getenv("SV_NT_MAXS") ? atoi(getenv("SV_NT_MAXS")) : 0;       // (experiments)
  if (cap > 0 && maxs > cap) maxs = cap;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  while (s > 1 && ((K % s) || ((K / s) % BK))) --s;       // slices of whole phases
  return s;
}

// n <= 2 problems per launch (the x / x-hat twins); zbase is filled here.  bm: 64 | 128 rows per tile.
int svk_nt_gemm_multi(NtGemmProb* p, int n, int bm, hipStream_t st) {
  if (n < 1 || n > 2 || (bm != 64 && bm != 128)) return SV_E_BADARG;
  NtGemmMulti m;
  m.n = n;
  int z = 0;
  for (int i = 0; i < n; ++i) {
    if (!svk_nt_gemm_supported(p[i]) || p[i].f32 != p[0].f32) return SV_E_UNSUPPORTED;
    p[i].zbase = z;
    z += p[i].splitk;
    m.p[i] = p[i];
  }
  if (n == 1) m.p[1] = m.p[0];
  static const int bm32 = getenv("SV_NT_F32_BM") ? atoi(getenv("SV_NT_F32_BM")) : 0;
  if (p[0].f32 && bm32) bm = bm32;
  if (p[0].f32) return bm == 64 ? launch_nt<float, 64>(m, st) : launch_nt<float, 128>(m, st);
  return bm == 64 ? launch_nt<bf16_t, 64>(m, st) : launch_nt<bf16_t, 128>(m, st);
}

int svk_nt_slab_reduce(const NtGemmProb* p, float* const* out, int n, hipStream_t st) {
  NtReduceMulti r;
  int64_t mx = 0;
  for (int i = 0; i < 2; ++i) {
    const int k = i < n ? i : 0;
    r.slab[i] = (const float*)p[k].out; r.out[i] = out[k]; r.S[i] = p[k].splitk; r.stride[i] = p[k].slab_stride;
    r.count[i] = (int64_t)p[k].M * p[k].ldo;
    mx = r.count[i] > mx ? r.count[i] : mx;
  }
  hipLaunchKernelGGL(nt_slab_reduce_kernel, dim3((unsigned)((mx / 4 + 255) / 256), n), dim3(256), 0, st, r);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradients of the latent block's Dense layers (Conv2DBackpropFilter + BiasAddGrad of tape.gradient for e4_mean / e4_sd / d1):
//     dW [Kw, N] = X^T . dY,   dbias [N] = column sums of dY,   X [M, Kw] and dY [M, N] both row-major (the contraction runs over
// their ROWS, M = the batch).  Same LDS-DMA phases as above (128 batch rows of X [.. x 128] and dY [.. x 128] per phase), the MFMA
// operands read k-major with ds_read_b64_tr_b16 (the lane map of wgrad.hip: K index 8g + 4h + q <-> row 16h + 4g + q on both
// operands).  A 16-B piece p of LDS row r sits in slot p ^ 2 (r & 7): the eight rows one lane-half of a transposed read touches land
// on eight different 32-B bank groups.  One workgroup owns a 128 x 128 tile of dW over the WHOLE batch: plain stores, no m-split,
// no atomics; the workgroups of the first Kw tile also produce dbias (an all-ones MFMA tap on the dY fragments).
namespace {

__device__ __forceinline__ short4_t tn_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
}

__global__ __launch_bounds__(256) void tn_wgrad_kernel(const TnWgradMulti mg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TnWgradProb& g = mg.p[blockIdx.z];
  const int w0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  if (w0 >= g.Kw || n0 >= g.N) return;
  char* sA = smem;                    // X tile  [128 rows][256 B]
  char* sB = smem + 128 * 256;        // dY tile [128 rows][256 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  f32x4 acc[4][4], bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = g.dbias != nullptr && blockIdx.x == 0 && wm == 0;
  const short8_t ones = (short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16_t* __restrict__ Xb = (const bf16_t*)g.X;
  const bf16_t* __restrict__ Yb = (const bf16_t*)g.dY;
  const int nph = (g.M + 127) / 128;
  auto issue = [&](int ph) {
    const int m_base = ph * 128;
#pragma unroll
    for (int q = wave; q < 32; q += 4) {            // 4 LDS rows per wave-instruction, 32 instructions per operand tile
      const int r = 4 * q + lg, gm = min(m_base + r, g.M - 1);
      const int piece = lr ^ (2 * (r & 7));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + (int64_t)gm * g.ldx + w0 + piece * 8),
                                       (__attribute__((address_space(3))) void*)(sA + q * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Yb + (int64_t)gm * g.ldy + n0 + piece * 8),
                                       (__attribute__((address_space(3))) void*)(sB + q * 1024), 16, 0, 0);
    }
  };
  issue(0);
  for (int ph = 0; ph < nph; ++ph) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int valid = min(128, g.M - ph * 128);      // M is a multiple of 32: whole 32-row chunks
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk * 32 >= valid) break;
      const int mrow = kk * 32 + 4 * lg + lq;        // (+16 for the high half: same row & 7)
      const int sw = 2 * (mrow & 7);
      short8_t af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = wm + i * 16 + 4 * lp;
        const char* p = sA + mrow * 256 + ((((col >> 3) ^ sw)) << 4) + (col & 7) * 2;
        const short4_t lo = tn_tr16(p), hi = tn_tr16(p + 16 * 256);
        af[i] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wn + j * 16 + 4 * lp;
        const char* p = sB + mrow * 256 + ((((col >> 3) ^ sw)) << 4) + (col & 7) * 2;
        const short4_t lo = tn_tr16(p), hi = tn_tr16(p + 16 * 256);
        bf[j] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, bf[j]), bacc[j], 0, 0, 0);
      }
    }
    if (ph + 1 < nph) {
      __syncthreads();
      issue(ph + 1);
    }
  }
  // D row = (lane >> 4) * 4 + reg -> Kw index, column = lane & 15 -> output channel
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = w0 + wm + i * 16 + lg * 4 + r;
      if (w >= g.Kw_real) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) g.dW[(int64_t)w * g.N + n0 + wn + j * 16 + lr] = acc[i][j][r];
    }
  if (do_bias && lane < 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) g.dbias[n0 + wn + j * 16 + lane] = bacc[j][0];
  }
}

// fp32: the same 128 x 128 tile of dW over the whole batch in exact fp32 (v_mfma_f32_16x16x4_f32 wants ONE float per lane and operand: A [i = lane & 15]
// [k = lane >> 4], B [k][j = lane & 15], so the k-major operands are plain ds_read_b32 of 16 consecutive columns of 4 consecutive batch rows).
// LDS rows are 512 B (128 floats); piece p of row r sits in slot p ^ 4 (r & 3): the four rows of a read land on four different 64-B bank groups.
// Phases are 32 batch rows (32 KB for both operands), two slots: phase ph + 1 is in flight beside the MFMAs of phase ph (inline-assembly
// transfers with counted waits, as nt_gemm_ring_kernel).
__global__ __launch_bounds__(256) void tn_wgrad_f32_kernel(const TnWgradMulti mg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TnWgradProb g = mg.p[blockIdx.z];
  const int w0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  if (w0 >= g.Kw || n0 >= g.N) return;
  constexpr int PR = 32, ROWB = 512, OPB = PR * ROWB, SLOTB = 2 * OPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  f32x4 acc[4][4], bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = g.dbias != nullptr && blockIdx.x == 0 && wm == 0;
  // one wave-instruction = 64 x 16 B = 2 LDS rows; lane -> (row 2 q + lane / 32, slot lane % 32) fetches piece slot ^ 4 (row & 3)
  uint32_t offX[4], offY[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 2 * (wave + 4 * i) + (lane >> 5), piece = (lane & 31) ^ ((r & 3) << 2);
    offX[i] = (uint32_t)(((int64_t)r * g.ldx + w0 + piece * 4) * 4);
    offY[i] = (uint32_t)(((int64_t)r * g.ldy + n0 + piece * 4) * 4);
  }
  const uint32_t phX = (uint32_t)PR * g.ldx * 4, phY = (uint32_t)PR * g.ldy * 4;
  auto issue = [&](int ph, int slot) {
    char* sX = smem + slot * SLOTB;
#pragma unroll
    for (int i = 0; i < 4; ++i) nt_dma16(g.X, offX[i] + (uint32_t)ph * phX, sX + (wave + 4 * i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) nt_dma16(g.dY, offY[i] + (uint32_t)ph * phY, sX + OPB + (wave + 4 * i) * 1024);
  };
  const int nph = g.M / PR;                          // (M is a multiple of 32: svk_tn_wgrad_supported)
  issue(0, 0);
  int slot = 0;
  for (int ph = 0; ph < nph; ++ph) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // phase ph is complete; phase ph - 1 is consumed by every wave
    if (ph + 1 < nph) issue(ph + 1, slot ^ 1);
    const char* sX = smem + slot * SLOTB + lg * ROWB + (lr & 3) * 4;
    const char* sY = sX + OPB;
    const int sw = lg << 2;                          // (row & 3 == lane group: rows 4 kk + lg)
#pragma unroll
    for (int kk = 0; kk < PR / 4; ++kk) {
      float af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const float*)(sX + kk * (4 * ROWB) + (((((wm + i * 16) >> 2) + (lr >> 2)) ^ sw) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *(const float*)(sY + kk * (4 * ROWB) + (((((wn + j * 16) >> 2) + (lr >> 2)) ^ sw) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bf[j], bacc[j], 0, 0, 0);
      }
    }
    slot ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = w0 + wm + i * 16 + lg * 4 + r;
      if (w >= g.Kw_real) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) g.dW[(int64_t)w * g.N + n0 + wn + j * 16 + lr] = acc[i][j][r];
    }
  if (do_bias && lane < 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) g.dbias[n0 + wn + j * 16 + lane] = bacc[j][0];
  }
}

}  // namespace

bool svk_tn_wgrad_supported(const TnWgradProb& p) {
  const int pe = p.f32 ? 3 : 7;
  return p.M >= 32 && !(p.M & 31) && p.Kw >= 128 && !(p.Kw & 127) && p.N >= 128 && !(p.N & 127) && !(p.ldx & pe) && !(p.ldy & pe) &&
         !((uintptr_t)p.X & 15) && !((uintptr_t)p.dY & 15) && p.Kw_real <= p.Kw && p.Kw_real > 0;
}

// n <= 4 problems per launch (both networks' d1, or the four head kernels); dW / dbias are ASSIGNED
int svk_tn_wgrad_multi(const TnWgradProb* p, int n, hipStream_t st) {
  if (n < 1 || n > 4) return SV_E_BADARG;
  TnWgradMulti m;
  int gx = 0, gy = 0;
  for (int i = 0; i < n; ++i) {
    if (!svk_tn_wgrad_supported(p[i]) || p[i].f32 != p[0].f32) return SV_E_UNSUPPORTED;
    m.p[i] = p[i];
    gx = max(gx, p[i].Kw / 128);
    gy = max(gy, p[i].N / 128);
  }
  for (int i = n; i < 4; ++i) m.p[i] = p[0];
  if (p[0].f32) {
    const size_t lds32 = 2 * 2 * 32 * 512;
    sv_ensure_dynamic_lds((const void*)tn_wgrad_f32_kernel, lds32);
    hipLaunchKernelGGL(tn_wgrad_f32_kernel, dim3(gx, gy, n), dim3(256), lds32, st, m);
    SV_LAUNCH_CHECK();
    return SV_OK;
  }
  const size_t lds = 2 * 128 * 256;
  sv_ensure_dynamic_lds((const void*)tn_wgrad_kernel, lds);
  hipLaunchKernelGGL(tn_wgrad_kernel, dim3(gx, gy, n), dim3(256), lds, st, m);
  SV_LAUNCH_CHECK();
  return SV_OK;
}
