// How much VALU work hides beside bf16 MFMAs on a gfx950 SIMD, per MFMA SHAPE?  (round 4: the 32x32x16 case the round-3 table lacked)
//
// Part A (round 3's question, both shapes now): one 8-wave workgroup per CU; waves 0-3 (one per SIMD) run a chain-free MFMA loop, waves
//   4-7 (the SIMD's second wave) a chain-free v_fma_f32 loop.  mode 1: MFMA waves only, 2: VALU waves only, 3: both.
//   overlap <=> t(3) ~ max(t(1), t(2)); no overlap <=> t(3) ~ t(1) + t(2).
// Part B (what the conv kernels actually do): the fillers sit in the SAME wave, F v_fma_f32 per 32 768 MFMA FLOP (= per one 32x32x16 or per
//   two 16x16x32), each instruction its own asm volatile statement so the order is the source's; one or two waves per SIMD, all waves alike.  Reported per shape and F:
//   wall time for the same FLOPs, cycles per 32 KFLOP by s_memtime, and the clock the chip held (cycles / wall).
// Operands are pseudo-random bf16 in [-1, 1) (MI355X_MICROARCH 'DVFS give-back' item 7: zero or trivial operands rank the shapes by cycles only).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_mfma scripts/micro/valu_mfma_overlap.hip ; run: /tmp/valu_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd(unsigned s) {
  s = s * 747796405u + 2891336453u;
  s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
  s = (s >> 22) ^ s;
  return (float)(s & 0xffff) * (2.0f / 65536.0f) - 1.0f;
}

__device__ __forceinline__ void operands(bf16x8& a, bf16x8& b) {
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)rnd(threadIdx.x * 16 + i + blockIdx.x * 9973);
    b[i] = (__bf16)rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 7919);
  }
}

// ---- part A: MFMA wave beside a VALU wave -------------------------------------------------------------------------------------------------
template <int S32>
__global__ __launch_bounds__(512) void ka(float* out, int mode, int nm, int nv) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    bf16x8 a, b;
    operands(a, b);
    if (S32) {
      f32x16 c0 = {}, c1 = {};
      for (int i = 0; i < nm; ++i) {           // 2 x 32x32x16 = the FLOPs of 4 x 16x16x32
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      }
      out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1];
    } else {
      f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      for (int i = 0; i < nm; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
      out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else {
    if (!(mode & 2)) return;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = (float)threadIdx.x + j;
    for (int i = 0; i < nv; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], 1.0001f, 1e-3f);
    float s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

// ---- part B: fillers inside the MFMA wave -------------------------------------------------------------------------------------------------
// one "unit" = 32 768 FLOP: S32 ? 1 MFMA + F fillers : 2 x (1 MFMA + F/2 fillers).  A loop iteration = 4 units (independent accumulators).
template <int S32, int F, int NT>
__global__ __launch_bounds__(NT) void kb(float* out, unsigned long long* cyc, int n) {
  bf16x8 a, b;
  operands(a, b);
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = rnd(threadIdx.x + j * 77);
  f32x16 C[4] = {};
  f32x4 c[8] = {};
  const float m = 1.0001f, ad = 1e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // every instruction of the loop body is its own asm volatile statement: hipcc keeps their order (it SLP-packs plain fmaf into v_pk_fma_f32
  // and sinks them behind the MFMAs otherwise, sched_group_barrier or not)
  for (int i = 0; i < n; ++i) {
    if (S32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(C[u]) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(u * F + f) & 7]) : "v"(m), "v"(ad));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[u]) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < F / 2; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(u * (F / 2) + f) & 7]) : "v"(m), "v"(ad));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int j = 0; j < 8; ++j) s += x[j];
  for (int u = 0; u < 4; ++u) s += C[u][u];
  for (int u = 0; u < 8; ++u) s += c[u][u & 3];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float time_us(void (*launch)(void*), void* ctx) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(ctx);
  hipEventRecord(a, 0);
  for (int r = 0; r < 5; ++r) launch(ctx);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / 5 * 1e3f;
}

struct ACtx { float* d; int s32, mode, nm, nv; };
static void launch_a(void* p) {
  ACtx* c = (ACtx*)p;
  if (c->s32) hipLaunchKernelGGL(ka<1>, dim3(256), dim3(512), 0, 0, c->d, c->mode, c->nm, c->nv);
  else hipLaunchKernelGGL(ka<0>, dim3(256), dim3(512), 0, 0, c->d, c->mode, c->nm, c->nv);
}

struct BCtx { float* d; unsigned long long* cyc; int n; };
template <int S32, int F, int NT>
static void launch_b(void* p) {
  BCtx* c = (BCtx*)p;
  hipLaunchKernelGGL((kb<S32, F, NT>), dim3(256), dim3(NT), 0, 0, c->d, c->cyc, c->n);
}

static unsigned long long med_cyc(unsigned long long* dcyc) {
  unsigned long long h[256];
  hipMemcpy(h, dcyc, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; ++i)
    for (int j = i + 1; j < 256; ++j)
      if (h[j] < h[i]) { unsigned long long t = h[i]; h[i] = h[j]; h[j] = t; }
  return h[128];
}

template <int S32, int F, int NT>
static void row_b(BCtx& c) {
  // warm the clock governor with ~50 ms of the same kernel before timing
  for (int r = 0; r < 40; ++r) launch_b<S32, F, NT>(&c);
  const float us = time_us(launch_b<S32, F, NT>, &c);
  const double units = 4.0 * c.n;                       // per wave
  const double cyc = (double)med_cyc(c.cyc);
  const int wps = NT / 256;
  const double tflops = units * 32768.0 * 256 * (NT / 64) / (us * 1e-6) / 1e12;
  printf("%-9s waves/SIMD %d  F=%2d : %8.1f us  %7.1f TFLOP/s  %6.1f SIMD cycles per 32 KFLOP (floor 32)  clock %.2f GHz\n",
         S32 ? "32x32x16" : "16x16x32", wps, F, us, tflops, cyc / units / wps, cyc / (us * 1e-6) / 1e9);
}

template <int NT>
static void table_b(BCtx& c) {
  row_b<0, 0, NT>(c);  row_b<1, 0, NT>(c);
  row_b<0, 2, NT>(c);  row_b<1, 2, NT>(c);
  row_b<0, 4, NT>(c);  row_b<1, 4, NT>(c);
  row_b<0, 6, NT>(c);  row_b<1, 6, NT>(c);
  row_b<0, 8, NT>(c);  row_b<1, 8, NT>(c);
  row_b<0, 12, NT>(c); row_b<1, 12, NT>(c);
  row_b<0, 16, NT>(c); row_b<1, 16, NT>(c);
}

int main() {
  float* d;
  unsigned long long* dcyc;
  hipMalloc(&d, 256 * 512 * 4);
  hipMalloc(&dcyc, 256 * 8);
  printf("== part A: one MFMA wave + one VALU wave per SIMD (us) ==\n");
  const int nm = 20000;                       // 80 000 16x16x32 (or 40 000 32x32x16) per wave = 1.28 M MFMA cycles
  for (int s32 = 0; s32 < 2; ++s32)
    for (int nv : {10000, 20000, 40000}) {    // 8 nv VALU instructions per wave
      ACtx c1{d, s32, 1, nm, nv}, c2{d, s32, 2, nm, nv}, c3{d, s32, 3, nm, nv};
      for (int r = 0; r < 10; ++r) launch_a(&c3);
      const float t1 = time_us(launch_a, &c1), t2 = time_us(launch_a, &c2), t3 = time_us(launch_a, &c3);
      printf("%-9s nv=%6d: MFMA only %8.1f, VALU only %8.1f, both %8.1f  (sum %8.1f, max %8.1f: both = %.0f %% of the sum)\n",
             s32 ? "32x32x16" : "16x16x32", nv, t1, t2, t3, t1 + t2, t1 > t2 ? t1 : t2, 100.0 * t3 / (t1 + t2));
    }
  printf("== part B: F v_fma_f32 per 32 KFLOP inside the MFMA waves ==\n");
  BCtx cb{d, dcyc, 10000};                    // 40 000 units per wave
  table_b<256>(cb);
  table_b<512>(cb);
  return 0;
}
