// Do VALU instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD (gfx950)?  One 8-wave workgroup per CU: waves 0-3 (one
// per SIMD) run a chain-free MFMA loop, waves 4-7 (the second wave of each SIMD) a chain-free v_fma_f32 / v_pk_fma_f32 loop.
// mode 1: MFMA waves only, 2: VALU waves only, 3: both.  Overlap <=> t(3) ~ max(t(1), t(2)); no overlap <=> t(3) ~ t(1) + t(2).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_mfma scripts/micro/valu_mfma_overlap.hip ; run: /tmp/valu_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(512) void k(float* out, int mode, int nm, int nv) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < nm; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    if (!(mode & 2)) return;
    if (PK) {
      f32x2 x[8], m = {1.0001f, 0.9999f}, ad = {1e-3f, -1e-3f};
      for (int j = 0; j < 8; ++j) x[j] = (f32x2){(float)threadIdx.x + j, (float)j};
      for (int i = 0; i < nv; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], m, ad);
      float s = 0;
      for (int j = 0; j < 8; ++j) s += x[j][0] + x[j][1];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
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
}

template <int PK>
static float run(float* d, int mode, int nm, int nv) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<PK>, dim3(256), dim3(512), 0, 0, d, mode, nm, nv);
  hipEventRecord(a, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<PK>, dim3(256), dim3(512), 0, 0, d, mode, nm, nv);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int nm = 20000;                       // 80 000 MFMAs per wave: 16 cycles each = 1.28 M cycles
  for (int nv : {10000, 20000, 40000}) {      // 8 nv VALU instructions per wave: 4 cycles each
    printf("v_fma_f32    nv=%6d: MFMA only %8.1f us, VALU only %8.1f us, both %8.1f us\n", nv, run<0>(d, 1, nm, nv), run<0>(d, 2, nm, nv), run<0>(d, 3, nm, nv));
    printf("v_pk_fma_f32 nv=%6d: MFMA only %8.1f us, VALU only %8.1f us, both %8.1f us\n", nv, run<1>(d, 1, nm, nv), run<1>(d, 2, nm, nv), run<1>(d, 3, nm, nv));
  }
  return 0;
}
