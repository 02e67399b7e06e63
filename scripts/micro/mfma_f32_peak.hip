// Practical ceiling of v_mfma_f32_16x16x4_f32 on gfx950: independent accumulator chains, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_f32_peak scripts/micro/mfma_f32_peak.hip && gpurun_out/mfma_f32_peak
// Prints TFLOP/s for 1, 2 and 4 waves per SIMD with NACC independent accumulators per wave (the fp32 kernels of this repo hold 8 .. 16),
// and with one ds_read_b32 pair per MFMA pair interleaved (the weight-gradient kernel's operand pattern).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ float sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = a0 + i;
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (LDS) { a = sh[(threadIdx.x + it * 64) & 4095]; b = sh[(threadIdx.x * 2 + it * 32) & 4095]; }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC, bool LDS>
void run(int wgs_per_cu, const char* tag) {
  float* out; hipMalloc(&out, 4);
  const int iters = 20000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, LDS><<<grid, 256>>>(out, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, LDS><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * NACC * 2048.0;
  printf("%-28s NACC %2d  %d waves/SIMD  %.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)\n", tag, NACC, wgs_per_cu, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
  hipFree(out);
}

int main() {
  run<16, false>(1, "mfma only"); run<16, false>(2, "mfma only"); run<16, false>(4, "mfma only");
  run<8, false>(1, "mfma only"); run<8, false>(2, "mfma only");
  run<4, false>(1, "mfma only"); run<4, false>(2, "mfma only");
  run<16, true>(1, "mfma + 2 ds_read / 16"); run<16, true>(2, "mfma + 2 ds_read / 16");
  run<2, true>(2, "mfma + 2 ds_read / 2"); run<2, true>(4, "mfma + 2 ds_read / 2");
  return 0;
}
