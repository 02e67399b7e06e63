// One 16-B piece of each operand per lane on the matrix pipe, for the small border kernels (poly_fix.hip, polyd_dgrad.hip):
//   bf16: 8 channels x 4 lane groups = 32 channels in ONE v_mfma_f32_16x16x32_bf16
//   fp32: 4 channels x 4 lane groups = 16 channels in FOUR v_mfma_f32_16x16x4_f32 (instruction e contracts channel 4 * group + e of both operands): exact fp32
#pragma once
#include "common.hip.h"

template <typename T> struct FixMma;
template <> struct FixMma<bf16_t> {
  static constexpr int CPG = 32;
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct FixMma<float> {
  static constexpr int CPG = 16;
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    const float4 af = __builtin_bit_cast(float4, a), bf = __builtin_bit_cast(float4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, c, 0, 0, 0);
  }
};

// hipcc 7.2 pads the wait states between an MFMA and a read of its result inside a basic block (10 for the 8-pass v_mfma_f32_16x16x4_f32, 6 for 16x16x32_bf16) but lost them across
// the s_branch that leaves a loop / branch around an MFMA chain (polyd_dgrad.hip: `ds_write_b128 v, a[0:3]` two instructions behind the last v_mfma; the -O1 build of the corner
// kernel: v_accvgpr_read one instruction behind it).  Kernels whose accumulator leaves such a region and is consumed at once spell the wait states out on the accumulator itself:
// a read-write operand in its AGPRs, so every later use depends on the asm and the asm on the last MFMA.  scripts/mfma_hazard_scan.py + tests/test_mfma_hazards.py check the assembly.
__device__ __forceinline__ void mfma_result_fence(f32x4& acc) { asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc)); }
