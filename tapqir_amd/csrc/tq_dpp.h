// tq_dpp.h -- device-only helpers: cross-lane sums over the 16 lanes of a DPP row, float2 values for the packed
// v_pk_{fma,mul,add}_f32 instructions.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL>
__device__ __forceinline__ float tq_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float tq_group_sum16(float v) {
  // sum over the 16 lanes of a unit = one DPP row: data-parallel-primitive adds, no LDS crossbar.
  v += tq_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += tq_dpp<0x4E>(v);   // quad_perm [2,3,0,1]   -> quad sums
  v += tq_dpp<0x141>(v);  // row_half_mirror       -> sums of 8
  v += tq_dpp<0x140>(v);  // row_mirror            -> sum of 16, in every lane
  return v;
}

// sum over the LANES (16 or 64) lanes of a unit, in every lane: the DPP row sum, then -- a unit per wave -- the four rows
template <int LANES>
__device__ __forceinline__ float tq_group_sum(float v) {
  v = tq_group_sum16(v);
  if (LANES == 64) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
  }
  return v;
}

// two FP32 values per lane that the compiler maps onto packed instructions (two operations per lane and issue slot)
typedef float tq_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tq_f2 tq2(float a) { return (tq_f2){a, a}; }
__device__ __forceinline__ tq_f2 tq2_rcp(tq_f2 a) { return (tq_f2){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ tq_f2 tq2_log2(tq_f2 a) { return (tq_f2){__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(a.y)}; }
__device__ __forceinline__ tq_f2 tq2_exp2(tq_f2 a) { return (tq_f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
