// Shared helpers for the gfx950 kernels of libladcast_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ladcast_hip.h"

// stream-K workspace layout: [counter block (zeroed once, every call leaves it zero)][partial-sum slabs]
#define LDC_GEMM_COUNTER_BYTES (1 << 20)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDC_CHECK_PTR(p) \
  do {                   \
    if ((p) == nullptr) return LDC_ERR_ARG; \
  } while (0)
#define LDC_CHECK_ALIGN16(p) \
  do {                       \
    if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) return LDC_ERR_ALIGN; \
  } while (0)

static inline int ldc_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LDC_OK : -(1000 + static_cast<int>(e));
}

static inline int ldc_cdiv(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

// Activations of the GEMM epilogues, on the transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each): 64 values per
// lane go through these at the end of every MLP tile, where libm tanhf / a correctly rounded division cost ~40 / ~10
// instructions per value (a third of the epilogue).
//   silu(x) = x * sigmoid(x) = x / (1 + 2^(-x log2 e))
__device__ __forceinline__ float ldc_silu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
// tanh-approximate GELU, torch's 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written through the identity
// 0.5 (1 + tanh(u)) = sigmoid(2 u) = 1 / (1 + 2^(-2 u log2 e))
__device__ __forceinline__ float ldc_gelu_tanh(float v) {
  const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const float t = v * (c0 + c1 * v * v);
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
__device__ __forceinline__ float ldc_apply_act(float v, int act) {
  switch (act) {
    case LDC_ACT_SILU: return ldc_silu(v);
    case LDC_ACT_GELU_TANH: return ldc_gelu_tanh(v);
    case LDC_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}

// split-bf16 helpers: x = hi + lo, hi = bf16(x) (RNE), lo = bf16(x - hi); a pair packs into one dword (first value low)
__device__ __forceinline__ unsigned ldc_pack_pair(float a, float b) {
  typedef __bf16 ldc_bf16x2 __attribute__((ext_vector_type(2)));
  ldc_bf16x2 v;
  v.x = static_cast<__bf16>(a);
  v.y = static_cast<__bf16>(b);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned ldc_split_pair(float a, float b, float& ra, float& rb) {
  const unsigned u = ldc_pack_pair(a, b);
  ra = a - __uint_as_float(u << 16);
  rb = b - __uint_as_float(u & 0xffff0000u);
  return u;
}

// wave64 butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
