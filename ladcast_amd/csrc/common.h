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

// 4 consecutive values (columns c .. c + 3, c % 4 == 0) of a row into the split activation format (LDC_FMT_SPLIT: columns
// 8 g .. 8 g + 7 in 32 bytes [hi x8 | lo x8]): 8 bytes of hi and 8 bytes of lo.  `row` = start of the row.
__device__ __forceinline__ void ldc_store_split4(unsigned char* row, int c, float x0, float x1, float x2, float x3) {
  float r0, r1, r2, r3;
  const unsigned h0 = ldc_split_pair(x0, x1, r0, r1), h1 = ldc_split_pair(x2, x3, r2, r3);
  unsigned char* g = row + 4 * (c & ~7) + 2 * (c & 7);
  *reinterpret_cast<uint2*>(g) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(g + 16) = make_uint2(ldc_pack_pair(r0, r1), ldc_pack_pair(r2, r3));
}
// the same for a row whose width C is 4 mod 8: the caller of the LAST 4 columns also clears the other half of the group, so the
// pad columns C .. C + 3 read as zeros (the GEMMs / convs read whole 8-column groups)
__device__ __forceinline__ void ldc_zero_split4(unsigned char* row, int c) {
  unsigned char* g = row + 4 * (c & ~7) + 2 * (c & 7);
  *reinterpret_cast<uint2*>(g) = make_uint2(0u, 0u);
  *reinterpret_cast<uint2*>(g + 16) = make_uint2(0u, 0u);
}

// the same two helpers with the operand format as an argument: LDC_FMT_SPLIT, or LDC_FMT_BF16 (plain bf16 row: 8 bytes at byte 2 c)
__device__ __forceinline__ void ldc_store_fmt4(unsigned char* row, int c, int fmt, float x0, float x1, float x2, float x3) {
  if (fmt == LDC_FMT_BF16) *reinterpret_cast<uint2*>(row + 2 * c) = make_uint2(ldc_pack_pair(x0, x1), ldc_pack_pair(x2, x3));
  else ldc_store_split4(row, c, x0, x1, x2, x3);
}
__device__ __forceinline__ void ldc_zero_fmt4(unsigned char* row, int c, int fmt) {
  if (fmt == LDC_FMT_BF16) *reinterpret_cast<uint2*>(row + 2 * c) = make_uint2(0u, 0u);
  else ldc_zero_split4(row, c);
}

// Source pixel (row-major index inside one image) of tap (ky, kx) of a ks x ks SphereConv2d at output pixel (h, w): rows past a
// pole are mirrored and rolled by W / 2, columns wrap, and at the two pole rows the kernel rows that reach over the pole are
// flipped left-right (models/sphere_conv.py:62-129,174-192).  Same rule as gemm_f32.hip / dcae.hip; the wrap is one conditional
// add / subtract instead of a modulo (ks / 2 <= W / 2, checked by the callers).
__device__ __forceinline__ int ldc_sphere_src_pixel(int h, int w, int ky, int kx, int H, int W, int ks) {
  const int p = ks >> 1;
  if ((h == 0 && ky < p) || (h == H - 1 && ky >= ks - p)) kx = ks - 1 - kx;
  int r = h + ky - p;
  int c = w + kx - p;
  if (r < 0) {
    r = -1 - r;
    c -= W >> 1;
  } else if (r >= H) {
    r = 2 * H - 1 - r;
    c -= W >> 1;
  }
  if (c < 0) c += W;
  if (c >= W) c -= W;
  return r * W + c;
}

// conv_halo.hip: the halo-staged 3 x 3 conv (LDC_ERR_UNSUPPORTED: shape not served - run the gathered conv of gemm_bf16x3_v3.hip)
int ldc_conv_halo_dispatch(const float* X, const void* Wp, const float* bias, const float* R, float* Y, int B, int H, int W, int cin,
                           int ldx, int cout, int ldy, int ldr, int act, int in_fmt, int out_fmt, void* workspace, long long workspace_bytes,
                           void* stream);

// Measurement switches (tile height / unit ranges / tile order / kernel choice forced from the environment) exist only in the A/B build
// (`make ab` -> libladcast_hip_ab.so, loaded by tools/ through LDC_LIB_PATH); the shipped library reads no environment variable.
#ifdef LDC_AB_BUILD
#define LDC_AB_GETENV(name) getenv(name)
#else
#define LDC_AB_GETENV(name) (static_cast<const char*>(nullptr))
#endif

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
