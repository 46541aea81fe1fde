// TEST FIXTURE, not part of libladcast_hip.so: the MFMA-streaming "aggressor" of tools/canary/synthetic_aggressor.hip as a tiny shared library
// (libladcast_test_aggressor.so, `make aggressor`; loaded only by tests/test_gpu_multistream_victims.py).  A 4-wave workgroup with ~230 live
// VGPRs per lane and 64 KiB of LDS - the footprint of the attention kernel next to which the first-read effect was found (DESIGN.md,
// "first-read effect"): at one workgroup per CU it leaves half of every SIMD's register file and 96 KiB of LDS to waves of OTHER kernels, which
// the test launches on a second stream.  Its loop is VALU fma + v_mfma_f32_16x16x32_bf16.
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void ldc_test_aggressor_kernel(float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  f32x4 acc[56];  // 224 live registers
#pragma unroll
  for (int i = 0; i < 56; ++i) acc[i] = f32x4{(float)(i + lane), 1.f, 2.f, 3.f};
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (lane + i));
    b[i] = (__bf16)(0.002f * (lane - i));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;  // all 224 registers stay live
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 56; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[threadIdx.x] = s[0] + reinterpret_cast<float*>(lds)[lane];
}

// one launch of `blocks` workgroups x `iters` loop iterations on `stream` (~0.45 us per iteration per workgroup); returns the HIP status
extern "C" int ldc_test_aggressor_launch(int blocks, int iters, float* sink, void* stream) {
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ldc_test_aggressor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL(ldc_test_aggressor_kernel, dim3(blocks), dim3(256), 65536, static_cast<hipStream_t>(stream), sink, iters);
  return static_cast<int>(hipGetLastError());
}
