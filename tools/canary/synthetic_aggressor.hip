// Synthetic aggressor for the first-read effect: a 4-wave workgroup with ~250 live VGPRs per lane and 64 KiB of LDS
// (two workgroups per CU, one wave per SIMD each - the footprint of attn_fwd_split_kernel<1, *>, which leaves half a SIMD's registers to
// another process's waves) that loops ONE instruction class.  Which class, next to the unguarded GEMV (A/B build -DLDC_LS_NO_FIRST_READ),
// brings the wrong words back?
//   hipcc --offload-arch=gfx950 -O3 tools/canary/synthetic_aggressor.hip -o /tmp/synth_aggr && /tmp/synth_aggr <mode> <seconds>
//   mode 0 VALU fma only | 1 + v_mfma_f32_16x16x32_bf16 | 2 + ds_read_b64_tr_b16 | 3 + global_load_lds_dwordx4 | 4 + v_exp_f32
//        5 + ds_read_b128 / ds_write_b128 | 6 all of 1..4 together
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void aggressor(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[56];  // 224 live registers
#pragma unroll
  for (int i = 0; i < 56; ++i) acc[i] = f32x4{(float)(i + lane), 1.f, 2.f, 3.f};
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  float e = 0.001f * lane;
  const float* g = src + (size_t)blockIdx.x * 4096;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;  // every mode: all 224 registers stay live
    if constexpr (MODE == 1 || MODE == 6) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    if constexpr (MODE == 2 || MODE == 6) {
#pragma unroll
      for (int i = 0; i < 28; ++i) {
        s16x4 t;
        const unsigned addr = (unsigned)(((lane * 8 + i * 512 + wave * 8192) & 0xfff8));
        asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr));
        acc[i][0] += (float)t[0];
      }
    }
    if constexpr (MODE == 3 || MODE == 6) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((it * 8 + i) & 3) * 1024 + lane * 4 + wave * 256),
                                         (__attribute__((address_space(3))) void*)(lds + 32768 + wave * 4096 + i * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc[it & 7][1] += reinterpret_cast<float*>(lds)[8192 + lane];
    }
    if constexpr (MODE == 4 || MODE == 6) {
#pragma unroll
      for (int i = 0; i < 56; ++i) { e = __builtin_amdgcn_exp2f(e * 0.5f); acc[i][2] += e; }
    }
    if constexpr (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 28; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(lds + ((lane * 16 + i * 1024 + wave * 8192) & 0x7ff0));
        acc[i] += t;
        *reinterpret_cast<f32x4*>(lds + 32768 + ((lane * 16 + i * 1024) & 0x7ff0)) = acc[i];
      }
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 56; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[threadIdx.x] = s[0];
}

template <int MODE>
static void run(double seconds, const float* src, float* sink, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 20; ++k, ++launches) hipLaunchKernelGGL(aggressor<MODE>, dim3(512), dim3(256), 65536, 0, src, sink, iters);
    hipDeviceSynchronize();
  }
  printf("synthetic aggressor mode %d: %llu launches\n", MODE, launches);
}

// the same kernels as a function, for an aggressor on a second stream of the VICTIM's own process (first_read_repro.hip):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/canary/synthetic_aggressor.hip -o /tmp/libsynth.so
static float *g_src = nullptr, *g_sink = nullptr;
extern "C" int synth_launch(int mode, int iters, int blocks, void* stream) {
  if (!g_src) {
    hipMalloc(&g_src, 512 * 4096 * 4 + 65536);
    hipMemset(g_src, 0, 512 * 4096 * 4 + 65536);
    hipMalloc(&g_sink, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  }
  if (blocks > 512) blocks = 512;
  if (mode == 1) hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 65536, static_cast<hipStream_t>(stream), g_src, g_sink, iters);
  else hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 65536, static_cast<hipStream_t>(stream), g_src, g_sink, iters);
  return (int)hipGetLastError();
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
  const int iters = argc > 3 ? atoi(argv[3]) : 200;
  float *src, *sink;
  hipMalloc(&src, 512 * 4096 * 4 + 65536);
  hipMemset(src, 0, 512 * 4096 * 4 + 65536);
  hipMalloc(&sink, 4096);
  switch (mode) {
    case 0: run<0>(seconds, src, sink, iters); break;
    case 1: run<1>(seconds, src, sink, iters); break;
    case 2: run<2>(seconds, src, sink, iters); break;
    case 3: run<3>(seconds, src, sink, iters); break;
    case 4: run<4>(seconds, src, sink, iters); break;
    case 5: run<5>(seconds, src, sink, iters); break;
    default: run<6>(seconds, src, sink, iters); break;
  }
  return 0;
}
