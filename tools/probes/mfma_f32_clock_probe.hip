// Probe (round 6): which shader clock does a DENSE v_mfma_f32_16x16x4_f32 stream sustain on the whole chip, as a function of the operand
// values?  Nothing but MFMAs (two waves per SIMD, four accumulation chains, 16 operand register pairs cycling so that the multiplier
// inputs change on every instruction, as a GEMM's do): no LDS, no global loads in the loop.  The rate this loop reaches with random
// operands is the ceiling any fp32 GEMM / attention kernel on this chip can reach - the 157.3 TFLOP/s of the data sheet are 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_clock_probe.hip -o tools/probes/build/mfma_f32_clock_probe && ./mfma_f32_clock_probe
// Clock = s_memtime (shader cycles) over s_memrealtime (100 MHz) around the loop; each pattern runs ~1.5 s before it is read (DVFS settles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(const float* __restrict__ ab, float* out, unsigned long long* stamps, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a[16], b[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    a[j] = ab[(j * 512 + threadIdx.x) * 2];
    b[j] = ab[(j * 512 + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[j & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 8 + threadIdx.x / 64;
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

static float gauss() {
  const float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.0f * logf(u)) * cosf(6.2831853f * v);
}

int main() {
  const int blocks = 256, threads = 512, iters = 8192;  // 131072 MFMAs per wave: ~4 ms per launch
  const int n = 16 * 512 * 2;
  float *ab, *out;
  unsigned long long* st;
  hipMalloc(&ab, n * sizeof(float));
  hipMalloc(&out, blocks * threads * sizeof(float));
  hipMalloc(&st, blocks * 8 * 2 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"all operands 0.0", "all operands 1.0", "one random pair, repeated (no operand toggling)", "random normal, 16 register pairs cycling",
                         "random normal x 1e-3 (small products)"};
  for (int pat = 0; pat < 5; ++pat) {
    std::vector<float> h(n);
    srand(1234);
    const float one_a = gauss(), one_b = gauss();
    for (int i = 0; i < n; ++i) h[i] = pat == 0 ? 0.f : pat == 1 ? 1.f : pat == 2 ? ((i & 1) ? one_b : one_a) : pat == 3 ? gauss() : 1e-3f * gauss();
    hipMemcpy(ab, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 350; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, ab, out, st, iters);  // ~1.5 s of settling
    hipEventRecord(e0);
    const int timed = 20;
    for (int rep = 0; rep < timed; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, ab, out, st, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(blocks * 8 * 2);
    hipMemcpy(s.data(), st, s.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, mhz;
    for (int w = 0; w < blocks * 8; ++w) {
      cyc.push_back(double(s[2 * w]));
      mhz.push_back(double(s[2 * w]) / (double(s[2 * w + 1]) / 100.0));
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(mhz.begin(), mhz.end());
    const double mfmas = double(iters) * 16.0, flops = double(timed) * blocks * 8 * mfmas * 2.0 * 16 * 16 * 4;
    printf("%-52s %6.2f SIMD cycles per MFMA, shader clock min %4.0f median %4.0f max %4.0f MHz, %6.1f TFLOP/s over %d launches (%.2f ms each)\n", names[pat],
           cyc[cyc.size() / 2] / mfmas / 2.0, mhz.front(), mhz[mhz.size() / 2], mhz.back(), flops / (ms * 1e-3) / 1e12, timed, ms / timed);
    fflush(stdout);
  }
  return 0;
}
