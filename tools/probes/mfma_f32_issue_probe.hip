// Probe (round 6): cycles per v_mfma_f32_16x16x4_f32 on one SIMD with ONE wave per SIMD vs TWO waves per SIMD, independent accumulators (two
// chains alternating, as the paired k-step of gemm_bf16x3_v3.hip issues them) - is there an arbitration bubble when the issuing wave changes?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_issue_probe.hip -o tools/probes/build/mfma_f32_issue_probe && ./mfma_f32_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc[j % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % CHAINS], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS>
void run(int threads, const char* what) {
  const int blocks = 256, iters = 4096;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * threads * sizeof(float));
  hipMalloc(&cyc, blocks * 8 * sizeof(unsigned long long));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = double(h[h.size() / 2]);
  const int waves_per_simd = threads / 256;
  printf("%-44s %d accumulation chain(s): median %.0f cycles per wave for %d MFMAs = %.2f cycles per MFMA per wave = %.2f cycles of the SIMD per MFMA (floor 32)\n", what, CHAINS, med,
         iters * 16, med / (iters * 16), med / (iters * 16) / waves_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(256, "one wave per SIMD (256 threads per CU)");
  run<2>(256, "one wave per SIMD (256 threads per CU)");
  run<4>(256, "one wave per SIMD (256 threads per CU)");
  run<1>(512, "two waves per SIMD (512 threads per CU)");
  run<2>(512, "two waves per SIMD (512 threads per CU)");
  run<4>(512, "two waves per SIMD (512 threads per CU)");
  return 0;
}
