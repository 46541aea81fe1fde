// Stand-alone reproducer of the first-read effect of DESIGN.md section 7 - no library, one process, two streams, plain HIP C++:
//   victim    a GEMV in the shape of linear_small_kernel<4> (x staged in LDS and read back with one 128-bit LDS read per lane and row,
//             four weight rows per wave streamed with 128-bit loads; W = 1 and x = 1 + 1000 (k mod 4) + lane, so every output is one known
//             integer and a wrong one says which products it lost), with and without the guard (one v_mov of x.y / x.w before their use)
//   aggressor a 4-wave workgroup with ~230 live VGPRs and 64 KiB of LDS that loops v_mfma_f32_16x16x32_bf16 (or, as the control, VALU only)
// Expected on an MI355X: control 0 wrong, MFMA aggressor + unguarded victim: most launches wrong (lost = x.y / x.w products of lanes 48..63),
// MFMA aggressor + guarded victim: 0 wrong.
//   hipcc --offload-arch=gfx950 -O3 tools/canary/first_read_repro.hip -o /tmp/first_read_repro && /tmp/first_read_repro [seconds per case]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 1536, NC = 58368, ROWS = 2;

template <bool GUARD>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [ROWS][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < ROWS * K / 4; i += 256) reinterpret_cast<float4*>(xs)[i] = reinterpret_cast<const float4*>(x)[i];
  __syncthreads();
  for (int it = 0; it < 4; ++it) {
    const int n0 = ((blockIdx.x * 4 + it) * 4 + wave) * 4;
    float acc[4][ROWS] = {};
    for (int c = lane; c < K / 4; c += 64) {
      float4 w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + static_cast<long long>(n0 + j) * K) + c);
        w[j] = make_float4(t.x, t.y, t.z, t.w);
      }
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        float4 xv = reinterpret_cast<const float4*>(xs + i * K)[c];
        if constexpr (GUARD) {
          float t0, t1;
          asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(t0), "=&v"(t1), "+v"(xv.y), "+v"(xv.w));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][i] += (w[j].x * xv.x + w[j].y * xv.y) + (w[j].z * xv.z + w[j].w * xv.w);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        float s = acc[j][i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) y[static_cast<long long>(i) * NC + n0 + j] = s;
      }
  }
}

template <bool MFMA>
__global__ __launch_bounds__(256, 2) void aggressor(float* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  f32x4 acc[56];
#pragma unroll
  for (int i = 0; i < 56; ++i) acc[i] = f32x4{(float)(i + lane), 1.f, 2.f, 3.f};
  reinterpret_cast<float*>(lds)[threadIdx.x] = (float)lane;
  __syncthreads();
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    if constexpr (MFMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 56; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[threadIdx.x] = s[0];
}

template <bool GUARD, bool MFMA>
static void run_case(const char* name, double seconds, const float* x, const float* W, float* y, float* sink, float want, hipStream_t sv, hipStream_t sa) {
  std::vector<float> h(static_cast<size_t>(ROWS) * NC);
  unsigned long long launches = 0, bad_launches = 0, bad_words = 0;
  std::map<long long, unsigned long long> lost;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipLaunchKernelGGL(aggressor<MFMA>, dim3(256), dim3(256), 65536, sa, sink, 4000);  // ~2 ms on the second stream
    (void)hipMemsetAsync(y, 0, h.size() * 4, sv);
    hipLaunchKernelGGL(victim<GUARD>, dim3(NC / 64), dim3(256), ROWS * K * 4, sv, x, W, y);
    (void)hipMemcpyAsync(h.data(), y, h.size() * 4, hipMemcpyDeviceToHost, sv);
    (void)hipStreamSynchronize(sv);
    ++launches;
    unsigned long long bad = 0;
    for (float v : h)
      if (v != want) { ++bad; ++lost[static_cast<long long>(want - v)]; }
    bad_words += bad;
    bad_launches += bad != 0;
    (void)hipStreamSynchronize(sa);
  }
  printf("%-58s %llu of %llu victim launches wrong, %llu wrong outputs", name, bad_launches, launches, bad_words);
  if (!lost.empty()) {
    printf("; want - got, most frequent:");
    std::vector<std::pair<unsigned long long, long long>> v;
    for (auto& kv : lost) v.push_back({kv.second, kv.first});
    std::sort(v.rbegin(), v.rend());
    for (size_t i = 0; i < v.size() && i < 4; ++i) printf(" %lld (x%llu)", v[i].second, v[i].first);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  std::vector<float> hx(ROWS * K), hW(static_cast<size_t>(NC) * K, 1.0f);
  double want = 0;
  for (int k = 0; k < K; ++k) { hx[k] = hx[K + k] = 1.f + 1000.f * (k % 4) + (k % 256) / 4; want += hx[k]; }
  float *x, *W, *y, *sink;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&W, hW.size() * 4); (void)hipMalloc(&y, static_cast<size_t>(ROWS) * NC * 4); (void)hipMalloc(&sink, 4096);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipStream_t sv, sa;
  (void)hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  printf("every output should be %.0f; a lost x.y (x.w) product of lanes 48..63 of one iteration shows as 16904 (48904)\n", want);
  run_case<false, false>("unguarded victim next to a VALU-only kernel (control):", seconds, x, W, y, sink, (float)want, sv, sa);
  run_case<false, true>("unguarded victim next to an MFMA-streaming kernel:", seconds, x, W, y, sink, (float)want, sv, sa);
  run_case<true, true>("guarded victim (one v_mov of x.y / x.w) next to the same:", seconds, x, W, y, sink, (float)want, sv, sa);
  return 0;
}
