// Streaming-load canary (debug aid for the GPU-sharing question): the access pattern of the wide AdaLN GEMV (linear_small_kernel<4>:
// 256 threads, a wave streams 4 rows of a [58368][1536] fp32 matrix with four non-temporal 16-byte loads per lane in flight, 12 KiB of
// dynamic LDS, 4 column groups per workgroup) with every loaded word CHECKED against a pattern computed from its index, instead of
// multiplied.  Run it while other processes loop the 4-wave split attention: a wrong word is reported with its index, lane and value -
// is the data that reaches the registers wrong, or the arithmetic after it?
//   hipcc --offload-arch=gfx950 -O3 tools/canary/stream_canary.hip -o /tmp/stream_canary && /tmp/stream_canary <seconds> [nt=1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

constexpr int K = 1536, NC = 58368;
__host__ __device__ inline unsigned pat(unsigned long long i) { return (unsigned)(i * 2654435761ull) ^ 0x5bd1e995u ^ (unsigned)(i >> 7); }

struct Rec { unsigned n, k, got, want, lane, launch; };

template <bool NT>
__global__ __launch_bounds__(256) void stream_check(const unsigned* __restrict__ W, unsigned* count, Rec* recs, int max_recs, unsigned launch) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * K; i += 256) xs[i] = 1.0f;  // the staged x of the GEMV
  __syncthreads();
  float keep = 0.f;
  for (int it = 0; it < 4; ++it) {
    const int n0 = ((blockIdx.x * 4 + it) * 4 + wave) * 4;
    for (int c = lane; c < K / 4; c += 64) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4* p = reinterpret_cast<const u32x4*>(W + (unsigned long long)(n0 + j) * K) + c;
        w[j] = NT ? __builtin_nontemporal_load(p) : *p;
      }
      const float4 xv = reinterpret_cast<const float4*>(xs)[c];
      keep += xv.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long base = (unsigned long long)(n0 + j) * K + 4 * c;
        const unsigned g[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned want = pat(base + e);
          if (g[e] != want) {
            const unsigned slot = atomicAdd(count, 1u);
            if (slot < (unsigned)max_recs) recs[slot] = Rec{(unsigned)(n0 + j), (unsigned)(4 * c + e), g[e], want, (unsigned)lane, launch};
          }
        }
      }
    }
  }
  if (keep == -1.f) count[1] = 1;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const bool nt = argc > 2 ? atoi(argv[2]) != 0 : true;
  const size_t words = (size_t)NC * K;
  unsigned* W;
  hipMalloc(&W, words * 4);
  unsigned* h = (unsigned*)malloc(words * 4);
  for (size_t i = 0; i < words; ++i) h[i] = pat(i);
  hipMemcpy(W, h, words * 4, hipMemcpyHostToDevice);
  free(h);
  const int max_recs = 4096;
  unsigned* count;
  Rec* recs;
  hipMalloc(&count, 16);
  hipMalloc(&recs, sizeof(Rec) * max_recs);
  hipMemset(count, 0, 16);
  unsigned long long launches = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 16; ++k, ++launches) {
      if (nt) hipLaunchKernelGGL(stream_check<true>, dim3(NC / 64), dim3(256), 2 * K * 4, 0, W, count, recs, max_recs, (unsigned)launches);
      else hipLaunchKernelGGL(stream_check<false>, dim3(NC / 64), dim3(256), 2 * K * 4, 0, W, count, recs, max_recs, (unsigned)launches);
    }
    hipDeviceSynchronize();
  }
  unsigned n_bad;
  hipMemcpy(&n_bad, count, 4, hipMemcpyDeviceToHost);
  printf("streaming-load canary (%s loads), W at %p: %llu launches, %u wrong words\n", nt ? "non-temporal" : "plain", (void*)W, launches, n_bad);
  const int show = n_bad < (unsigned)max_recs ? (int)n_bad : max_recs;
  Rec* hr = (Rec*)malloc(sizeof(Rec) * max_recs);
  hipMemcpy(hr, recs, sizeof(Rec) * max_recs, hipMemcpyDeviceToHost);
  unsigned lanes[64] = {0}, comp[4] = {0}, par[2] = {0}, zero = 0;
  for (int i = 0; i < show; ++i) { ++lanes[hr[i].lane & 63]; ++comp[hr[i].k & 3]; ++par[hr[i].n & 1]; zero += hr[i].got == 0; }
  if (show) {
    printf("  of the first %d: got == 0 in %u; by row parity: even %u odd %u; by component: %u %u %u %u\n  by lane:", show, zero, par[0], par[1], comp[0], comp[1], comp[2], comp[3]);
    for (int l = 0; l < 64; ++l) printf(" %u", lanes[l]);
    printf("\n");
    for (int i = 0; i < show && i < 40; ++i)
      printf("  launch %u row %u k %u lane %u: got 0x%08x want 0x%08x\n", hr[i].launch, hr[i].n, hr[i].k, hr[i].lane, hr[i].got, hr[i].want);
  }
  return 0;
}
