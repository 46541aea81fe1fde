// LDS canary (debug aid for the GPU-sharing reproducibility question): every workgroup fills its LDS allocation with a pattern, idles for a
// while, and checks it.  Run it while OTHER processes run a suspect kernel on the same GPU: a kernel that writes LDS outside its own
// allocation (e.g. an LDS-DMA landing past the end) shows up here as corrupted words in a workgroup that shares the CU with it.
//   hipcc --offload-arch=gfx950 -O2 tools/canary/lds_canary.hip -o /tmp/lds_canary && /tmp/lds_canary <seconds> <lds KiB> <spin>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

// mode 1: a GLOBAL-memory canary instead - no LDS; every thread re-reads words of a read-only pattern buffer `gbuf` (gwords words) while
// it idles and counts the reads that do not return the pattern: data of ANOTHER process's address space arriving through a shared cache
// would show up here
__global__ void canary(unsigned* report, int words, int spin, unsigned tag, const unsigned* gbuf, int gwords, int mode) {
  extern __shared__ unsigned lds[];
  if (mode == 2) {
    // cross-lane canary: ds_bpermute_b32 (what __shfl_xor compiles to: the LDS unit's crossbar, no LDS memory) in a loop; every lane
    // checks that it received its partner's value.  The victim of the GPU-sharing effect (linear_small_kernel<4>) reduces with it.
    const unsigned lane = threadIdx.x & 63u;
    unsigned bad = 0, first = 0xffffffffu, val = 0;
    unsigned x = (blockIdx.x * 2654435761u) ^ ((threadIdx.x >> 6) * 40503u) ^ tag;  // wave-uniform
    for (int s = 0; s < spin; ++s) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned mine = x ^ (lane * 0x9E3779B1u) ^ (unsigned)(s * 64 + o);
        const unsigned want = x ^ ((lane ^ (unsigned)o) * 0x9E3779B1u) ^ (unsigned)(s * 64 + o);  // x is wave-uniform below
        const unsigned got = (unsigned)__shfl_xor((int)mine, o, 64);
        if (got != want) { ++bad; if ((unsigned)s < first) { first = s; val = got ^ want; } }
      }
    }
    if (bad) {
      atomicAdd(&report[0], bad);
      atomicAdd(&report[1], 1u);
      atomicMin(&report[2], first);
      report[3] = val;
      atomicMax(&report[4], first);
    }
    return;
  }
  if (mode == 1) {
    unsigned bad = 0, first = 0xffffffffu, val = 0;
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    for (int s = 0; s < spin; ++s) {
      idx = (idx * 1664525u + 1013904223u) % (unsigned)(gwords / 4) * 4u;
      const uint4 v = *reinterpret_cast<const uint4*>(gbuf + idx);
      const unsigned e = 0xFACE0000u;
      if (v.x != (e ^ idx) || v.y != (e ^ (idx + 1)) || v.z != (e ^ (idx + 2)) || v.w != (e ^ (idx + 3))) {
        ++bad;
        if (idx < first) { first = idx; val = v.x; }
      }
      __builtin_amdgcn_s_sleep(16);
    }
    if (bad) {
      atomicAdd(&report[0], bad);
      atomicAdd(&report[1], 1u);
      atomicMin(&report[2], first);
      report[3] = val;
      atomicMax(&report[4], first);
    }
    return;
  }
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xC0DE0000u ^ (unsigned)i ^ tag;
  __syncthreads();
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  unsigned bad = 0, first = 0xffffffffu, val = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) {
    const unsigned v = lds[i];
    if (v != (0xC0DE0000u ^ (unsigned)i ^ tag)) {
      ++bad;
      if ((unsigned)i < first) { first = i; val = v; }
    }
  }
  if (bad) {
    atomicAdd(&report[0], bad);
    atomicAdd(&report[1], 1u);
    atomicMin(&report[2], first);
    report[3] = val;
    atomicMax(&report[4], first);
  }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const int kib = argc > 2 ? atoi(argv[2]) : 16;
  const int spin = argc > 3 ? atoi(argv[3]) : 200;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const int words = kib * 256;
  unsigned* rep;
  hipMalloc(&rep, 64);
  const int gwords = 64 << 20;  // 256 MiB pattern buffer (mode 1)
  unsigned* gbuf = nullptr;
  if (mode == 1) {
    hipMalloc(&gbuf, (size_t)gwords * 4);
    unsigned* h = (unsigned*)malloc((size_t)gwords * 4);
    for (int i = 0; i < gwords; ++i) h[i] = 0xFACE0000u ^ (unsigned)i;
    hipMemcpy(gbuf, h, (size_t)gwords * 4, hipMemcpyHostToDevice);
    free(h);
  }
  hipFuncSetAttribute(reinterpret_cast<const void*>(canary), hipFuncAttributeMaxDynamicSharedMemorySize, kib * 1024);
  unsigned long long launches = 0, bad_words = 0, bad_wgs = 0;
  unsigned lo = 0xffffffffu, hi = 0, sample = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    unsigned init[5] = {0, 0, 0xffffffffu, 0, 0};
    hipMemcpy(rep, init, sizeof(init), hipMemcpyHostToDevice);
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(canary, dim3(1024), dim3(256), mode != 0 ? 0 : kib * 1024, 0, rep, words, spin, (unsigned)(launches + k) * 2654435761u, gbuf, gwords, mode);
    hipDeviceSynchronize();
    unsigned out[5];
    hipMemcpy(out, rep, sizeof(out), hipMemcpyDeviceToHost);
    launches += 50;
    bad_words += out[0];
    bad_wgs += out[1];
    if (out[1]) { if (out[2] < lo) lo = out[2]; if (out[4] > hi) hi = out[4]; sample = out[3]; }
  }
  printf("%s canary (spin %d): %d KiB per workgroup, %llu launches x 1024 workgroups: %llu corrupted words in %llu workgroups", mode == 2 ? "ds_bpermute (cross-lane)" : mode == 1 ? "global-load" : "lds", spin, kib, launches, bad_words, bad_wgs);
  if (bad_wgs) printf("; first corrupted word index in [%u, %u], a corrupted value 0x%08x", lo, hi, sample);
  printf("\n");
  return 0;
}
