// experiment: one C++ process (no Python, no torch), the victim = ldc_linear_small of a library build given on the command line, the aggressor =
// synth_launch of /tmp/libsynth.so (tools/canary/synthetic_aggressor.hip built -shared), two streams
//   hipcc tools/canary/first_read_repro_dl.cpp -o /tmp/repro_dl -ldl && /tmp/repro_dl <victim library> [mode] [aggressor library]
//   tools/canary/build_code_object_variants.sh builds the victim (guard compiled out) and the aggressor as two libraries, as one library of two
//   translation units, and as one library of ONE translation unit (one code object)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int (*ls_fn)(const float*, int, const float*, const float*, const float*, int, float*, int, int, int, int, int, void*);
typedef int (*synth_fn)(int, int, int, void*);
int main(int argc, char** argv) {
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  void* syn = dlopen(argc > 3 ? argv[3] : "/tmp/libsynth.so", RTLD_NOW | RTLD_LOCAL);  // (the same path as argv[1]: both kernels from one library)
  if (!lib || !syn) { printf("dlopen failed: %s\n", dlerror()); return 1; }
  ls_fn ls = (ls_fn)dlsym(lib, "ldc_linear_small");
  synth_fn synth = (synth_fn)dlsym(syn, "synth_launch");
  const int mode = argc > 2 ? atoi(argv[2]) : 1;
  const int K = 1536, NC = 58368, ROWS = 2;
  std::vector<float> hx(ROWS * K), hW((size_t)NC * K, 1.0f), h((size_t)ROWS * NC);
  double want = 0;
  for (int k = 0; k < K; ++k) { hx[k] = hx[K + k] = 1.f + 1000.f * (k % 4) + (k % 256) / 4; want += hx[k]; }
  float *x, *W, *y;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&W, hW.size() * 4); (void)hipMalloc(&y, h.size() * 4);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStream_t sv = nullptr;
  unsigned long long launches = 0, bad_l = 0, bad_w = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) {
    for (int q = 0; q < 16; ++q) {
      synth(mode, 2000, 256, sa);
      (void)hipMemsetAsync(y, 0, h.size() * 4, sv);
      if (ls(x, ROWS, W, nullptr, nullptr, 0, y, ROWS, NC, K, 0, 0, sv)) { printf("launch failed\n"); return 1; }
      (void)hipMemcpyAsync(h.data(), y, h.size() * 4, hipMemcpyDeviceToHost, sv);
      (void)hipStreamSynchronize(sv);
      ++launches;
      unsigned long long bad = 0;
      for (float v : h) bad += v != (float)want;
      bad_w += bad; bad_l += bad != 0;
    }
    (void)hipDeviceSynchronize();
  }
  printf("%s next to synthetic aggressor mode %d from %s, one C++ process: %llu of %llu launches wrong, %llu wrong outputs\n", argv[1], mode, argc > 3 ? argv[3] : "/tmp/libsynth.so", bad_l, launches, bad_w);
  return 0;
}
