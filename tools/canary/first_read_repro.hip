// Reproducer of the first-read effect of DESIGN.md section 7: ONE executable, one translation unit (one code object), one process, two streams,
// no Python, no library to load.
//   victim    the library's own GEMV source (#include of csrc/rowops.hip: ldc_linear_small -> linear_small_kernel<4>), 2 rows x 58 368 x 1 536 on
//             the null stream.  W = 1 and x = 1 + 1000 (k mod 4) + lane, so every output is one known integer and a wrong one says which
//             products it lost (16904 / 48904 = the x.y / x.w products of lanes 48..63 of one iteration).
//   aggressor tools/canary/synthetic_aggressor.hip (#included): 4-wave workgroups with ~230 live VGPRs and 64 KiB of LDS looping
//             v_mfma_f32_16x16x32_bf16 (mode 1) or VALU only (mode 0, the control), launched back to back on a second stream.
// Build it twice - the guard is a compile-time switch of the library source:
//   F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iladcast_amd/csrc -Itools/canary -Wno-unused-value"
//   hipcc $F -DLDC_AB_BUILD -DLDC_LS_NO_FIRST_READ tools/canary/first_read_repro.hip -o /tmp/repro_unguarded     # guard compiled out
//   hipcc $F                                        tools/canary/first_read_repro.hip -o /tmp/repro_guarded       # as shipped
// Measured on an MI355X (profiles/r04_z_gpu_sharing_first_read.log): unguarded 2880 of 2880 launches wrong next to the MFMA aggressor, 0 of
// 10 032 next to the VALU-only one; guarded 0 wrong.  The effect depends on the generated code of the victim: a hand-simplified copy of the
// same loop (same instructions in the hot loop as far as one can read them, other register numbers) shows nothing in this very harness.
#include "rowops.hip"
#define main synth_main_unused
#include "synthetic_aggressor.hip"
#undef main
#include <algorithm>
#include <map>
#include <vector>

static void run_case(int mode, double seconds, const float* x, const float* W, float* y, float want, hipStream_t sa) {
  const int K = 1536, NC = 58368, ROWS = 2;
  std::vector<float> h(static_cast<size_t>(ROWS) * NC);
  unsigned long long launches = 0, bad_l = 0, bad_w = 0;
  std::map<long long, unsigned long long> lost;
  hipStream_t sv = nullptr;  // the victim on the null stream
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int q = 0; q < 16; ++q) {  // the aggressor stream is not drained inside this loop
      synth_launch(mode, 2000, 256, sa);
      (void)hipMemsetAsync(y, 0, h.size() * 4, sv);
      if (ldc_linear_small(x, ROWS, W, nullptr, nullptr, 0, y, ROWS, NC, K, 0, 0, sv)) { printf("launch failed\n"); exit(1); }
      (void)hipMemcpyAsync(h.data(), y, h.size() * 4, hipMemcpyDeviceToHost, sv);
      (void)hipStreamSynchronize(sv);
      ++launches;
      unsigned long long bad = 0;
      for (float v : h)
        if (v != want) { ++bad; ++lost[static_cast<long long>(want - v)]; }
      bad_w += bad;
      bad_l += bad != 0;
    }
    (void)hipDeviceSynchronize();
  }
#if defined(LDC_AB_BUILD) && defined(LDC_LS_NO_FIRST_READ)
  const char* guard = "UNGUARDED";
#else
  const char* guard = "guarded";
#endif
  printf("%s GEMV next to the %s aggressor: %llu of %llu launches wrong, %llu wrong outputs", guard, mode ? "MFMA-streaming" : "VALU-only", bad_l, launches, bad_w);
  if (!lost.empty()) {
    std::vector<std::pair<unsigned long long, long long>> v;
    for (auto& kv : lost) v.push_back({kv.second, kv.first});
    std::sort(v.rbegin(), v.rend());
    printf("; want - got, most frequent:");
    for (size_t i = 0; i < v.size() && i < 4; ++i) printf(" %lld (x%llu)", v[i].second, v[i].first);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  const int K = 1536, NC = 58368, ROWS = 2;
  std::vector<float> hx(ROWS * K), hW(static_cast<size_t>(NC) * K, 1.0f);
  double want = 0;
  for (int k = 0; k < K; ++k) { hx[k] = hx[K + k] = 1.f + 1000.f * (k % 4) + (k % 256) / 4; want += hx[k]; }
  float *x, *W, *y;
  (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&W, hW.size() * 4); (void)hipMalloc(&y, static_cast<size_t>(ROWS) * NC * 4);
  (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  printf("every output should be %.0f\n", want);
  run_case(0, seconds, x, W, y, (float)want, sa);
  run_case(1, seconds, x, W, y, (float)want, sa);
  return 0;
}
