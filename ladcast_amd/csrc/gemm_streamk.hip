// Persistent "stream-K" scheduling of the fp32-MFMA NT GEMM for the small, awkward grids of the
// AR transformer (M = 1800 / 450 / 2250 rows per member: 48..864 output tiles for 256 CUs).
//
// Why: with one workgroup per 128x128 output tile the chip runs 0.2-0.85 full "waves" of tiles
// (measured 17-100 TF/s per shape against 122 TF/s for the same inner loop on a large grid,
// profiles/r01_a_gemm_attn_shapes.log).  Here the unit of work is one (tile, 32-deep k-step);
// the U units of up to LDC_GEMM_MAX_PROBLEMS problems (e.g. the pred-stream and cond-stream
// projections of a dual block, which use different weights) are laid end to end and cut into G
// equal contiguous ranges, G = 2 workgroups per CU, so every CU gets the same number of MFMAs and
// two waves per SIMD cover each other's staging stalls.
//
// A workgroup whose range covers a tile's full K applies the epilogue directly.  Otherwise it
// stores its raw accumulators to a caller-provided workspace slot (<= 2 slots per workgroup) and
// a second tiny kernel -- after a plain kernel boundary, so no in-launch inter-workgroup
// hand-off is needed -- sums the slots of each split tile in fixed order and runs the same
// epilogue.  Both kernels derive the schedule from (U, G) alone; results are deterministic.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDP = 36;
constexpr int STAGE_FLOATS = (BM + BN) * LDP;
constexpr int SLOT_FLOATS = BM * BN;
constexpr int MAXP = LDC_GEMM_MAX_PROBLEMS;

struct DevProblem {
  const float* A;
  const float* W;
  const float* bias;
  const float* gate;
  const float* R;
  float* C;
  ldc_gemm_desc d;
  int tm, tn, kt;
  long long unit0;  // first global unit of this problem
  long long tile0;  // first global tile of this problem
};

struct SKArgs {
  DevProblem pr[MAXP];
  int np;
  int G;
  long long U;
  long long tiles;
  float* ws;
};

__device__ __forceinline__ long long range_start(long long g, long long U, int G) { return (g * U) / G; }

// epilogue shared by the main and the fix-up kernel; acc layout = MFMA C/D layout of the 2x2 wave grid
__device__ __forceinline__ void tile_epilogue(const DevProblem& P, int b, int bm, int bn, const f32x16 (&acc)[2][2],
                                              int wm, int wn, int lane) {
  const int M = P.d.M, N = P.d.N;
  float* __restrict__ C = P.C + static_cast<long long>(b) * P.d.c_bs;
  const float* __restrict__ R = P.R ? P.R + static_cast<long long>(b) * P.d.r_bs : nullptr;
  const float* __restrict__ gate = P.gate ? P.gate + static_cast<long long>(b) * P.d.gate_bs : nullptr;
  const int act = P.d.act;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = bn * BN + wn * 64 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const float bias_n = P.bias ? P.bias[n] : 0.f;
    const float gate_n = gate ? gate[n] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        float v = acc[i][j][r] + bias_n;
        v = ldc_apply_act(v, act);
        if (gate) v *= gate_n;
        if (R) v += R[static_cast<long long>(m) * P.d.ldr + n];
        C[static_cast<long long>(m) * P.d.ldc + n] = v;
      }
    }
  }
}

__device__ __forceinline__ int find_problem_by_unit(const SKArgs& a, long long u) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < MAXP; ++k)
    if (k < a.np && u >= a.pr[k].unit0) pi = k;
  return pi;
}

__global__ __launch_bounds__(256, 2) void gemm_streamk_kernel(SKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.x;
  const long long u_begin = range_start(g, a.U, a.G);
  const long long u_end = range_start(g + 1, a.U, a.G);
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  long long u = u_begin;
  while (u < u_end) {
    const int pi = find_problem_by_unit(a, u);
    const DevProblem& P = a.pr[pi];
    const long long local = u - P.unit0;
    const int tile = static_cast<int>(local / P.kt);
    const int k0 = static_cast<int>(local - static_cast<long long>(tile) * P.kt);
    const long long left = u_end - u;
    const int k1 = (P.kt - k0 <= left) ? P.kt : k0 + static_cast<int>(left);
    const int bn = tile % P.tn;
    const int bmb = tile / P.tn;
    const int bm = bmb % P.tm;
    const int b = bmb / P.tm;

    const int M = P.d.M, N = P.d.N, K = P.d.K;
    const float* __restrict__ A = P.A + static_cast<long long>(b) * P.d.a_bs;
    const float* __restrict__ W = P.W;
    const int lda = P.d.lda, ldw = P.d.ldw;

    float4 ra[4], rb[4];
    auto gload = [&](int kt) {
      const int kk = kt * BK + c4 * 4;
      const bool kin = kk < K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        const int gm = bm * BM + row;
        const int gn = bn * BN + row;
        ra[i] = (kin && gm < M) ? *reinterpret_cast<const float4*>(A + static_cast<long long>(gm) * lda + kk)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[i] = (kin && gn < N) ? *reinterpret_cast<const float4*>(W + static_cast<long long>(gn) * ldw + kk)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto sstore = [&](int stage) {
      float* As = smem + stage * STAGE_FLOATS;
      float* Bs = As + BM * LDP;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        *reinterpret_cast<float4*>(As + row * LDP + c4 * 4) = ra[i];
        *reinterpret_cast<float4*>(Bs + row * LDP + c4 * 4) = rb[i];
      }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(k0);
    sstore(0);
    __syncthreads();
    for (int kt = k0; kt < k1; ++kt) {
      const int st = (kt - k0) & 1;
      if (kt + 1 < k1) gload(kt + 1);
      const float* As = smem + st * STAGE_FLOATS;
      const float* Bs = As + BM * LDP;
      const float* a_base = As + (wm * 64 + frag_row) * LDP + frag_k;
      const float* b_base = Bs + (wn * 64 + frag_row) * LDP + frag_k;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const float4 a0 = *reinterpret_cast<const float4*>(a_base + kg * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDP + kg * 8);
        const float4 b0 = *reinterpret_cast<const float4*>(b_base + kg * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(b_base + 32 * LDP + kg * 8);
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][r], bv[j][r], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < k1) sstore(st ^ 1);
      __syncthreads();
    }

    if (k0 == 0 && k1 == P.kt) {
      tile_epilogue(P, b, bm, bn, acc, wm, wn, lane);
    } else {
      // raw accumulators, lane-contiguous: slot[(wave*4 + i*2 + j)*16 + r][lane]
      float* slot = a.ws + (static_cast<long long>(2 * g) + (k0 > 0 ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    u += k1 - k0;
  }
}

// one workgroup per output tile; tiles that one range covered completely were finished above
__global__ __launch_bounds__(256) void gemm_streamk_fixup_kernel(SKArgs a) {
  const long long t = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int k = 1; k < MAXP; ++k)
    if (k < a.np && t >= a.pr[k].tile0) pi = k;
  const DevProblem& P = a.pr[pi];
  const int tile = static_cast<int>(t - P.tile0);
  const long long f = P.unit0 + static_cast<long long>(tile) * P.kt;  // tile's first unit
  const long long l = f + P.kt;
  long long g0 = (f * a.G) / a.U;
  while (g0 + 1 < a.G && range_start(g0 + 1, a.U, a.G) <= f) ++g0;
  while (g0 > 0 && range_start(g0, a.U, a.G) > f) --g0;
  if (range_start(g0 + 1, a.U, a.G) >= l) return;  // a single range holds the whole tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (long long g = g0; g < a.G; ++g) {
    const long long s = range_start(g, a.U, a.G), e = range_start(g + 1, a.U, a.G);
    if (s >= l) break;
    const long long ob = s > f ? s : f, oe = e < l ? e : l;
    if (oe <= ob) continue;
    const float* slot = a.ws + (2 * g + (ob > f ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += slot[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane];
  }
  const int bn = tile % P.tn;
  const int bmb = tile / P.tn;
  tile_epilogue(P, bmb / P.tm, bmb % P.tm, bn, acc, wm, wn, lane);
}

}  // namespace

extern "C" int ldc_sizeof_gemm_problem(void) { return static_cast<int>(sizeof(ldc_gemm_problem)); }

extern "C" long long ldc_gemm_grouped_workspace_bytes(void) {
  // 2 slots per workgroup, 2 workgroups per CU on a 256-CU MI355X
  return 2LL * 512 * SLOT_FLOATS * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_gemm_grouped(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                                void* stream) {
  LDC_CHECK_PTR(problems);
  if (n <= 0 || n > MAXP) return LDC_ERR_ARG;
  SKArgs a{};
  a.np = n;
  long long U = 0, tiles = 0;
  for (int i = 0; i < n; ++i) {
    const ldc_gemm_problem& q = problems[i];
    LDC_CHECK_PTR(q.A);
    LDC_CHECK_PTR(q.W);
    LDC_CHECK_PTR(q.C);
    const ldc_gemm_desc& d = q.d;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.batch <= 0) return LDC_ERR_ARG;
    LDC_CHECK_ALIGN16(q.A);
    LDC_CHECK_ALIGN16(q.W);
    if ((d.K & 3) || (d.lda & 3) || (d.ldw & 3) || (d.a_bs & 3)) return LDC_ERR_ALIGN;
    if (d.act < LDC_ACT_NONE || d.act > LDC_ACT_RELU) return LDC_ERR_UNSUPPORTED;
    DevProblem& P = a.pr[i];
    P.A = q.A;
    P.W = q.W;
    P.bias = q.bias;
    P.gate = q.gate;
    P.R = q.R;
    P.C = q.C;
    P.d = d;
    P.tm = ldc_cdiv(d.M, BM);
    P.tn = ldc_cdiv(d.N, BN);
    P.kt = ldc_cdiv(d.K, BK);
    P.unit0 = U;
    P.tile0 = tiles;
    const long long t = static_cast<long long>(d.batch) * P.tm * P.tn;
    tiles += t;
    U += t * P.kt;
  }
  if (tiles > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  const long long slot_bytes = SLOT_FLOATS * static_cast<long long>(sizeof(float));
  long long G = 512;
  if (U < G) G = U;
  if (workspace == nullptr || workspace_bytes < 2 * G * slot_bytes) {
    // not enough scratch for 2 slots per workgroup: shrink the grid to what fits (>= 1 tile-complete fallback)
    const long long fit = workspace ? workspace_bytes / (2 * slot_bytes) : 0;
    if (fit < 1) return LDC_ERR_ARG;
    if (fit < G) G = fit;
  }
  LDC_CHECK_ALIGN16(workspace);
  a.G = static_cast<int>(G);
  a.U = U;
  a.tiles = tiles;
  a.ws = static_cast<float*>(workspace);
  const size_t lds = 2 * STAGE_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_streamk_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    attr_set = true;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_streamk_kernel, dim3(a.G), dim3(256), lds, s, a);
  int st = ldc_launch_status();
  if (st != LDC_OK) return st;
  // some tile is split whenever the ranges are not tile-aligned; the fix-up exits at once for whole tiles
  hipLaunchKernelGGL(gemm_streamk_fixup_kernel, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, s, a);
  return ldc_launch_status();
}
