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
// publishes its raw accumulators as a write-through (sc1) slab in the caller-provided workspace
// (<= 2 slabs per workgroup), drains, and draws ONE relaxed agent-scope ticket on the tile's
// counter; the piece whose ticket is the last re-reads every slab of the tile in workgroup order
// behind an agent-scope acquire and applies the epilogue (round 3: the protocol of
// gemm_bf16x3_v3.hip / cdna_hip_programming.md G16 - nobody waits, the sum order is fixed, so the
// result is bitwise reproducible and equal to what the former second "fix-up" launch produced).
#include <math.h>
#include <stdlib.h>
#include <string.h>

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
  unsigned* counters;  // one per tile (head of the workspace, zeroed by ldc_gemm_grouped_workspace_init, re-armed by the last arriver)
};

__device__ __forceinline__ long long range_start(long long g, long long U, int G) { return (g * U) / G; }

// epilogue (whole tiles and the last-arriving piece of split tiles); acc layout = MFMA C/D layout of the 2x2 wave grid
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

// A piece of a split tile: publish, take a ticket, and - if this piece arrived last - sum all pieces in workgroup order and
// finish the tile.  `flag`: one LDS word that is idle here (the staging ring behind the k loop's last barrier).
__device__ __forceinline__ void publish_or_reduce(const SKArgs& a, const DevProblem& P, int g, int tile, int k0, int b, int bm, int bn,
                                                  f32x16 (&acc)[2][2], int wave, int wm, int wn, int lane, unsigned* flag) {
  float* slot = a.ws + (static_cast<long long>(2 * g) + (k0 > 0 ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __hip_atomic_store(slot + ((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long long f = P.unit0 + static_cast<long long>(tile) * P.kt;  // the tile's first unit
  const long long l = f + P.kt;
  long long g_first = g;
  while (g_first > 0 && range_start(g_first, a.U, a.G) > f) --g_first;
  long long g_last = g;
  while (g_last + 1 < a.G && range_start(g_last + 1, a.U, a.G) < l) ++g_last;
  const unsigned pieces = static_cast<unsigned>(g_last - g_first + 1);
  unsigned* cnt = a.counters + (P.tile0 + tile);
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (ticket == pieces - 1) ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next call
    }
    *flag = last;
  }
  __syncthreads();
  const unsigned is_last = *flag;
  __syncthreads();  // the flag word is staging memory: everyone has read it before the next segment stages into it
  if (!is_last) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (long long gp = g_first; gp <= g_last; ++gp) {
    const long long s = range_start(gp, a.U, a.G);
    const float* sl = a.ws + (2 * gp + (s > f ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += sl[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane];
  }
  tile_epilogue(P, b, bm, bn, acc, wm, wn, lane);
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
      publish_or_reduce(a, P, g, tile, k0, b, bm, bn, acc, wave, wm, wn, lane, reinterpret_cast<unsigned*>(smem));
    }
    u += k1 - k0;
  }
}

// ---------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant: the same contraction on the 16x faster bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16) with fp32-grade error compensation.  Each fp32 operand is written as
// hi + lo with hi = bf16(x), lo = bf16(x - hi) (24 -> 16 significant bits kept), and
//     A.W^T  ~=  Ah.Wh^T + Ah.Wl^T + Al.Wh^T          (fp32 accumulation inside the MFMA)
// dropping only the lo*lo term (~2^-18 relative per product).  Measured on this model: 4e-6 rel-L2
// per forward, 2e-6 over a 20-step Heun chunk vs exact fp32 (tools/split_precision_study.py),
// against a 1e-4 budget; a plain bf16 GEMM is 2e-3.
// Weights are pre-split once (ldc_pack_weight_bf16x2: [N][K/8][hi x8 | lo x8], 4 bytes/element like
// fp32); activations are split in the loader on their way to LDS (v_cvt_pk_bf16_f32).
// LDS row = [hi 64 B | lo 64 B | pad 16 B] (pitch 36 dwords: conflict-free b128 reads and writes).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PITCH_B = 144;                     // bytes per LDS row
constexpr int STAGE_BYTES_B = (BM + BN) * PITCH_B;  // 36 KiB per stage

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// (a, b) -> one dword of two bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) + the fp32 residuals
__device__ __forceinline__ unsigned split_pair(float a, float b, float& ra, float& rb) {
  bf16x2 v;
  v.x = static_cast<__bf16>(a);
  v.y = static_cast<__bf16>(b);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  ra = a - __uint_as_float(u << 16);
  rb = b - __uint_as_float(u & 0xffff0000u);
  return u;
}
__device__ __forceinline__ unsigned pack_pair(float a, float b) {
  bf16x2 v;
  v.x = static_cast<__bf16>(a);
  v.y = static_cast<__bf16>(b);
  return __builtin_bit_cast(unsigned, v);
}
// 8 fp32 -> hi (8 bf16 = 16 B) and lo = bf16(x - hi) (16 B)
__device__ __forceinline__ void split8(const float4 p, const float4 q, uint4& h, uint4& l) {
  float r0, r1, r2, r3, r4, r5, r6, r7;
  h.x = split_pair(p.x, p.y, r0, r1);
  h.y = split_pair(p.z, p.w, r2, r3);
  h.z = split_pair(q.x, q.y, r4, r5);
  h.w = split_pair(q.z, q.w, r6, r7);
  l.x = pack_pair(r0, r1);
  l.y = pack_pair(r2, r3);
  l.z = pack_pair(r4, r5);
  l.w = pack_pair(r6, r7);
}

__global__ __launch_bounds__(256, 2) void gemm_streamk_bf16x3_kernel(SKArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.x;
  const long long u_begin = range_start(g, a.U, a.G);
  const long long u_end = range_start(g + 1, a.U, a.G);
  const int frag_row = lane & 31;
  const int frag_h = lane >> 5;

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
    const uint4* __restrict__ Wp = reinterpret_cast<const uint4*>(P.W);  // [N][K/8][2] x 16 B
    const int lda = P.d.lda;
    const int kchunks = K >> 3;

    // staging: chunk c = tid + 256*i (i = 0,1): row = c>>2, 8-wide k chunk kc = c&3.
    // Out-of-range chunks load from a valid clamped address and are zeroed by a select (no branches,
    // so the staging registers stay in VGPRs).
    const int srow0 = tid >> 2, srow1 = srow0 + 64, skc = tid & 3;
    const bool am0 = bm * BM + srow0 < M, am1 = bm * BM + srow1 < M;
    const bool wn0 = bn * BN + srow0 < N, wn1 = bn * BN + srow1 < N;
    const float* a_src0 = A + static_cast<long long>(am0 ? bm * BM + srow0 : 0) * lda + skc * 8;
    const float* a_src1 = A + static_cast<long long>(am1 ? bm * BM + srow1 : 0) * lda + skc * 8;
    const uint4* w_src0 = Wp + (static_cast<long long>(wn0 ? bn * BN + srow0 : 0) * kchunks + skc) * 2;
    const uint4* w_src1 = Wp + (static_cast<long long>(wn1 ? bn * BN + srow1 : 0) * kchunks + skc) * 2;
    float4 ra00, ra01, ra10, ra11;
    uint4 rw00, rw01, rw10, rw11;
    auto gload = [&](int kt) {
      const bool kin = kt * BK + skc * 8 < K;
      const int ko = kin ? kt * BK : 0;  // element offset along K (clamped when the chunk is past K)
      const float4* pa0 = reinterpret_cast<const float4*>(a_src0 + ko);
      const float4* pa1 = reinterpret_cast<const float4*>(a_src1 + ko);
      const uint4* pw0 = w_src0 + (ko >> 3) * 2;
      const uint4* pw1 = w_src1 + (ko >> 3) * 2;
      ra00 = pa0[0];
      ra01 = pa0[1];
      ra10 = pa1[0];
      ra11 = pa1[1];
      rw00 = pw0[0];
      rw01 = pw0[1];
      rw10 = pw1[0];
      rw11 = pw1[1];
      // zero out-of-range chunks with component-wise masks (a whole-vector select is lowered through scratch)
      const unsigned m0a = (kin && am0) ? 0xffffffffu : 0u, m1a = (kin && am1) ? 0xffffffffu : 0u;
      const unsigned m0w = (kin && wn0) ? 0xffffffffu : 0u, m1w = (kin && wn1) ? 0xffffffffu : 0u;
#define LDC_MASKF4(v, m)                                 \
  v.x = __uint_as_float(__float_as_uint(v.x) & (m));     \
  v.y = __uint_as_float(__float_as_uint(v.y) & (m));     \
  v.z = __uint_as_float(__float_as_uint(v.z) & (m));     \
  v.w = __uint_as_float(__float_as_uint(v.w) & (m));
#define LDC_MASKU4(v, m) \
  v.x &= (m);            \
  v.y &= (m);            \
  v.z &= (m);            \
  v.w &= (m);
      LDC_MASKF4(ra00, m0a) LDC_MASKF4(ra01, m0a) LDC_MASKF4(ra10, m1a) LDC_MASKF4(ra11, m1a)
      LDC_MASKU4(rw00, m0w) LDC_MASKU4(rw01, m0w) LDC_MASKU4(rw10, m1w) LDC_MASKU4(rw11, m1w)
    };
    auto sstore = [&](int stage) {
      unsigned char* As = smem_b + stage * STAGE_BYTES_B;
      unsigned char* Bs = As + BM * PITCH_B;
      uint4 h, l;
      split8(ra00, ra01, h, l);
      *reinterpret_cast<uint4*>(As + srow0 * PITCH_B + skc * 16) = h;
      *reinterpret_cast<uint4*>(As + srow0 * PITCH_B + 64 + skc * 16) = l;
      split8(ra10, ra11, h, l);
      *reinterpret_cast<uint4*>(As + srow1 * PITCH_B + skc * 16) = h;
      *reinterpret_cast<uint4*>(As + srow1 * PITCH_B + 64 + skc * 16) = l;
      *reinterpret_cast<uint4*>(Bs + srow0 * PITCH_B + skc * 16) = rw00;
      *reinterpret_cast<uint4*>(Bs + srow0 * PITCH_B + 64 + skc * 16) = rw01;
      *reinterpret_cast<uint4*>(Bs + srow1 * PITCH_B + skc * 16) = rw10;
      *reinterpret_cast<uint4*>(Bs + srow1 * PITCH_B + 64 + skc * 16) = rw11;
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
      const unsigned char* As = smem_b + st * STAGE_BYTES_B;
      const unsigned char* Bs = As + BM * PITCH_B;
      const unsigned char* a_base = As + (wm * 64 + frag_row) * PITCH_B + frag_h * 16;
      const unsigned char* b_base = Bs + (wn * 64 + frag_row) * PITCH_B + frag_h * 16;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {  // two 16-deep MFMA steps per 32-deep k-tile
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(a_base + s2 * 32);
        const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(a_base + 64 + s2 * 32);
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(a_base + 32 * PITCH_B + s2 * 32);
        const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(a_base + 32 * PITCH_B + 64 + s2 * 32);
        const bf16x8 wh0 = *reinterpret_cast<const bf16x8*>(b_base + s2 * 32);
        const bf16x8 wl0 = *reinterpret_cast<const bf16x8*>(b_base + 64 + s2 * 32);
        const bf16x8 wh1 = *reinterpret_cast<const bf16x8*>(b_base + 32 * PITCH_B + s2 * 32);
        const bf16x8 wl1 = *reinterpret_cast<const bf16x8*>(b_base + 32 * PITCH_B + 64 + s2 * 32);
#define LDC_MFMA3(ACC, AH, AL, WH, WL)                                        \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, WH, ACC, 0, 0, 0);        \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, WL, ACC, 0, 0, 0);        \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, WH, ACC, 0, 0, 0);
        LDC_MFMA3(acc[0][0], ah0, al0, wh0, wl0)
        LDC_MFMA3(acc[0][1], ah0, al0, wh1, wl1)
        LDC_MFMA3(acc[1][0], ah1, al1, wh0, wl0)
        LDC_MFMA3(acc[1][1], ah1, al1, wh1, wl1)
      }
      if (kt + 1 < k1) sstore(st ^ 1);
      __syncthreads();
    }

    if (k0 == 0 && k1 == P.kt) {
      tile_epilogue(P, b, bm, bn, acc, wm, wn, lane);
    } else {
      publish_or_reduce(a, P, g, tile, k0, b, bm, bn, acc, wave, wm, wn, lane, reinterpret_cast<unsigned*>(smem_b));
    }
    u += k1 - k0;
  }
}

// W [N][K] fp32 (row stride ldw) -> packed split [N][K/8][hi x8 | lo x8] bf16
__global__ __launch_bounds__(256) void pack_weight_bf16x2_kernel(const float* __restrict__ W, uint4* __restrict__ out,
                                                                 int N, int K, int ldw) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const int kchunks = K >> 3;
  if (idx >= static_cast<long long>(N) * kchunks) return;
  const int n = static_cast<int>(idx / kchunks), kc = static_cast<int>(idx - static_cast<long long>(n) * kchunks);
  const float4* src = reinterpret_cast<const float4*>(W + static_cast<long long>(n) * ldw + kc * 8);
  uint4 h, l;
  split8(src[0], src[1], h, l);
  out[idx * 2] = h;
  out[idx * 2 + 1] = l;
}

}  // namespace

extern "C" int ldc_sizeof_gemm_problem(void) { return static_cast<int>(sizeof(ldc_gemm_problem)); }

extern "C" long long ldc_gemm_grouped_workspace_bytes(void) {
  // counter block + 2 slabs per workgroup, 2 workgroups per CU on a 256-CU MI355X
  return LDC_GEMM_COUNTER_BYTES + 2LL * 512 * SLOT_FLOATS * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_gemm_grouped_workspace_init(void* workspace, long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(workspace);
  if (workspace_bytes < LDC_GEMM_COUNTER_BYTES) return LDC_ERR_ARG;
  hipError_t e = hipMemsetAsync(workspace, 0, LDC_GEMM_COUNTER_BYTES, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? LDC_OK : -(1000 + static_cast<int>(e));
}

static int gemm_grouped_impl(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                             void* stream, bool split_bf16) {
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
    if ((d.K & 3) || (d.lda & 3) || (d.a_bs & 3)) return LDC_ERR_ALIGN;
    if (split_bf16 ? (d.K & 7) != 0 : (d.ldw & 3) != 0) return LDC_ERR_ALIGN;
    if (d.act < LDC_ACT_NONE || d.act > LDC_ACT_RELU) return LDC_ERR_UNSUPPORTED;
    if ((d.flags & ~LDC_GEMM_F32_REGSTAGE) != 0) return LDC_ERR_UNSUPPORTED;  // split activation formats: the LDS-DMA bf16x3 kernel only (K % 32 == 0)
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
  if (tiles > LDC_GEMM_COUNTER_BYTES / 4 - 16) return LDC_ERR_UNSUPPORTED;  // one hand-off counter per tile (1 MiB block: 262 128 tiles)
  const long long slot_bytes = SLOT_FLOATS * static_cast<long long>(sizeof(float));
  long long G = 512;
  {
    // Balance k-steps per workgroup against slabs per split tile: time ~ (U/G) t_unit + (kt G/U) t_slab is
    // smallest at U/G = sqrt(kt t_slab / t_unit); t_slab / t_unit ~ 2 (fp32 units) or ~ 7 (bf16x3 units).
    const double kt_avg = static_cast<double>(U) / static_cast<double>(tiles);
    long long umin = static_cast<long long>(sqrt(kt_avg * (split_bf16 ? 7.0 : 2.0)) + 0.5);
    if (umin < 1) umin = 1;
    const long long gmax = U / umin > 0 ? U / umin : 1;
    if (gmax < G) G = gmax;
  }
  if (U < G) G = U;
  if (workspace == nullptr || workspace_bytes < LDC_GEMM_COUNTER_BYTES + 2 * slot_bytes) return LDC_ERR_ARG;
  {
    const long long fit = (workspace_bytes - LDC_GEMM_COUNTER_BYTES) / (2 * slot_bytes);
    if (fit < G) G = fit;  // not enough scratch for 2 slabs per workgroup: shrink the grid to what fits
  }
  LDC_CHECK_ALIGN16(workspace);
  a.G = static_cast<int>(G);
  a.U = U;
  a.tiles = tiles;
  a.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES);
  a.counters = static_cast<unsigned*>(workspace);
  const size_t lds = split_bf16 ? 2 * STAGE_BYTES_B : 2 * STAGE_FLOATS * sizeof(float);
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_streamk_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(2 * STAGE_FLOATS * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_streamk_bf16x3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES_B);
    return true;
  }();
  (void)attr_set;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (split_bf16) hipLaunchKernelGGL(gemm_streamk_bf16x3_kernel, dim3(a.G), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(gemm_streamk_kernel, dim3(a.G), dim3(256), lds, s, a);
  return ldc_launch_status();
}

int ldc_gemm_grouped_f32_ring(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                              void* stream);  // gemm_bf16x3_v3.hip, TERMS = 0

extern "C" int ldc_gemm_grouped(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                                void* stream) {
  // round 3: the LDS-DMA ring kernel with fp32 operand rows and the fp32 matrix instruction where the shapes allow (K % 32 == 0,
  // contiguous weights, 16-byte rows); the register-staged stream-K kernel below otherwise
  const int st = ldc_gemm_grouped_f32_ring(problems, n, workspace, workspace_bytes, stream);
  if (st != LDC_ERR_UNSUPPORTED) return st;
  return gemm_grouped_impl(problems, n, workspace, workspace_bytes, stream, false);
}

int ldc_gemm_grouped_bf16x3_v3(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                               void* stream);  // gemm_bf16x3_v3.hip

extern "C" int ldc_gemm_grouped_bf16x3(const ldc_gemm_problem* problems, int n, void* workspace,
                                       long long workspace_bytes, void* stream) {
  // Pre-split activations (LDC_GEMM_A_SPLIT) and K % 32 == 0 - every launch of the models: the 16x16x32 ring kernel
  // (gemm_bf16x3_v3.hip); anything else (fp32 activations, K % 32 != 0): the register-staged kernel below, which splits in its loop.
  // A/B build only: LDC_BF16X3_KERNEL=regstage forces the register-staged kernel.
  int force = 0;
#ifdef LDC_AB_BUILD
  static const int force_env = [] {
    const char* e = getenv("LDC_BF16X3_KERNEL");
    return (e != nullptr && strcmp(e, "regstage") == 0) ? 1 : 0;
  }();
  force = force_env;
#endif
  if (force == 0) {
    const int st = ldc_gemm_grouped_bf16x3_v3(problems, n, workspace, workspace_bytes, stream);
    if (st != LDC_ERR_UNSUPPORTED) return st;
  }
  return gemm_grouped_impl(problems, n, workspace, workspace_bytes, stream, true);
}

namespace {
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* __restrict__ W, uint2* __restrict__ out, long long N, int K4, int ldw) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;  // one float4 -> 4 bf16
  if (idx >= N * K4) return;
  const long long n = idx / K4;
  const int c = static_cast<int>(idx - n * K4);
  const float4 v = reinterpret_cast<const float4*>(W + n * ldw)[c];
  out[idx] = make_uint2(ldc_pack_pair(v.x, v.y), ldc_pack_pair(v.z, v.w));
}
}  // namespace

extern "C" int ldc_pack_weight_bf16(const float* W, void* out, int N, int K, int ldw, void* stream) {
  LDC_CHECK_PTR(W);
  LDC_CHECK_PTR(out);
  if (N <= 0 || K <= 0) return LDC_ERR_ARG;
  if ((K & 7) || (ldw & 3)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(W);
  LDC_CHECK_ALIGN16(out);
  const long long total = static_cast<long long>(N) * (K >> 2);
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(ldc_cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), W,
                     static_cast<uint2*>(out), static_cast<long long>(N), K >> 2, ldw);
  return ldc_launch_status();
}

extern "C" int ldc_pack_weight_bf16x2(const float* W, void* out, int N, int K, int ldw, void* stream) {
  LDC_CHECK_PTR(W);
  LDC_CHECK_PTR(out);
  if (N <= 0 || K <= 0) return LDC_ERR_ARG;
  if ((K & 7) || (ldw & 3)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(W);
  LDC_CHECK_ALIGN16(out);
  const long long total = static_cast<long long>(N) * (K >> 3);
  hipLaunchKernelGGL(pack_weight_bf16x2_kernel, dim3(ldc_cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     W, static_cast<uint4*>(out), N, K, ldw);
  return ldc_launch_status();
}
