// Split-bf16 ("bf16x3") stream-K GEMM, second generation: 256x128 tile, 8 waves, operands brought in
// by LDS-DMA (global_load_lds_dwordx4) through a 3-stage ring, one barrier per k-step.
//
// Why a second kernel: the register-staged bf16x3 kernel (gemm_streamk.hip) multiplies a 32-deep k-step
// in ~770 MFMA cycles per wave, which is shorter than one L2/MALL miss; with staging registers for
// only one k-step in flight (two would spill: 64 accumulators + 2x32 staging + fragments > 256 VGPRs)
// it waits on memory every step and stops at ~290 TFLOP/s-equivalent.  Here neither operand touches a
// VGPR on its way to LDS, so the ring can run TWO k-steps ahead of the MFMAs:
//   * A tile: 256 rows x 32 fp32 (raw activations, 128 B per row), W tile: 128 rows x [hi|lo] bf16 x 32
//     (pre-split by ldc_pack_weight_bf16x2, also 128 B per row) -> 48 KiB per stage, 3 stages = 144 KiB,
//     one workgroup of 8 waves (2 per SIMD) per CU;
//   * LDS rows are unpadded (a DMA instruction writes 1 KiB = 8 rows linearly); bank conflicts are
//     avoided by an XOR swizzle applied to the per-lane SOURCE address and to the fragment reads:
//     16-byte slot p of row r holds chunk p ^ ((r >> 1) & 7)  (conflict-free for the b128 lane groups);
//   * the activation split x = hi + lo happens after the LDS read, on the fragment (v_cvt_pk_bf16_f32);
//     waves are arranged 8(M) x 1(N) so every A row is read and split by exactly one wave;
//   * per k-step: s_waitcnt vmcnt(6) (this step's 6 DMA instructions of this wave have landed, the next
//     step's 6 stay in flight) -> raw s_barrier -> issue the DMA of step t+2 into the stage read at
//     step t-1 -> 2 x 12 MFMAs.  No ordinary global load lives in the loop, all LDS is one array
//     (cdna_hip_programming.md section 5, "Pipelining across barriers");
//   * the fragment reads are inline-asm ds_read_b128 with their own lgkmcnt wait: a compiler-visible
//     LDS read makes hipcc (ROCm 7.2) drain vmcnt(0) in front of it while an LDS-DMA is in flight,
//     which would collapse the ring to depth 0 (seen in the .s of the first build of this kernel).
// Scheduling and epilogue are the stream-K scheme of gemm_streamk.hip with BM = 256.  Split tiles are
// reduced IN the launch by whichever piece arrives last ("last arriver reduces", no workgroup ever
// waits, so nothing depends on co-residency or dispatch order): every piece stores its accumulators
// to its slab with write-through (sc1) stores, every storing wave drains them (s_waitcnt vmcnt(0)),
// the workgroup barriers, ONE lane takes a ticket on the tile's counter with a relaxed agent-scope
// atomic add; the piece whose ticket is pieces-1 does one agent-scope acquire, then re-reads ALL slabs
// of the tile in workgroup order (its own too, so the sum order is fixed and the result is bitwise
// reproducible), applies the epilogue and re-arms the counter (cdna_hip_programming.md Guideline 16,
// R1 + the counter form of "In-launch split-K reduction").  A separate fix-up launch remains only as
// the fallback for grids with more tiles than counters.
// Needs K % 32 == 0 (no K-edge zero fill through DMA); other shapes use the register-staged kernel.
#include <math.h>

#include "common.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int ROW_B = 128;                       // bytes per LDS row (both operands)
constexpr int STAGE_B = (BM + BN) * ROW_B;       // 48 KiB
constexpr int NSTAGE = 3;
constexpr int SLOT_FLOATS = BM * BN;
constexpr int MAXP = LDC_GEMM_MAX_PROBLEMS;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct DevProblem {
  const float* A;
  const unsigned char* W;  // packed split weights
  const float* bias;
  const float* gate;
  const float* R;
  float* C;
  ldc_gemm_desc d;
  int tm, tn, kt;
  long long unit0;
  long long tile0;
};

struct SKArgs {
  DevProblem pr[MAXP];
  int np;
  int G;
  long long U;
  long long tiles;
  float* ws;
  unsigned* counters;  // one per tile (nullptr -> fix-up launch instead of in-launch reduction)
};

__device__ __forceinline__ long long range_start(long long g, long long U, int G) { return (g * U) / G; }

__device__ __forceinline__ int find_problem_by_unit(const SKArgs& a, long long u) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < MAXP; ++k)
    if (k < a.np && u >= a.pr[k].unit0) pi = k;
  return pi;
}

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
__device__ __forceinline__ void split8(const float4 p, const float4 q, bf16x8& hi, bf16x8& lo) {
  float r0, r1, r2, r3, r4, r5, r6, r7;
  uint4 h, l;
  h.x = split_pair(p.x, p.y, r0, r1);
  h.y = split_pair(p.z, p.w, r2, r3);
  h.z = split_pair(q.x, q.y, r4, r5);
  h.w = split_pair(q.z, q.w, r6, r7);
  l.x = pack_pair(r0, r1);
  l.y = pack_pair(r2, r3);
  l.z = pack_pair(r4, r5);
  l.w = pack_pair(r6, r7);
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// acc[j][r]: wave w owns rows [32w, 32w+32) of the tile and all 128 columns (4 MFMA column tiles)
__device__ __forceinline__ void tile_epilogue(const DevProblem& P, int b, int bm, int bn, const f32x16 (&acc)[4], int wave,
                                              int lane) {
  const int M = P.d.M, N = P.d.N;
  float* __restrict__ C = P.C + static_cast<long long>(b) * P.d.c_bs;
  const float* __restrict__ R = P.R ? P.R + static_cast<long long>(b) * P.d.r_bs : nullptr;
  const float* __restrict__ gate = P.gate ? P.gate + static_cast<long long>(b) * P.d.gate_bs : nullptr;
  const int act = P.d.act;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = bn * BN + j * 32 + (lane & 31);
    if (n >= N) continue;
    const float bias_n = P.bias ? P.bias[n] : 0.f;
    const float gate_n = gate ? gate[n] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = bm * BM + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= M) continue;
      float v = acc[j][r] + bias_n;
      v = ldc_apply_act(v, act);
      if (gate) v *= gate_n;
      if (R) v += R[static_cast<long long>(m) * P.d.ldr + n];
      C[static_cast<long long>(m) * P.d.ldc + n] = v;
    }
  }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

// untracked LDS reads (section 5.7 form (ii)): every destination is named "+v" in the wait statement below
#define LDC_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<unsigned long long>(p));  // low 32 bits of a flat LDS pointer = LDS offset
}

__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(512) void gemm_bf16x3_dma_kernel(SKArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // CONTIGUOUS run of unit ranges; with the row-tile-fastest tile order below, the ~G/8 workgroups of an XCD
  // then walk the same few W column panels and A row panels in step and hit in that XCD's private L2
  // (bijective for any G, cdna_hip_programming.md T1).  Measured before: 28 % L2 hit rate, ~15x over-fetch.
  int g;
  {
    const int bid = blockIdx.x, G = a.G;
    const int q = G >> 3, r = G & 7, xcd = bid & 7;
    g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const long long u_begin = range_start(g, a.U, a.G);
  const long long u_end = range_start(g + 1, a.U, a.G);
  const int fr = lane & 31;
  const int fh = lane >> 5;
  // DMA lane geometry: a wave instruction covers 8 rows x 128 B; lane -> (row lr, 16-byte slot lp)
  const int lr = lane >> 3, lp = lane & 7;
  const unsigned smem_lds = lds_addr(smem);

  long long u = u_begin;
  while (u < u_end) {
    const int pi = find_problem_by_unit(a, u);
    const DevProblem& P = a.pr[pi];
    const long long local = u - P.unit0;
    const int tile = static_cast<int>(local / P.kt);
    const int k0 = static_cast<int>(local - static_cast<long long>(tile) * P.kt);
    const long long left = u_end - u;
    const int k1 = (P.kt - k0 <= left) ? P.kt : k0 + static_cast<int>(left);
    const int bm = tile % P.tm;  // row tile fastest: consecutive tiles share one W column panel
    const int bnb = tile / P.tm;
    const int bn = bnb % P.tn;
    const int b = bnb / P.tn;
    const int M = P.d.M, N = P.d.N, K = P.d.K;
    const float* __restrict__ A = P.A + static_cast<long long>(b) * P.d.a_bs;
    const int lda = P.d.lda;
    const long long w_row_bytes = static_cast<long long>(K) * 4;  // packed row: K/8 chunks x 32 B

    // per-lane DMA sources (k-step 0); rows past the edge are clamped (their outputs are never stored)
    // A instruction q (0..31) covers tile rows [8q, 8q+8); this wave issues q = wave + 8i, i = 0..3
    // W instruction q (0..15) covers tile rows [8q, 8q+8); this wave issues q = wave + 8i, i = 0..1
    const unsigned char* a_src[4];
    const unsigned char* w_src[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * (wave + 8 * i) + lr;
      int gm = bm * BM + r;
      gm = gm < M ? gm : M - 1;
      const int c = lp ^ ((r >> 1) & 7);
      a_src[i] = reinterpret_cast<const unsigned char*>(A + static_cast<long long>(gm) * lda) + c * 16;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (wave + 8 * i) + lr;
      int gn = bn * BN + r;
      gn = gn < N ? gn : N - 1;
      const int c = lp ^ ((r >> 1) & 7);
      w_src[i] = P.W + static_cast<long long>(gn) * w_row_bytes + c * 16;
    }
    // one DMA instruction i (0..3: A rows, 4..5: W rows) of k-step kt into `stage`
    auto issue_one = [&](int kt, int stage, int i) {
      unsigned char* sA = smem + stage * STAGE_B;
      unsigned char* sW = sA + BM * ROW_B;
      const long long koff = static_cast<long long>(kt) * (BK * 4);  // 128 B per k-step in both operands
      if (i < 4) dma16(a_src[i] + koff, sA + (wave + 8 * i) * 1024);
      else dma16(w_src[i - 4] + koff, sW + (wave + 8 * (i - 4)) * 1024);
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
      for (int i = 0; i < 6; ++i) issue_one(kt, stage, i);
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // fragment read offsets (bytes) inside a stage: row * 128 + ((chunk ^ swz(row)) * 16)
    const int a_row = wave * 32 + fr;
    const int a_swz = (a_row >> 1) & 7;
    const int w_swz = (fr >> 1) & 7;  // rows 32j + fr: (row >> 1) & 7 == (fr >> 1) & 7 ... + 16j & 7 = same

    issue(k0, 0);
    if (k0 + 1 < k1) issue(k0 + 1, 1);
    for (int kt = k0; kt < k1; ++kt) {
      const int st = (kt - k0) % NSTAGE;
      if (kt + 1 < k1) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      // the 6 DMA instructions of k-step kt+2 are spread between the MFMA groups below instead of issued as a
      // burst behind the barrier (where both waves of a SIMD would stall the matrix pipe together)
      const bool pre = kt + 2 < k1;
      const int pst = (kt + 2 - k0) % NSTAGE;
      const unsigned aaddr = smem_lds + st * STAGE_B + a_row * ROW_B;
      const unsigned waddr = smem_lds + st * STAGE_B + BM * ROW_B + fr * ROW_B;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int m = 2 * s2 + fh;  // 8-wide k chunk of this lane half
        const unsigned a0 = aaddr + (((2 * m) ^ a_swz) << 4), a1 = aaddr + (((2 * m + 1) ^ a_swz) << 4);
        const unsigned w0 = waddr + (((2 * m) ^ w_swz) << 4), w1 = waddr + (((2 * m + 1) ^ w_swz) << 4);
        f32x4v p, q;
        i32x4v h0, l0, h1, l1, h2, l2, h3, l3;
        LDC_DS_READ(p, a0, 0);
        LDC_DS_READ(q, a1, 0);
        LDC_DS_READ(h0, w0, 0);
        LDC_DS_READ(l0, w1, 0);
        LDC_DS_READ(h1, w0, 4096);
        LDC_DS_READ(l1, w1, 4096);
        LDC_DS_READ(h2, w0, 8192);
        LDC_DS_READ(l2, w1, 8192);
        LDC_DS_READ(h3, w0, 12288);
        LDC_DS_READ(l3, w1, 12288);
        // counted waits (LDS returns in order): start multiplying as soon as the A fragment and the first W pair are
        // back; the later W pairs land under the earlier MFMAs
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(p), "+v"(q), "+v"(h0), "+v"(l0));
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 ah, al;
        split8(make_float4(p.x, p.y, p.z, p.w), make_float4(q.x, q.y, q.z, q.w), ah, al);
#define LDC_MFMA3(ACC, WH, WL)                                                                              \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(bf16x8, WH), ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, WL), ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, WH), ACC, 0, 0, 0);
        __builtin_amdgcn_s_setprio(1);
        LDC_MFMA3(acc[0], h0, l0)
        if (pre) issue_one(kt + 2, pst, 3 * s2);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(h1), "+v"(l1));
        __builtin_amdgcn_sched_barrier(0);
        LDC_MFMA3(acc[1], h1, l1)
        if (pre) issue_one(kt + 2, pst, 3 * s2 + 1);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(h2), "+v"(l2));
        __builtin_amdgcn_sched_barrier(0);
        LDC_MFMA3(acc[2], h2, l2)
        if (pre) issue_one(kt + 2, pst, 3 * s2 + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h3), "+v"(l3));
        __builtin_amdgcn_sched_barrier(0);
        LDC_MFMA3(acc[3], h3, l3)
        __builtin_amdgcn_s_setprio(0);
      }
    }
    // all waves must be done reading the ring before the next segment's prologue overwrites stage 0/1
    __builtin_amdgcn_s_barrier();

    if (k0 == 0 && k1 == P.kt) {
      tile_epilogue(P, b, bm, bn, acc, wave, lane);
    } else {
      float* slot = a.ws + (static_cast<long long>(2 * g) + (k0 > 0 ? 0 : 1)) * SLOT_FLOATS;
      if (a.counters == nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[((wave * 4 + j) * 16 + r) * 64 + lane] = acc[j][r];
      } else {
        // ---- publish this piece (write-through slab, drained, ONE ticket per workgroup) ----
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(slot + ((wave * 4 + j) * 16 + r) * 64 + lane, acc[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // pieces of this tile = workgroups whose range meets [f, l)
        const long long f = P.unit0 + static_cast<long long>(tile) * P.kt;
        const long long l = f + P.kt;
        long long g_first = g;
        while (g_first > 0 && range_start(g_first, a.U, a.G) > f) --g_first;
        long long g_last = g;
        while (g_last + 1 < a.G && range_start(g_last + 1, a.U, a.G) < l) ++g_last;
        const unsigned pieces = static_cast<unsigned>(g_last - g_first + 1);
        unsigned* cnt = a.counters + (P.tile0 + tile);
        unsigned* flag = reinterpret_cast<unsigned*>(smem);  // the ring is idle here (barrier above)
        if (tid == 0) {
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
        __syncthreads();  // flag word is ring memory: everyone has read it before the next prologue's DMA lands
        if (is_last) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
          for (long long gp = g_first; gp <= g_last; ++gp) {
            const long long s = range_start(gp, a.U, a.G);
            const float* sl = a.ws + (2 * gp + (s > f ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[j][r] += sl[((wave * 4 + j) * 16 + r) * 64 + lane];
              if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
          }
          tile_epilogue(P, b, bm, bn, acc, wave, lane);
        }
      }
    }
    u += k1 - k0;
  }
}

__global__ __launch_bounds__(512) void gemm_bf16x3_dma_fixup_kernel(SKArgs a) {
  const long long t = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int k = 1; k < MAXP; ++k)
    if (k < a.np && t >= a.pr[k].tile0) pi = k;
  const DevProblem& P = a.pr[pi];
  const int tile = static_cast<int>(t - P.tile0);
  const long long f = P.unit0 + static_cast<long long>(tile) * P.kt;
  const long long l = f + P.kt;
  long long g0 = (f * a.G) / a.U;
  while (g0 + 1 < a.G && range_start(g0 + 1, a.U, a.G) <= f) ++g0;
  while (g0 > 0 && range_start(g0, a.U, a.G) > f) --g0;
  if (range_start(g0 + 1, a.U, a.G) >= l) return;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (long long g = g0; g < a.G; ++g) {
    const long long s = range_start(g, a.U, a.G), e = range_start(g + 1, a.U, a.G);
    if (s >= l) break;
    const long long ob = s > f ? s : f, oe = e < l ? e : l;
    if (oe <= ob) continue;
    const float* slot = a.ws + (2 * g + (ob > f ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += slot[((wave * 4 + j) * 16 + r) * 64 + lane];
  }
  const int bm = tile % P.tm;
  const int bnb = tile / P.tm;
  tile_epilogue(P, bnb / P.tn, bm, bnb % P.tn, acc, wave, lane);
}

}  // namespace

// returns LDC_ERR_UNSUPPORTED when a problem does not fit this kernel (caller falls back)
int ldc_gemm_grouped_bf16x3_dma(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
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
    if (d.K % BK) return LDC_ERR_UNSUPPORTED;
    LDC_CHECK_ALIGN16(q.A);
    LDC_CHECK_ALIGN16(q.W);
    if ((d.lda & 3) || (d.a_bs & 3)) return LDC_ERR_ALIGN;
    if (d.act < LDC_ACT_NONE || d.act > LDC_ACT_RELU) return LDC_ERR_UNSUPPORTED;
    DevProblem& P = a.pr[i];
    P.A = q.A;
    P.W = reinterpret_cast<const unsigned char*>(q.W);
    P.bias = q.bias;
    P.gate = q.gate;
    P.R = q.R;
    P.C = q.C;
    P.d = d;
    P.tm = ldc_cdiv(d.M, BM);
    P.tn = ldc_cdiv(d.N, BN);
    P.kt = d.K / BK;
    P.unit0 = U;
    P.tile0 = tiles;
    const long long t = static_cast<long long>(d.batch) * P.tm * P.tn;
    tiles += t;
    U += t * P.kt;
  }
  if (tiles > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  const long long slot_bytes = SLOT_FLOATS * static_cast<long long>(sizeof(float));
  // Grid size: at most one 8-wave workgroup per CU, and -- when every problem has the same k-depth -- a divisor of
  // tiles * s for a small split factor s, so that every range starts on a (1/s)-tile boundary: all workgroups then
  // walk K in phase and the panels shared inside an XCD are fetched once (otherwise plain equal ranges).
  long long G = 256;
  {
    bool same_kt = true;
    for (int i = 1; i < n; ++i) same_kt = same_kt && (a.pr[i].kt == a.pr[0].kt);
    long long best = 0;
    if (same_kt) {
      const int kt = a.pr[0].kt;
      for (int sfac = 1; sfac <= 4; ++sfac) {
        if (kt % sfac || kt / sfac < 8) continue;
        const long long items = tiles * sfac;
        long long gd = items < 256 ? items : 256;
        while (gd > 1 && items % gd) --gd;
        const long long score = gd * (34 - sfac);  // mild preference for fewer splits
        if (score > best * (34 - 1) / 33 && gd > best) best = gd;
      }
    }
    if (best >= 160) {
      G = best;
    } else {
      // units per workgroup >= sqrt(kt * t_slab / t_unit), t_slab / t_unit ~ 8 for this tile
      const double kt_avg = static_cast<double>(U) / static_cast<double>(tiles);
      long long umin = static_cast<long long>(sqrt(kt_avg * 8.0) + 0.5);
      if (umin < 1) umin = 1;
      const long long gmax = U / umin > 0 ? U / umin : 1;
      if (gmax < G) G = gmax;
    }
  }
  if (U < G) G = U;
  if (workspace == nullptr) return LDC_ERR_ARG;
  // workspace = [counter block: LDC_GEMM_COUNTER_BYTES, zeroed once by ldc_gemm_grouped_workspace_init][slabs]
  if (workspace_bytes < LDC_GEMM_COUNTER_BYTES + 2 * slot_bytes) return LDC_ERR_ARG;
  {
    const long long fit = (workspace_bytes - LDC_GEMM_COUNTER_BYTES) / (2 * slot_bytes);
    if (fit < G) G = fit;
  }
  LDC_CHECK_ALIGN16(workspace);
  a.G = static_cast<int>(G);
  a.U = U;
  a.tiles = tiles;
  a.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES);
  a.counters = (tiles <= LDC_GEMM_COUNTER_BYTES / 4) ? static_cast<unsigned*>(workspace) : nullptr;
  const size_t lds = NSTAGE * STAGE_B;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_dma_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    attr_set = true;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_bf16x3_dma_kernel, dim3(a.G), dim3(512), lds, s, a);
  int st = ldc_launch_status();
  if (st != LDC_OK || a.counters != nullptr) return st;
  hipLaunchKernelGGL(gemm_bf16x3_dma_fixup_kernel, dim3(static_cast<unsigned>(tiles)), dim3(512), 0, s, a);
  return ldc_launch_status();
}
