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
#include <stdlib.h>

#include "common.h"

namespace {

// Two instantiations: BM = 256 (8 waves, two per SIMD) for grids that fill the chip with whole tiles, and
// BM = 128 (4 waves, one per SIMD: the rolling pipeline below keeps LDS and VALU work under the MFMAs of the
// SAME wave) for small problems, where 256-row tiles leave CUs idle or force every tile to be split along K.
constexpr int BN = 128, BK = 32;
constexpr int ROW_B = 128;                       // bytes per LDS row (both operands)
constexpr int NSTAGE = 3;
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
  int vec4;     // epilogue may use 16-byte accesses
  int c_split;  // LDC_GEMM_C_SPLIT
  // implicit-GEMM SphereConv2d (CONV instantiation): A is an NHWC image [B][cH][cW][lda], M = B*cH*cW; the reduction
  // runs over ks*ks taps x (cin rounded up to 32 << kshift ... i.e. 2^kshift k-steps per tap); W is packed with the
  // same per-tap padding (zero weights behind cin)
  int cH, cW, cin, ks, kshift;
  const unsigned char* zero16;  // 16 zero bytes (for the lanes of a tap's last k-step that lie behind cin)
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

// acc[j]: wave w owns rows [32w, 32w+32) of the tile and all 128 columns (4 MFMA column tiles j).  The MFMAs are
// issued as W-fragment x A-fragment, so a lane holds ONE output row (m = 32w + lane % 32) and its registers run along
// the columns: register r of tile j is column 32 j + 8 (r >> 2) + 4 (lane / 32) + (r & 3).  Four consecutive registers
// are 16 contiguous bytes of C: the epilogue is 16 dwordx4 stores per lane instead of 64 dword stores (the store
// tail of a tile is issue-bound, MI355X_MICROARCH.md "attention epilogue store tail").
template <int BM>
__device__ __forceinline__ void tile_epilogue(const DevProblem& P, int b, int bm, int bn, const f32x16 (&acc)[4], int wave,
                                              int lane) {
  const int M = P.d.M, N = P.d.N;
  float* __restrict__ C = P.C + static_cast<long long>(b) * P.d.c_bs;
  const float* __restrict__ R = P.R ? P.R + static_cast<long long>(b) * P.d.r_bs : nullptr;
  const float* __restrict__ gate = P.gate ? P.gate + static_cast<long long>(b) * P.d.gate_bs : nullptr;
  const int act = P.d.act;
  const int m = bm * BM + wave * 32 + (lane & 31);
  if (m >= M) return;
  float* crow = C + static_cast<long long>(m) * P.d.ldc;
  const float* rrow = R ? R + static_cast<long long>(m) * P.d.ldr : nullptr;
  const int n_lane = bn * BN + 4 * (lane >> 5);
  if (P.vec4) {  // N % 4 == 0 and every row / vector 16-byte aligned (checked on the host)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n_lane + 32 * j + 8 * g;
        if (n >= N) continue;
        float4 v = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
        if (P.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(P.bias + n);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        v.x = ldc_apply_act(v.x, act); v.y = ldc_apply_act(v.y, act); v.z = ldc_apply_act(v.z, act); v.w = ldc_apply_act(v.w, act);
        if (gate) {
          const float4 gv = *reinterpret_cast<const float4*>(gate + n);
          v.x *= gv.x; v.y *= gv.y; v.z *= gv.z; v.w *= gv.w;
        }
        if (rrow) {
          const float4 rv = *reinterpret_cast<const float4*>(rrow + n);
          v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (P.c_split) {
          // LDC_GEMM_C_SPLIT: columns 8c..8c+7 of a row live in 32 bytes [hi x8 | lo x8]; this lane has half of a group
          float r0, r1, r2, r3;
          uint2 hi, lo;
          hi.x = split_pair(v.x, v.y, r0, r1);
          hi.y = split_pair(v.z, v.w, r2, r3);
          lo.x = pack_pair(r0, r1);
          lo.y = pack_pair(r2, r3);
          unsigned char* grp = reinterpret_cast<unsigned char*>(crow + (n & ~7)) + 2 * (n & 4);
          *reinterpret_cast<uint2*>(grp) = hi;
          *reinterpret_cast<uint2*>(grp + 16) = lo;
        } else {
          *reinterpret_cast<float4*>(crow + n) = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_lane + 32 * j + 8 * (r >> 2) + (r & 3);
        if (n >= N) continue;
        float v = acc[j][r] + (P.bias ? P.bias[n] : 0.f);
        v = ldc_apply_act(v, act);
        if (gate) v *= gate[n];
        if (rrow) v += rrow[n];
        crow[n] = v;
      }
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

// Diagnostic build only (-DLDC_GEMM_STAMPS): wall-clock stamps (100 MHz) per workgroup into the upper half of the
// counter block, which nothing else reads; the product build contains no stamp.
#ifdef LDC_GEMM_STAMPS
#define LDC_STAMP(i)                                                                                        \
  if (threadIdx.x == 0 && a.counters)                                                                       \
    reinterpret_cast<unsigned long long*>(a.counters)[65536 + blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#define LDC_STAMP_CLK(i)                                                                                    \
  if (threadIdx.x == 0 && a.counters)                                                                       \
    reinterpret_cast<unsigned long long*>(a.counters)[65536 + blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define LDC_STAMP(i)
#define LDC_STAMP_CLK(i)
#endif

// Source pixel of tap (ky, kx) for output pixel (h, w) under SphereConv2d's padding and pole rule
// (models/sphere_conv.py:62-129,174-192; same index arithmetic as gemm_f32.hip): rows beyond a pole are the first /
// last p rows mirrored and rolled by W/2, columns wrap, and output row 0 (H-1) sees the first (last) p kernel rows
// flipped horizontally = the mirrored column with the unflipped weight.
__device__ __forceinline__ int sphere_src_pixel(int h, int w, int ky, int kx, int H, int W, int ks) {
  const int p = ks >> 1;
  if ((h == 0 && ky < p) || (h == H - 1 && ky >= ks - p)) kx = ks - 1 - kx;
  int r = h + ky - p;
  int c = w + kx - p;
  if (r < 0) {
    r = -1 - r;
    c -= W >> 1;
  } else if (r >= H) {
    r = 2 * H - 1 - r;
    c -= W >> 1;
  }
  c %= W;
  if (c < 0) c += W;
  return r * W + c;
}

template <int BM, bool APK, bool CONV = false>
__global__ __launch_bounds__(BM * 2) void gemm_bf16x3_dma_kernel(SKArgs a) {
  constexpr int NW = BM / 32;                  // waves per workgroup, 8(M) x 1(N) or 4(M) x 1(N)
  constexpr int STAGE_B = (BM + BN) * ROW_B;   // 48 KiB / 32 KiB
  constexpr int SLOT_FLOATS = BM * BN;
  constexpr int NWI = 16 / NW;                 // W DMA instructions per wave and k-step (A: always 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  LDC_STAMP(0)
  int seg_ = 0;
  (void)seg_;
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
    // A instruction q (0..BM/8-1) covers tile rows [8q, 8q+8); this wave issues q = wave + NW i, i = 0..3
    // W instruction q (0..15) covers tile rows [8q, 8q+8); this wave issues q = wave + NW i, i = 0..NWI-1
    const unsigned char* a_src[4];
    const unsigned char* w_src[NWI];
    int pix_b[4], pix_h[4], pix_w[4], a_chan[4];  // CONV: image base pixel, (h, w) and first channel (4 c) of this lane's chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * (wave + NW * i) + lr;
#ifdef LDC_GEMM_DIAG_SAMEPANEL  // diagnostic build: every workgroup streams the same panels (operands stay in L2)
      int gm = r;
#else
      int gm = bm * BM + r;
#endif
      gm = gm < M ? gm : M - 1;
      const int c = lp ^ ((r >> 1) & 7);
      if constexpr (CONV) {
        const int hw = P.cH * P.cW;
        const int bimg = gm / hw;
        const int rem = gm - bimg * hw;
        pix_b[i] = bimg * hw;
        pix_h[i] = rem / P.cW;
        pix_w[i] = rem - pix_h[i] * P.cW;
        a_chan[i] = 4 * c;
        a_src[i] = nullptr;  // per tap, below
      } else {
        a_src[i] = reinterpret_cast<const unsigned char*>(A + static_cast<long long>(gm) * lda) + c * 16;
      }
    }
    // CONV: source of this lane's chunk for tap `tap`: pixel under the sphere padding rule, channel a_chan
    auto conv_tap_base = [&](int i, int tap) {
      const int ky = tap / P.ks, kx = tap - ky * P.ks;
      const int src = pix_b[i] + sphere_src_pixel(pix_h[i], pix_w[i], ky, kx, P.cH, P.cW, P.ks);
      return reinterpret_cast<const unsigned char*>(A + static_cast<long long>(src) * lda + a_chan[i]);
    };
    if constexpr (CONV) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a_src[i] = conv_tap_base(i, k0 >> P.kshift);
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int r = 8 * (wave + NW * i) + lr;
#ifdef LDC_GEMM_DIAG_SAMEPANEL
      int gn = r;
#else
      int gn = bn * BN + r;
#endif
      gn = gn < N ? gn : N - 1;
      const int c = lp ^ ((r >> 1) & 7);
      w_src[i] = P.W + static_cast<long long>(gn) * w_row_bytes + c * 16;
    }
    // one DMA instruction i (0..3: A rows, 4..ND-1: W rows) of k-step kt into `stage`
    constexpr int ND = 4 + NWI;    // DMA instructions per wave and k-step (6 / 8)
    constexpr int NDH = ND / 2;    // issued per half-step
    auto issue_one = [&](int kt, int stage, int i) {
      unsigned char* sA = smem + stage * STAGE_B;
      unsigned char* sW = sA + BM * ROW_B;
      const long long koff = static_cast<long long>(kt) * (BK * 4);  // 128 B per k-step in both operands
#ifdef LDC_GEMM_DIAG_NODMA  // diagnostic build: no operand traffic at all (results are garbage)
      if (kt >= 0) return;
#endif
      if (i < 4) {
        if constexpr (CONV) {
          // 2^kshift k-steps per tap: k-step kt = (tap, 32-channel chunk).  Piece i is issued once per k-step in
          // increasing kt, so its per-tap base is recomputed exactly when a tap starts (wave-uniform branch).
          const int chunk = kt & ((1 << P.kshift) - 1);
          if (chunk == 0) a_src[i] = conv_tap_base(i, kt >> P.kshift);
          const unsigned char* src = a_src[i] + chunk * (BK * 4);
          if (chunk * BK + BK > P.cin) {  // the tap's last k-step: lanes behind cin read zeros (their weights are zero too)
            if (chunk * BK + a_chan[i] >= P.cin) src = P.zero16;
          }
          dma16(src, sA + (wave + NW * i) * 1024);
        } else {
          dma16(a_src[i] + koff, sA + (wave + NW * i) * 1024);
        }
      } else {
        dma16(w_src[i - 4] + koff, sW + (wave + NW * (i - 4)) * 1024);
      }
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
      for (int i = 0; i < ND; ++i) issue_one(kt, stage, i);
    };
    // the part of half `hf` (0 / 1) of a k-step's DMAs that goes behind MFMA group `slot` (0..2) of a half-step
    auto issue_part = [&](int kt, int stage, int hf, int slot) {
      const int base = hf * NDH;
      if constexpr (NDH == 3) {
        issue_one(kt, stage, base + slot);
      } else {
        if (slot == 0) {
          issue_one(kt, stage, base);
          issue_one(kt, stage, base + 1);
        } else {
          issue_one(kt, stage, base + slot + 1);
        }
      }
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- main loop: rolling software pipeline over HALF k-steps H = (kt, s2), hand-placed instruction stream ----
    // Facts this is built on (MI355X_MICROARCH.md cycle constants; in-kernel clock stamps of the earlier versions):
    //  * a v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles but the SIMD's vector issue for only 8, so
    //    ~24 cycles of other vector work (4-5 VALU, or 1-2 ds_read_b128) hide behind EVERY MFMA -- and only there:
    //    the earlier "reads, wait, split, 12 MFMAs" half-step ran 2390 cycles per k-step against 1536 of MFMA, with
    //    or without operand traffic (the no-DMA diagnostic build measured the same), because the fillers sat in
    //    bursts between MFMA groups where both waves of a SIMD issued no MFMA;
    //  * LDS reads return in order, so waits are counted: reads issued after the awaited ones stay in flight.
    // Every fragment register is re-loaded for half-step H+1 right behind the last MFMA of H that reads it, the
    // activation split of H+1 (24 VALU) is cut into 8 pieces of 3 placed one or two per MFMA gap, and the DMA
    // instructions of later k-steps sit alone in a gap each:
    //   M1 [pq'] M2 [l0'] M3 [h0' DMA] | M4 [s] M5 [l1' s] M6 [h1' DMA] | M7 [s s] M8 [l2' s] M9 [h2' DMA] |
    //   M10 [s s] M11 [l3' s] M12 [h3']            (x' = LDS read of x for H+1, s = one split piece)
    // H = (kt,1) starts with barrier(kt+1) = {lgkmcnt(0): this wave is done READING stage kt; vmcnt: its DMAs of
    // k-step kt+1 landed; s_barrier}: behind it stage kt+1 may be read and stage kt may be overwritten (by the DMAs
    // of k-step kt+3).  DMA schedule: k-step t's instructions are issued half in H = (t-3,1), half in H = (t-2,0).
    const unsigned swz = (fr >> 1) & 7;  // (row >> 1) & 7 is the same for A rows 32*wave + fr and W rows 32*j + fr
    const unsigned arow = smem_lds + (wave * 32 + fr) * ROW_B;
    const unsigned wrow = smem_lds + BM * ROW_B + fr * ROW_B;
    // byte offset of the two 16-byte chunks (2m, 2m+1), m = 2*s2 + fh, of this lane's 8-wide k slice
    const unsigned c00 = ((2u * fh) ^ swz) << 4, c01 = c00 ^ 16u;            // s2 = 0
    const unsigned c10 = ((2u * (2 + fh)) ^ swz) << 4, c11 = c10 ^ 16u;      // s2 = 1

    f32x4v p = {0.f, 0.f, 0.f, 0.f}, q = p;
    i32x4v h0, l0, h1, l1, h2, l2, h3, l3;
    i32x4v ahA = {0, 0, 0, 0}, alA = ahA, ahB = ahA, alB = ahA;

#define LDC_SB __builtin_amdgcn_sched_barrier(0)
#define LDC_MFMA(ACC, X, Y)                                                                                 \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Y), __builtin_bit_cast(bf16x8, X), ACC, 0, 0, 0); \
  LDC_SB;
    // split piece 1 of a pair (x0, x1): hi pair -> AH[i]; the two hi values as fp32 stay in t0, t1
#define LDC_SPLIT_A(AH, I, X0, X1)                                                                          \
  if constexpr (!APK) {                                                                                     \
    unsigned u_ = pack_pair(X0, X1);                                                                        \
    asm volatile("" : "+v"(u_)); /* keep ONE cvt_pk: the compiler otherwise re-converts x0 alone */         \
    AH[I] = static_cast<int>(u_);                                                                           \
    t0 = __uint_as_float(u_ << 16);                                                                         \
    t1 = __uint_as_float(u_ & 0xffff0000u);                                                                 \
    asm volatile("" : "+v"(t0), "+v"(t1)); /* anchor: IR-level sinking would move the piece to its use */   \
  }
    // split piece 2: residuals -> lo pair -> AL[i]
#define LDC_SPLIT_B(AL, I, X0, X1)                                                                          \
  if constexpr (!APK) {                                                                                     \
    unsigned lo_ = pack_pair(X0 - t0, X1 - t1);                                                             \
    asm volatile("" : "+v"(lo_));                                                                           \
    AL[I] = static_cast<int>(lo_);                                                                          \
  }
    // raw fp32 activations: two chunks of 4 floats into p, q (split below); pre-split activations
    // (LDC_GEMM_A_SPLIT): the same two chunks ARE the hi and lo fragments
#define LDC_READ_PQ(SB, C0, C1, AHN, ALN)                                                                   \
  {                                                                                                         \
    const unsigned a0_ = arow + (SB) + (C0), a1_ = arow + (SB) + (C1);                                      \
    if constexpr (APK) {                                                                                    \
      LDC_DS_READ(AHN, a0_, 0);                                                                             \
      LDC_DS_READ(ALN, a1_, 0);                                                                             \
    } else {                                                                                                \
      LDC_DS_READ(p, a0_, 0);                                                                               \
      LDC_DS_READ(q, a1_, 0);                                                                               \
    }                                                                                                       \
  }
    // one half-step: 12 MFMAs with A fragments (AH, AL); loads for H+1 from stage offset SBN with chunk offsets
    // CN0/CN1; builds (AHN, ALN); DMA part (kt DK, stage DS, half DH) when DP
#define LDC_HALF_STEP(AH, AL, AHN, ALN, SBN, CN0, CN1, W_H1, DP, DK, DS, DH)                                \
  {                                                                                                         \
    float t0 = 0.f, t1 = 0.f;                                                                               \
    const unsigned w0_ = wrow + (SBN) + (CN0), w1_ = wrow + (SBN) + (CN1);                                  \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[0], AL, h0)                                                                                \
    LDC_READ_PQ(SBN, CN0, CN1, AHN, ALN)                                                                    \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[0], AH, l0)                                                                                \
    LDC_DS_READ(l0, w1_, 0);                                                                                \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[0], AH, h0)                                                                                \
    LDC_DS_READ(h0, w0_, 0);                                                                                \
    if (DP) issue_part(DK, DS, DH, 0);                                                                      \
    W_H1                                                                                                    \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[1], AL, h1)                                                                                \
    if constexpr (APK) {                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(AHN), "+v"(ALN), "+v"(h2), "+v"(l2), "+v"(h3), "+v"(l3));  \
    } else {                                                                                                \
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(p), "+v"(q), "+v"(h2), "+v"(l2), "+v"(h3), "+v"(l3));      \
    }                                                                                                       \
    LDC_SB;                                                                                                 \
    LDC_SPLIT_A(AHN, 0, p.x, p.y)                                                                           \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[1], AH, l1)                                                                                \
    LDC_DS_READ(l1, w1_, 4096);                                                                             \
    LDC_SPLIT_B(ALN, 0, p.x, p.y)                                                                           \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[1], AH, h1)                                                                                \
    LDC_DS_READ(h1, w0_, 4096);                                                                             \
    if (DP) issue_part(DK, DS, DH, 1);                                                                      \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[2], AL, h2)                                                                                \
    LDC_SPLIT_A(AHN, 1, p.z, p.w)                                                                           \
    LDC_SPLIT_B(ALN, 1, p.z, p.w)                                                                           \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[2], AH, l2)                                                                                \
    LDC_DS_READ(l2, w1_, 8192);                                                                             \
    LDC_SPLIT_A(AHN, 2, q.x, q.y)                                                                           \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[2], AH, h2)                                                                                \
    LDC_DS_READ(h2, w0_, 8192);                                                                             \
    if (DP) issue_part(DK, DS, DH, 2);                                                                      \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[3], AL, h3)                                                                                \
    LDC_SPLIT_B(ALN, 2, q.x, q.y)                                                                           \
    LDC_SPLIT_A(AHN, 3, q.z, q.w)                                                                           \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[3], AH, l3)                                                                                \
    LDC_DS_READ(l3, w1_, 12288);                                                                            \
    LDC_SPLIT_B(ALN, 3, q.z, q.w)                                                                           \
    LDC_SB;                                                                                                 \
    LDC_MFMA(acc[3], AH, h3)                                                                                \
    LDC_DS_READ(h3, w0_, 12288);                                                                            \
    LDC_SB;                                                                                                 \
  }

    // prologue: k-steps k0 and k0+1 in flight, fragments of H = (k0,0) loaded and split
    issue(k0, 0);
    if (k0 + 1 < k1) {
      issue(k0 + 1, 1);
      if constexpr (ND == 6) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    LDC_STAMP(1 + 4 * seg_)
    if (seg_ == 0) { LDC_STAMP_CLK(13) }
    LDC_READ_PQ(0u, c00, c01, ahA, alA)
    {
      const unsigned w0_ = wrow + c00, w1_ = wrow + c01;
      LDC_DS_READ(l0, w1_, 0);
      LDC_DS_READ(h0, w0_, 0);
      LDC_DS_READ(l1, w1_, 4096);
      LDC_DS_READ(h1, w0_, 4096);
      LDC_DS_READ(l2, w1_, 8192);
      LDC_DS_READ(h2, w0_, 8192);
      LDC_DS_READ(l3, w1_, 12288);
      LDC_DS_READ(h3, w0_, 12288);
    }
    if (k0 + 2 < k1) {
#pragma unroll
      for (int i = 0; i < NDH; ++i) issue_one(k0 + 2, 2, i);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(p), "+v"(q), "+v"(ahA), "+v"(alA), "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1), "+v"(h2), "+v"(l2), "+v"(h3),
                   "+v"(l3));
    LDC_SB;
    {
      float t0 = 0.f, t1 = 0.f;
      LDC_SPLIT_A(ahA, 0, p.x, p.y)
      LDC_SPLIT_B(alA, 0, p.x, p.y)
      LDC_SPLIT_A(ahA, 1, p.z, p.w)
      LDC_SPLIT_B(alA, 1, p.z, p.w)
      LDC_SPLIT_A(ahA, 2, q.x, q.y)
      LDC_SPLIT_B(alA, 2, q.x, q.y)
      LDC_SPLIT_A(ahA, 3, q.z, q.w)
      LDC_SPLIT_B(alA, 3, q.z, q.w)
    }
    LDC_SB;

    unsigned sb = 0;          // LDS byte offset of stage kt
    int st = 0;               // stage index of k-step kt
    for (int kt = k0; kt < k1; ++kt) {
      const int st1 = st == NSTAGE - 1 ? 0 : st + 1;   // stage of kt+1
      const int st2 = st1 == NSTAGE - 1 ? 0 : st1 + 1; // stage of kt+2 ( = stage of kt-1)
      const unsigned sb1 = st1 * STAGE_B;
      const bool dma2 = kt + 2 < k1;  // second half of k-step kt+2's DMAs
      const bool dma3 = kt + 3 < k1;  // first half of k-step kt+3's DMAs (into stage st, free behind barrier(kt+1))
      // ---------------- H = (kt, 0); H+1 = (kt, 1), same stage ----------------
      // W fragments of H were issued in the previous half-step: [.. l0 h0 .. l1 h1 .. l2 h2 .. l3 h3]
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(h0), "+v"(l0));
      LDC_HALF_STEP(ahA, alA, ahB, alB, sb, c10, c11,
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(h1), "+v"(l1));, dma2, kt + 2, st2, 1)
      // ---------------- H = (kt, 1); H+1 = (kt+1, 0), next stage ----------------
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1), "+v"(h2), "+v"(l2), "+v"(h3), "+v"(l3));
      if (kt + 1 < k1) {
        // k-step kt+1 landed: its DMAs precede all ND of k-step kt+2
        if (dma2) {
          if constexpr (ND == 6) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      // (in the last k-step the reads of H+1 fetch stale ring data that is never used; they only keep the loop
      // body branch-free around registers that are in flight)
      LDC_HALF_STEP(ahB, alB, ahA, alA, sb1, c00, c01, , dma3, kt + 3, st, 0)
      st = st1;
      sb = sb1;
    }
    // drain the (unused) reads of the last half-step before their registers are reused
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(p), "+v"(q), "+v"(ahA), "+v"(alA), "+v"(ahB), "+v"(alB), "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1), "+v"(h2),
                   "+v"(l2), "+v"(h3), "+v"(l3));
#undef LDC_READ_PQ
#undef LDC_HALF_STEP
#undef LDC_SPLIT_A
#undef LDC_SPLIT_B
#undef LDC_MFMA
#undef LDC_SB
    // all waves must be done reading the ring before the next segment's prologue overwrites stage 0/1
    __builtin_amdgcn_s_barrier();
    if (seg_ == 0) { LDC_STAMP_CLK(14) }
    LDC_STAMP(2 + 4 * seg_)

    if (k0 == 0 && k1 == P.kt) {
      tile_epilogue<BM>(P, b, bm, bn, acc, wave, lane);
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
        LDC_STAMP(3 + 4 * seg_)
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
          tile_epilogue<BM>(P, b, bm, bn, acc, wave, lane);
        }
      }
    }
    u += k1 - k0;
    LDC_STAMP(4 + 4 * seg_)
#ifdef LDC_GEMM_STAMPS
    if (seg_ < 2) ++seg_;
#endif
  }
  LDC_STAMP(15)
}

template <int BM>
__global__ __launch_bounds__(BM * 2) void gemm_bf16x3_dma_fixup_kernel(SKArgs a) {
  constexpr int SLOT_FLOATS = BM * BN;
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
  tile_epilogue<BM>(P, bnb / P.tn, bm, bnb % P.tn, acc, wave, lane);
}

}  // namespace

// returns LDC_ERR_UNSUPPORTED when a problem does not fit this kernel (caller falls back)
namespace {

struct ConvParams {
  int H, W, cin, ks, kshift;
};

template <int BM, bool APK, bool CONV = false>
int launch_dma(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes, void* stream,
               const ConvParams* cp = nullptr) {
  constexpr int SLOT_FLOATS = BM * BN;
  constexpr int STAGE_B = (BM + BN) * ROW_B;
  constexpr int CUS = 256;  // one workgroup per CU
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
    {
      auto al16 = [](const void* q_) { return (reinterpret_cast<unsigned long long>(q_) & 15ull) == 0; };
      bool v4 = (d.N % 4 == 0) && (d.ldc % 4 == 0) && (d.c_bs % 4 == 0) && al16(q.C);
      if (q.bias) v4 = v4 && al16(q.bias);
      if (q.gate) v4 = v4 && al16(q.gate) && (d.gate_bs % 4 == 0);
      if (q.R) v4 = v4 && al16(q.R) && (d.ldr % 4 == 0) && (d.r_bs % 4 == 0);
      P.vec4 = v4 ? 1 : 0;
      P.c_split = (d.flags & LDC_GEMM_C_SPLIT) ? 1 : 0;
      if (P.c_split && !(v4 && d.N % 8 == 0 && d.ldc % 8 == 0 && d.c_bs % 8 == 0 &&
                         (reinterpret_cast<unsigned long long>(q.C) & 31ull) == 0))
        return LDC_ERR_ALIGN;
      if ((d.flags & LDC_GEMM_A_SPLIT) && ((d.lda % 8) || (d.a_bs % 8) || (reinterpret_cast<unsigned long long>(q.A) & 31ull)))
        return LDC_ERR_ALIGN;
      if (((d.flags & LDC_GEMM_A_SPLIT) != 0) != APK) return LDC_ERR_UNSUPPORTED;  // one activation format per launch
    }
    if constexpr (CONV) {
      P.cH = cp->H; P.cW = cp->W; P.cin = cp->cin; P.ks = cp->ks; P.kshift = cp->kshift;
      // 16 zero bytes: the tail of the counter block (zeroed by ldc_gemm_grouped_workspace_init, written by nobody)
      P.zero16 = static_cast<const unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES - 64;
    }
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
  long long G = CUS;
  {
    bool same_kt = true;
    for (int i = 1; i < n; ++i) same_kt = same_kt && (a.pr[i].kt == a.pr[0].kt);
    long long best = 0;
    if (same_kt) {
      const int kt = a.pr[0].kt;
      for (int sfac = 1; sfac <= 4; ++sfac) {
        if (kt % sfac || (sfac > 1 && kt / sfac < 8)) continue;  // whole tiles (sfac 1) at any depth
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
  a.counters = (tiles <= LDC_GEMM_COUNTER_BYTES / 4 - 16) ? static_cast<unsigned*>(workspace) : nullptr;  // last 64 B: zero page
  const size_t lds = NSTAGE * STAGE_B;
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_dma_kernel<BM, APK, CONV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    return true;
  }();
  (void)attr_set;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((gemm_bf16x3_dma_kernel<BM, APK, CONV>), dim3(a.G), dim3(BM * 2), lds, s, a);
  int st = ldc_launch_status();
  if (st != LDC_OK || a.counters != nullptr) return st;
  hipLaunchKernelGGL(gemm_bf16x3_dma_fixup_kernel<BM>, dim3(static_cast<unsigned>(tiles)), dim3(BM * 2), 0, s, a);
  return ldc_launch_status();
}

}  // namespace

int ldc_gemm_grouped_bf16x3_dma(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                                void* stream) {
  LDC_CHECK_PTR(problems);
  if (n <= 0 || n > MAXP) return LDC_ERR_ARG;
  // Tile height: 256 rows when whole tiles fill the chip (two waves per SIMD, less operand traffic per FLOP);
  // 128 rows when they do not -- half-height tiles then double the tile count, so fewer (or no) tiles are split
  // along K and fewer rows are padding.  LDC_BF16X3_BM = 128 / 256 forces one (measurement aid).
  long long tiles256 = 0;
  for (int i = 0; i < n; ++i) {
    const ldc_gemm_desc& d = problems[i].d;
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return LDC_ERR_ARG;
    tiles256 += static_cast<long long>(d.batch) * ldc_cdiv(d.M, 256) * ldc_cdiv(d.N, BN);
  }
  bool small = tiles256 <= 128;  // measured cross-over (tools/gemm_bench.py, both heights forced): about half the CUs
  static const char* const force_bm = getenv("LDC_BF16X3_BM");  // measurement aid, read once
  if (force_bm) small = (atoi(force_bm) == 128);
  const bool apk = (problems[0].d.flags & LDC_GEMM_A_SPLIT) != 0;
  if (apk)
    return small ? launch_dma<128, true>(problems, n, workspace, workspace_bytes, stream)
                 : launch_dma<256, true>(problems, n, workspace, workspace_bytes, stream);
  return small ? launch_dma<128, false>(problems, n, workspace, workspace_bytes, stream)
               : launch_dma<256, false>(problems, n, workspace, workspace_bytes, stream);
}

// SphereConv2d (dense, stride 1, k = 3 / 5) as an implicit GEMM on the split-bf16 kernel: X is NHWC fp32 [B*H*W][ldx],
// Wp = ldc_pack_weight_bf16x2 of the [cout][k*k][cin rounded up to 32 * 2^j] tap-major weight (zero behind cin).
extern "C" int ldc_sphere_conv_nhwc_bf16x3(const float* X, const void* Wp, const float* bias, const float* R, float* Y, int B,
                                           int H, int W, int cin, int ldx, int cout, int ldy, int ldr, int ksize, int act,
                                           void* workspace, long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(X);
  LDC_CHECK_PTR(Wp);
  LDC_CHECK_PTR(Y);
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return LDC_ERR_ARG;
  if (ksize != 1 && ksize != 3 && ksize != 5) return LDC_ERR_UNSUPPORTED;  // 1 = pointwise conv / Linear with any cin % 4 == 0
  if (ksize > 1 && ((W & 1) || H < 2 || H < ksize / 2)) return LDC_ERR_UNSUPPORTED;  // reference asserts even width
  if ((cin & 3) || (ldx & 3) || ldx < cin || ldy < cout) return LDC_ERR_ALIGN;
  const long long M = static_cast<long long>(B) * H * W;
  if (M > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  ConvParams cp{H, W, cin, ksize, 0};
  int ktpt = ldc_cdiv(cin, BK);  // k-steps per tap, rounded up to a power of two (the kernel shifts instead of dividing)
  while ((1 << cp.kshift) < ktpt) ++cp.kshift;
  ktpt = 1 << cp.kshift;
  ldc_gemm_problem q{};
  q.A = X;
  q.W = static_cast<const float*>(Wp);
  q.bias = bias;
  q.R = R;
  q.C = Y;
  q.d.M = static_cast<int>(M);
  q.d.N = cout;
  q.d.K = ksize * ksize * ktpt * BK;
  q.d.batch = 1;
  q.d.lda = ldx;
  q.d.ldw = q.d.K;
  q.d.ldc = ldy;
  q.d.ldr = ldr;
  q.d.act = act;
  const long long tiles256 = static_cast<long long>(ldc_cdiv(M, 256)) * ldc_cdiv(cout, BN);
  return tiles256 <= 128 ? launch_dma<128, false, true>(&q, 1, workspace, workspace_bytes, stream, &cp)
                         : launch_dma<256, false, true>(&q, 1, workspace, workspace_bytes, stream, &cp);
}
