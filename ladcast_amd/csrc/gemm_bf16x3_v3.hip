// Split-bf16 ("bf16x3") stream-K GEMM, third generation: pre-split activations (LDC_GEMM_A_SPLIT) only,
// v_mfma_f32_16x16x32_bf16, 8 waves for both tile heights.
//
// Why a third kernel (measurements of round 1's fp32-activation LDS-DMA kernel, removed in round 6; in-kernel clock stamps, tools/gemm_stamps.py):
//   * the chip is at its power limit in these loops -- the 256-row kernel at 76 % MFMA busy ran at 1.30 GHz, the
//     128-row kernel at 52 % busy at 1.88 GHz, both delivering about the same FLOP/s per CU -- so what is left to
//     gain is energy per FLOP.  MI355X_MICROARCH.md (DVFS give-back, item 7) measures the 16x16x32 shape at
//     1.12-1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP (half the accumulator register traffic per FLOP);
//   * with the producers writing activations already split (LayerNorm, attention, GEMM epilogues), the main loop has
//     no VALU work left, so one k-step is just 2 + 16 fragment reads, 6 (4) DMA instructions and 48 (24) MFMAs;
//   * a 16-row MFMA lets the 128-row tile keep EIGHT waves (16 rows each, two per SIMD) instead of four 32-row waves
//     at one per SIMD, which is what made the half-height tile 20 % slower per CU.
// Tile BM x 128 x 32 per k-step, BM = 256 (wave = 2 row tiles of 16) or 128 (1 row tile); operands through the same
// 3-stage LDS-DMA ring, unit-range stream-K scheduling, XCD-aware placement and in-launch "last arriver reduces"
// hand-off as the round-1 kernel it replaced (protocol: the publish / ticket / last-arriver block below); only the LDS image swizzle, the k-step body and the
// accumulator layout differ:
//   * a row of a stage is 4 k-groups x [hi 16 B | lo 16 B]; chunk c = 2 kg + (0 hi | 1 lo) of row r sits in slot
//     c ^ f(r), f(r) = ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1): conflict-free for the four 16-lane groups of
//     a ds_read_b128 when lanes 0-15 / 16-31 / 32-47 / 48-63 read k-group 0 / 1 / 2 / 3 of 16 consecutive rows;
//   * MFMA(W fragment, A fragment): a lane owns output row m = lane % 16 and 4 consecutive columns
//     n = 16 ct + 4 (lane / 16) + r of column tile ct -> the epilogue stores 16 bytes per lane and accumulator.
// k-step kt: column tiles 0-3 (their W fragments were loaded during k-step kt-1), reloading each slot with column
// tile +4 of the same stage | column tile 4 | barrier(kt+1) | A fragments of kt+1 | column tiles 5-7, reloading with tiles
// 0-3 of kt+1 (the barrier's lgkmcnt(0) then waits for reads that were issued a whole column tile earlier).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

#include "gemm_v3_common.inc"
#include "gemm_v3_kstep_f32.inc"

struct SKArgs {
  DevProblem pr[MAXP];
  int np;
  int G;
  long long U;
  long long tiles;
  float* ws;
  unsigned* counters;  // one per tile
  unsigned upg, urem;  // U / G and U % G: unit range g = [g upg + min(g, urem), ...) - sizes differ by at most one, no division in the kernel
};

// Round 3: the unit ranges are cut without a division (the kernel's start-up was 2.5 / 5.8 us of scalar code before the first DMA
// instruction, most of it inlined 64-bit divisions: profiles/r03_f_gemm_startup_diagnostics.log).  U < 2^31 (checked by the host).
__device__ __forceinline__ long long range_start(long long g, const SKArgs& a) {
  const unsigned gu = static_cast<unsigned>(g);
  return static_cast<long long>(gu * a.upg + (gu < a.urem ? gu : a.urem));
}

__device__ __forceinline__ int find_problem_by_unit(const SKArgs& a, long long u) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < MAXP; ++k)
    if (k < a.np && u >= a.pr[k].unit0) pi = k;
  return pi;
}

// Diagnostic build only (-DLDC_GEMM_STAMPS, `make stamps`): wall-clock / shader-clock stamps per workgroup into the
// upper half of the counter block, which nothing else reads; the product build contains no stamp.
#ifdef LDC_GEMM_STAMPS
#define LDC_STAMP(i)                                                                                        \
  if (threadIdx.x == 0)                                                                                     \
    reinterpret_cast<unsigned long long*>(a.counters)[65536 + blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#define LDC_STAMP_CLK(i)                                                                                    \
  if (threadIdx.x == 0)                                                                                     \
    reinterpret_cast<unsigned long long*>(a.counters)[65536 + blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define LDC_STAMP(i)
#define LDC_STAMP_CLK(i)
#endif

// TERMS = 3: split-bf16 contraction Ah.Wh + Ah.Wl + Al.Wh over operand rows of [hi x8 | lo x8] groups (32 k per 128-byte k-step).
// TERMS = 1 (LDC_GEMM_BF16_1TERM, the "bf16" mixed-precision mode): the operands are PLAIN bf16 rows, so a 128-byte k-step holds
// 64 k; the same DMAs, LDS image and fragment reads - the chunk the split mode calls "hi" of k-group g is k 16 g .. 16 g + 7 here,
// its "lo" chunk k 16 g + 8 .. 16 g + 15 (any assignment of k to the two MFMA k-blocks works as long as A and W agree) - and two
// MFMAs per column tile, (first chunks) + (second chunks), instead of three: two thirds of the MFMAs for twice the k.
// CONV: the A rows are gathered - row m of k-step (tap, chunk) is the 128-byte channel chunk of the tap's source pixel under the
// sphere padding rule, a per-lane DMA source recomputed when a tap starts (wave-uniform branch, 2^kshift k-steps apart); everything
// behind the DMA issue is the GEMM.  X is NHWC in the split format (TERMS = 3: 32 channels per k-step) or plain bf16 rows (TERMS = 1:
// 64 channels per k-step), written by the DCAE's producers (dcae.hip).
// TERMS = 0 (round 3, the exact-fp32 mode): the operands are PLAIN fp32 rows - 32 k per 128-byte k-step, the same row geometry as the
// split groups - through the same ring, swizzle and fragment reads; the lane that the split mode hands the hi / lo chunks of k-group
// g holds k = 8 g .. 8 g + 7 as floats, and MFMA j of a column tile (v_mfma_f32_16x16x4_f32: one float per lane and operand, 4 k per
// instruction) takes element j of both operands: k = 8 (lane / 16) + j.  8 MFMAs of 32 cycles per column and row tile: the loop is
// matrix-pipe bound by construction (2.7x the split mode's MFMA cycles per k-step).  Launched with 128-row tiles only.
template <int BM, int TERMS, bool CONV = false>
__global__ __launch_bounds__(512) void gemm_bf16x3_v3_kernel(SKArgs a) {
  constexpr int CPK = TERMS == 1 ? 2 * BK : BK;  // CONV: channels per 128-byte k-step (plain bf16 | split groups, fp32)
  constexpr int RT = BM / 128;                 // 16-row tiles per wave
  constexpr int NACC = RT * 8;                 // accumulators (f32x4) per lane
  constexpr int STAGE_B = (BM + BN) * ROW_B;   // 48 KiB / 32 KiB
  constexpr int SLOT_FLOATS = BM * BN;
  constexpr int NAI = BM / 64;                 // A DMA instructions per wave and k-step (4 / 2); W: always 2
  constexpr int ND = NAI + 2;                  // DMA instructions per wave and k-step (6 / 4)
  constexpr int NDH = ND / 2;                  // issued per half k-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  LDC_STAMP(0)
  int seg_ = 0;
  (void)seg_;
  int g;
  {  // XCD-aware placement: each XCD gets a contiguous run of unit ranges
    const int bid = blockIdx.x, G = a.G;
    const int q = G >> 3, r = G & 7, xcd = bid & 7;
    g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // the 64-bit division runs on the vector ALU: pin the (wave-uniform) results into scalar registers so that the whole
  // k loop is scalar control flow
  auto uniform64 = [](long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(static_cast<unsigned long long>(v) >> 32));
    return static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
  };
  const long long u_begin = uniform64(range_start(g, a));
  const long long u_end = uniform64(range_start(g + 1, a));
  const unsigned smem_lds = lds_addr(smem);

  // One SEGMENT = the k-steps [k0, k1) of one output tile (a whole tile when the unit ranges are tile-aligned).  Its coordinates are
  // wave-uniform scalars decoded with 32-bit arithmetic; the NEXT segment is decoded while the current one's prologue DMAs are in flight.
  struct Seg {
    int pi, tile, k0, k1, b, bm, bn;
  };
  auto decode = [&](long long u_) {
    Seg sg;
    sg.pi = find_problem_by_unit(a, u_);
    const DevProblem& Q = a.pr[sg.pi];
    const unsigned local = static_cast<unsigned>(u_ - Q.unit0), kt_u = static_cast<unsigned>(Q.kt);
    sg.tile = __builtin_amdgcn_readfirstlane(static_cast<int>(local / kt_u));
    sg.k0 = static_cast<int>(local - static_cast<unsigned>(sg.tile) * kt_u);
    const long long left = u_end - u_;
    sg.k1 = (Q.kt - sg.k0 <= left) ? Q.kt : sg.k0 + static_cast<int>(left);
    // tile order (speed only): batch, then super-rows of rm row tiles, then column panels, row tile fastest.  An XCD owns a
    // contiguous run of units, i.e. about rm row panels x (tiles per XCD / rm) column panels: with rm = tm (one super-row)
    // every XCD streams ALL of A and 1/8 of W; a squarer block fetches fewer bytes through the fabric (launch_v3 picks rm)
    const unsigned per_b = static_cast<unsigned>(Q.tm * Q.tn), tile_u = static_cast<unsigned>(sg.tile);
    const unsigned b_ = tile_u / per_b;
    const unsigned t_in = tile_u - b_ * per_b;
    const unsigned strip = static_cast<unsigned>(Q.rm * Q.tn);
    const unsigned sr = t_in / strip;
    const unsigned r_in = t_in - sr * strip;
    const int rest = Q.tm - static_cast<int>(sr) * Q.rm;
    const unsigned h_sr = static_cast<unsigned>(rest < Q.rm ? rest : Q.rm);
    const unsigned bn_ = r_in / h_sr;
    sg.b = __builtin_amdgcn_readfirstlane(static_cast<int>(b_));
    sg.bn = __builtin_amdgcn_readfirstlane(static_cast<int>(bn_));
    sg.bm = __builtin_amdgcn_readfirstlane(static_cast<int>(sr) * Q.rm + static_cast<int>(r_in - bn_ * h_sr));
    return sg;
  };
  // Round 5 (exact-fp32 instances): the NEXT segment's first two k-steps leave for stages 0 / 1 as soon as this segment's loop has released the ring - before its
  // epilogue / hand-off instead of after it: the 2.3 - 2.8 us between "tile done" and "first DMA of the next tile landed" (launch stamps,
  // profiles/r04_x_gemm_launch_stamps.log) overlap the 2.8 - 4 us epilogue.  The sources are formed exactly as the segment's own prologue
  // forms them (same order of issue); every static vmcnt below still holds - the epilogue's loads and stores are issued AFTER these DMAs,
  // so a wait that leaves the N youngest operations outstanding covers at least what it covered before.
  auto prefetch_next = [&](const Seg& sg) __attribute__((always_inline)) {
    const DevProblem& Q = a.pr[sg.pi];
    const int ln = fresh_lane();
    const int lr_ = ln >> 3, lp_ = ln & 7;
    const float* __restrict__ An = Q.A + static_cast<long long>(sg.b) * Q.d.a_bs;
    const long long wrb = static_cast<long long>(Q.d.K) * (TERMS == 1 ? 2 : 4);
    unsigned char* dump = smem + NSTAGE * STAGE_B + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool live = sg.k0 + j < sg.k1;
      const long long koff = static_cast<long long>(live ? sg.k0 + j : sg.k1 - 1) * (BK * 4);
      unsigned char* sA = smem + j * STAGE_B;
      unsigned char* sW = sA + BM * ROW_B;
#pragma unroll
      for (int i = 0; i < NAI; ++i) {
        const int r = 8 * (wave + 8 * i) + lr_;
        int gm = sg.bm * BM + r;
        gm = gm < Q.d.M ? gm : Q.d.M - 1;
        dma16(reinterpret_cast<const unsigned char*>(An + static_cast<long long>(gm) * Q.d.lda) + ((lp_ ^ swz(r)) << 4) + koff,
              live ? sA + (wave + 8 * i) * 1024 : dump);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 8 * (wave + 8 * i) + lr_;
        int gn = sg.bn * BN + r;
        gn = gn < Q.d.N ? gn : Q.d.N - 1;
        dma16(Q.W + static_cast<long long>(gn) * wrb + ((lp_ ^ swz(r)) << 4) + koff, live ? sW + (wave + 8 * (i)) * 1024 : dump);
      }
    }
  };
  long long u = u_begin;
  Seg cur{};
  if (u < u_end) cur = decode(u);
  bool prefetched = false;  // (wave-uniform) this segment's prologue DMAs are already in flight
  while (u < u_end) {
    const int pi = cur.pi;
    const DevProblem& P = a.pr[pi];
    const int tile = cur.tile, k0 = cur.k0, k1 = cur.k1, b = cur.b, bm = cur.bm, bn = cur.bn;
    // per-lane geometry, formed per segment from a fresh lane index (fresh_lane: nothing per-lane is live across the epilogue)
    const int lane = fresh_lane();
    const int fr = lane & 15;   // fragment row (of the 16-row / 16-column MFMA tile)
    const int kg = lane >> 4;   // k-group: k = 8 kg .. 8 kg + 7 of the k-step
    // DMA lane geometry: a wave instruction covers 8 rows x 128 B; lane -> (row lr, 16-byte slot lp)
    const int lr = lane >> 3, lp = lane & 7;
    // fragment read addresses inside a stage: A row tile rt of this wave, W column tile ct; hi chunk 2 kg, lo chunk 2 kg + 1
    const unsigned f_r = swz(fr);  // f only depends on r & 15, and every 16-row tile starts at a multiple of 16
    const unsigned off_hi = fr * ROW_B + (((2 * kg) ^ f_r) << 4);
    const unsigned off_lo = fr * ROW_B + (((2 * kg + 1) ^ f_r) << 4);
    const unsigned a_hi = smem_lds + (16 * RT * wave) * ROW_B + off_hi;  // + rt * 16 * ROW_B
    const unsigned a_lo = smem_lds + (16 * RT * wave) * ROW_B + off_lo;
    const unsigned w_hi = smem_lds + BM * ROW_B + off_hi;                // + ct * 16 * ROW_B
    const unsigned w_lo = smem_lds + BM * ROW_B + off_lo;
    const int M = P.d.M, N = P.d.N, K = P.d.K;
    // a wave whose 16 RT rows all lie past the last row of a ragged tile (1800 = 14 x 128 + 8 rows: seven of the last tile's eight
    // waves) issues no MFMA: its share of the tile's matrix-core energy is what the power-limited launch gets back as clock
    const bool wave_rows = bm * BM + 16 * RT * wave < M;
    const float* __restrict__ A = P.A + static_cast<long long>(b) * P.d.a_bs;
    const int lda = P.d.lda;
    const long long w_row_bytes = static_cast<long long>(K) * (TERMS == 1 ? 2 : 4);  // plain bf16 row | packed split row, fp32 row

    // per-lane DMA sources (k-step 0); rows past the edge are clamped (their outputs are never stored).
    // A instruction q (0..BM/8-1) covers tile rows [8q, 8q+8): this wave issues q = wave + 8 i; W likewise (16 of them)
#ifdef LDC_GEMM_STAMPS_PROLOGUE  // one-off diagnosis of the first segment's start-up (launches of <= 2 segments: slots 9-11 are free)
    if (seg_ == 0) { LDC_STAMP(9) }
#endif
    const unsigned char* a_src[NAI];
    const unsigned char* w_src[2];
    int pix_b[NAI], pix_hw[NAI];  // CONV: first pixel of the row's image, (h << 16 | w) inside it
    // the 16-byte slot this lane fills is the same for all its A instructions: swz(r) only looks at r & 15 = 8 (wave & 1) + lr
    const int cslot = lp ^ swz(8 * (wave & 1) + lr);
    auto conv_tap_base = [&](int i, int tap) {  // CONV: this lane's chunk of the source pixel of tap `tap` (channel chunk 0)
      const int ky = tap / P.ks, kx = tap - ky * P.ks;
      const int src = pix_b[i] + ldc_sphere_src_pixel(pix_hw[i] >> 16, pix_hw[i] & 0xffff, ky, kx, P.cH, P.cW, P.ks);
      return reinterpret_cast<const unsigned char*>(A + static_cast<long long>(src) * lda) + (cslot << 4);
    };
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int r = 8 * (wave + 8 * i) + lr;
      int gm = bm * BM + r;
      gm = gm < M ? gm : M - 1;
      if constexpr (CONV) {
        const int hw = P.cH * P.cW;
        const int bimg = gm / hw;
        const int rem = gm - bimg * hw;
        const int h = rem / P.cW;
        pix_b[i] = bimg * hw;
        pix_hw[i] = (h << 16) | (rem - h * P.cW);
        a_src[i] = conv_tap_base(i, k0 >> P.kshift);
      } else {
        a_src[i] = reinterpret_cast<const unsigned char*>(A + static_cast<long long>(gm) * lda) + ((lp ^ swz(r)) << 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (wave + 8 * i) + lr;
      int gn = bn * BN + r;
      gn = gn < N ? gn : N - 1;
      w_src[i] = P.W + static_cast<long long>(gn) * w_row_bytes + ((lp ^ swz(r)) << 4);
    }
    // k-steps past the end of the segment are clamped to its last one and land in a per-wave dump slot behind the ring:
    // the loop body then has no branch around a DMA, every vmcnt is static, and nothing stale can land in the ring (or
    // on the hand-off flag word in it) after the segment
    auto issue_one = [&](int kt_, int stage, int i) {
      const bool live = kt_ < k1;
      unsigned char* sA = smem + stage * STAGE_B;
      unsigned char* sW = sA + BM * ROW_B;
      unsigned char* dump = smem + NSTAGE * STAGE_B + wave * 1024;
      const int kt = live ? kt_ : k1 - 1;
      const long long koff = static_cast<long long>(kt) * (BK * 4);  // 128 B per k-step in both operands
#ifdef LDC_GEMM_DIAG_NODMA  // diagnostic build: no operand traffic at all (results are garbage)
      if (kt >= 0) return;
#endif
      if (i < NAI) {
        if constexpr (CONV) {
          // piece i is issued once per k-step in increasing kt (clamped repeats of the last one included), so its base is
          // recomputed exactly when a tap starts
          const int chunk = kt & ((1 << P.kshift) - 1);
          if (chunk == 0) a_src[i] = conv_tap_base(i, kt >> P.kshift);
          const unsigned char* src = a_src[i] + chunk * (BK * 4);
          if (chunk * CPK + CPK > P.cin) {  // the tap's last k-step: 8-column groups behind cin read zeros (their weights are zero too)
            // this lane's 16-byte slot: split rows - the hi or lo half of group cslot / 2; plain bf16 rows - columns 8 cslot .. + 7
            if (chunk * CPK + (TERMS == 0 ? 4 * cslot : 8 * (TERMS == 3 ? (cslot >> 1) : cslot)) >= P.cin) src = P.zero16;
          }
          dma16(src, live ? sA + (wave + 8 * i) * 1024 : dump);
        } else {
          dma16(a_src[i] + koff, live ? sA + (wave + 8 * i) * 1024 : dump);
        }
      } else {
        dma16(w_src[i - NAI] + koff, live ? sW + (wave + 8 * (i - NAI)) * 1024 : dump);
      }
    };

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x4v wh0, wl0, wh1, wl1, wh2, wl2, wh3, wl3;           // W fragment window: 4 column tiles
    i32x4v ah0[RT], al0[RT], ah1[RT], al1[RT];               // A fragments of the current / next k-step

#define LDC_SB __builtin_amdgcn_sched_barrier(0)
    // SBH / SBL: lane base (hi / lo chunk) + stage offset; the tile offset is an immediate
#define LDC_RD_W(WH, WL, SBH, SBL, CT)                                           \
  {                                                                              \
    LDC_DS_READ(WH, SBH, (CT) * (16 * ROW_B));                                   \
    LDC_DS_READ(WL, SBL, (CT) * (16 * ROW_B));                                   \
  }
#define LDC_RD_A(AH, AL, SBH, SBL)                                               \
  {                                                                              \
    LDC_DS_READ(AH[0], SBH, 0);                                                  \
    LDC_DS_READ(AL[0], SBL, 0);                                                  \
    if constexpr (RT == 2) {                                                     \
      LDC_DS_READ(AH[RT - 1], SBH, 16 * ROW_B);                                  \
      LDC_DS_READ(AL[RT - 1], SBL, 16 * ROW_B);                                  \
    }                                                                            \
  }
#define LDC_MM(ACC, WF, AF)                                                                                  \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WF), __builtin_bit_cast(bf16x8, AF), ACC, 0, 0, 0); \
  LDC_SB;
    // the 3 RT MFMAs of column tile CT with A fragments (AH, AL); with two row tiles their accumulation chains alternate
#define LDC_MMF(ACC, WF, AF, J)                                                                               \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(__int_as_float(WF[J]), __int_as_float(AF[J]), ACC, 0, 0, 0);     \
  LDC_SB;
#define LDC_CT(CT, WH, WL, AH, AL)                                               \
  if (wave_rows) {                                                               \
    if constexpr (TERMS == 0) {                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                         \
        LDC_MMF(acc[(CT)], WH, AH[0], j_)                                        \
        if constexpr (RT == 2) {                                                 \
          LDC_MMF(acc[8 + (CT)], WH, AH[RT - 1], j_)                             \
        }                                                                        \
      }                                                                          \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                         \
        LDC_MMF(acc[(CT)], WL, AL[0], j_)                                        \
        if constexpr (RT == 2) {                                                 \
          LDC_MMF(acc[8 + (CT)], WL, AL[RT - 1], j_)                             \
        }                                                                        \
      }                                                                          \
    } else if constexpr (TERMS == 1) {                                           \
      LDC_MM(acc[(CT)], WH, AH[0])                                               \
      if constexpr (RT == 2) {                                                   \
        LDC_MM(acc[8 + (CT)], WH, AH[RT - 1])                                    \
      }                                                                          \
      LDC_MM(acc[(CT)], WL, AL[0])                                               \
      if constexpr (RT == 2) {                                                   \
        LDC_MM(acc[8 + (CT)], WL, AL[RT - 1])                                    \
      }                                                                          \
    } else if constexpr (RT == 2) {                                              \
      LDC_MM(acc[(CT)], WH, AL[0])                                               \
      LDC_MM(acc[8 + (CT)], WH, AL[RT - 1])                                      \
      LDC_MM(acc[(CT)], WL, AH[0])                                               \
      LDC_MM(acc[8 + (CT)], WL, AH[RT - 1])                                      \
      LDC_MM(acc[(CT)], WH, AH[0])                                               \
      LDC_MM(acc[8 + (CT)], WH, AH[RT - 1])                                      \
    } else {                                                                     \
      LDC_MM(acc[(CT)], WH, AL[0])                                               \
      LDC_MM(acc[(CT)], WL, AH[0])                                               \
      LDC_MM(acc[(CT)], WH, AH[0])                                               \
    }                                                                            \
  }
#ifdef LDC_GEMM_DIAG_NOBARRIER  // diagnostic build: what the k-step costs without its barrier (results are garbage)
#define LDC_KSTEP_BARRIER
#else
#define LDC_KSTEP_BARRIER __builtin_amdgcn_s_barrier();
#endif
    // wait until the fragment window holds at most 3 column tiles (6 reads) of outstanding reads
#define LDC_WAIT(N, X, Y)                                                        \
  asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(X), "+v"(Y));                       \
  LDC_SB;
    // one k-step: (AH, AL) current A fragments, (AHN, ALN) receive those of k-step kt+1
#define LDC_KSTEP(AH, AL, AHN, ALN)                                                                          \
  {                                                                                                          \
    const int st1 = st == NSTAGE - 1 ? 0 : st + 1;                                                           \
    const int st2 = st1 == NSTAGE - 1 ? 0 : st1 + 1;                                                         \
    const unsigned sb1 = st1 * STAGE_B;                                                                      \
    const unsigned wch = w_hi + sb, wcl = w_lo + sb, wnh = w_hi + sb1, wnl = w_lo + sb1;                     \
    const unsigned anh = a_hi + sb1, anl = a_lo + sb1;                                                       \
    constexpr bool dma2 = true; /* second half of k-step kt+2's DMAs (clamped past the end) */             \
    constexpr bool dma3 = true; /* first half of k-step kt+3's (into stage st, free behind barrier(kt+1)) */ \
    /* W(kt, 0..3) were issued in the order 0, 1, 2, 3 and nothing after them except A(kt) before them */    \
    if constexpr (RT == 2) {                                                                                 \
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(AH[0]), "+v"(AL[0]), "+v"(AH[1]), "+v"(AL[1]), "+v"(wh0), "+v"(wl0)); \
    } else {                                                                                                 \
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(AH[0]), "+v"(AL[0]), "+v"(wh0), "+v"(wl0));                 \
    }                                                                                                        \
    LDC_SB;                                                                                                  \
    LDC_CT(0, wh0, wl0, AH, AL)                                                                              \
    LDC_RD_W(wh0, wl0, wch, wcl, 4)                                                                                \
    if (dma2) issue_one(kt + 2, st2, NDH + 0);                                                               \
    LDC_WAIT(6, wh1, wl1)                                                                                    \
    LDC_CT(1, wh1, wl1, AH, AL)                                                                              \
    LDC_RD_W(wh1, wl1, wch, wcl, 5)                                                                                \
    if (dma2) issue_one(kt + 2, st2, NDH + 1);                                                               \
    LDC_WAIT(6, wh2, wl2)                                                                                    \
    LDC_CT(2, wh2, wl2, AH, AL)                                                                              \
    LDC_RD_W(wh2, wl2, wch, wcl, 6)                                                                                \
    if constexpr (NDH == 3) {                                                                                \
      if (dma2) issue_one(kt + 2, st2, NDH + 2);                                                             \
    }                                                                                                        \
    LDC_WAIT(6, wh3, wl3)                                                                                    \
    LDC_CT(3, wh3, wl3, AH, AL)                                                                              \
    LDC_RD_W(wh3, wl3, wch, wcl, 7)                                                                          \
    LDC_WAIT(6, wh0, wl0)                                                                                    \
    LDC_CT(4, wh0, wl0, AH, AL)                                                                              \
    /* barrier(kt+1): this wave is done READING stage kt (its last reads were issued one column tile ago);  \
       its DMAs of k-step kt+1 landed */                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh1), "+v"(wl1), "+v"(wh2), "+v"(wl2), "+v"(wh3), "+v"(wl3)); \
    if constexpr (ND == 6) {                                                                                 \
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                       \
    } else {                                                                                                 \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                       \
    }                                                                                                        \
    LDC_KSTEP_BARRIER                                                                                        \
    LDC_SB;                                                                                                  \
    /* (in the last k-step the reads below fetch stale ring data that is never used) */                      \
    LDC_RD_A(AHN, ALN, anh, anl)                                                                             \
    LDC_RD_W(wh0, wl0, wnh, wnl, 0)                                                                          \
    if (dma3) issue_one(kt + 3, st, 0);                                                                      \
    LDC_SB;                                                                                                  \
    LDC_CT(5, wh1, wl1, AH, AL)                                                                              \
    LDC_RD_W(wh1, wl1, wnh, wnl, 1)                                                                               \
    if (dma3) issue_one(kt + 3, st, 1);                                                                      \
    LDC_SB;                                                                                                  \
    LDC_CT(6, wh2, wl2, AH, AL)                                                                              \
    LDC_RD_W(wh2, wl2, wnh, wnl, 2)                                                                               \
    if constexpr (NDH == 3) {                                                                                \
      if (dma3) issue_one(kt + 3, st, 2);                                                                    \
    }                                                                                                        \
    LDC_SB;                                                                                                  \
    LDC_CT(7, wh3, wl3, AH, AL)                                                                              \
    LDC_RD_W(wh3, wl3, wnh, wnl, 3)                                                                               \
    LDC_SB;                                                                                                  \
    st = st1;                                                                                                \
    sb = sb1;                                                                                                \
  }

    // Round 6, exact-fp32 128-row instances: LDC_KSTEP_Q (gemm_v3_kstep_f32.inc, generated by tools/gen_gemm_f32_kstep.py) - column tiles in pairs
    // whose MFMAs alternate, every fragment read / DMA issue in the gap behind one of the wave's own MFMAs.  MM: the MFMA itself, or nothing for a
    // wave whose rows all lie past the end of a ragged tile (the branch sits around the whole loop, not around every MFMA).
#define LDC_MM_ON(ACC, WF, AF, J) LDC_MMF(ACC, WF, AF, J)
#define LDC_MM_OFF(ACC, WF, AF, J)
#if defined(LDC_AB_BUILD) && defined(LDC_GEMM_F32_NO_PAIRS)  // A/B aid (make variant DIAG=-DLDC_GEMM_F32_NO_PAIRS): round 5's k-step
  constexpr bool PAIRED = false;
#else
  constexpr bool PAIRED = TERMS == 0 && RT == 1;
#endif

    // prologue: k-steps k0 and k0+1 in flight, fragments of k0 (A, W column tiles 0-3) loading
#ifdef LDC_GEMM_STAMPS_PROLOGUE
    if (seg_ == 0) { LDC_STAMP(10) }
#endif
    if (!prefetched) {
#pragma unroll
      for (int i = 0; i < ND; ++i) issue_one(k0, 0, i);
#pragma unroll
      for (int i = 0; i < ND; ++i) issue_one(k0 + 1, 1, i);
    }
#ifdef LDC_GEMM_STAMPS_PROLOGUE
    if (seg_ == 0) { LDC_STAMP(11) }
#endif
    // the next segment's coordinates, while this one's first two stages are in flight
    const long long u_next = u + (k1 - k0);
    Seg nxt = cur;
    if (u_next < u_end) nxt = decode(u_next);
    if constexpr (ND == 6) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    LDC_STAMP(1 + 4 * seg_)
    if (seg_ == 0) { LDC_STAMP_CLK(13) }
    if constexpr (PAIRED) {
      LDC_KSTEP_Q_PROLOGUE(ah0, al0)
    } else {
      LDC_RD_A(ah0, al0, a_hi, a_lo)
      LDC_RD_W(wh0, wl0, w_hi, w_lo, 0)
      LDC_RD_W(wh1, wl1, w_hi, w_lo, 1)
      LDC_RD_W(wh2, wl2, w_hi, w_lo, 2)
      LDC_RD_W(wh3, wl3, w_hi, w_lo, 3)
    }
#pragma unroll
    for (int i = 0; i < NDH; ++i) issue_one(k0 + 2, 2, i);
    LDC_SB;

    unsigned sb = 0;  // LDS byte offset of stage kt
    int st = 0;       // stage index of k-step kt
    if constexpr (PAIRED) {
      static_assert(!PAIRED || ND == 4, "the interleaved k-step is written for the 128-row tile's 4 DMA pieces");
      if (wave_rows) {
        for (int kt = k0; kt < k1; ++kt) {
          LDC_KSTEP_Q(ah0, al0, ah1, al1, LDC_MM_ON)
          ++kt;
          if (kt >= k1) break;
          LDC_KSTEP_Q(ah1, al1, ah0, al0, LDC_MM_ON)
        }
      } else {
        for (int kt = k0; kt < k1; ++kt) {
          LDC_KSTEP_Q(ah0, al0, ah1, al1, LDC_MM_OFF)
          ++kt;
          if (kt >= k1) break;
          LDC_KSTEP_Q(ah1, al1, ah0, al0, LDC_MM_OFF)
        }
      }
    } else {
      for (int kt = k0; kt < k1; ++kt) {
        LDC_KSTEP(ah0, al0, ah1, al1)
        ++kt;
        if (kt >= k1) break;
        LDC_KSTEP(ah1, al1, ah0, al0)
      }
    }
    // drain the (unused) reads of the last k-step before their registers are reused
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh0), "+v"(wl0), "+v"(wh1), "+v"(wl1), "+v"(wh2), "+v"(wl2), "+v"(wh3), "+v"(wl3));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) asm volatile("" : "+v"(ah0[rt]), "+v"(al0[rt]), "+v"(ah1[rt]), "+v"(al1[rt]));
#undef LDC_KSTEP
#undef LDC_MM_ON
#undef LDC_MM_OFF
#undef LDC_WAIT
#undef LDC_CT
#undef LDC_MMF
#undef LDC_MM
#undef LDC_RD_A
#undef LDC_RD_W
#undef LDC_SB
    // all waves must be done reading the ring before the next segment's prologue overwrites stage 0/1
    __builtin_amdgcn_s_barrier();
    prefetched = false;
    // exact-fp32 mode only: measured same-box (profiles/r05_q_gemm_prologue_prefetch_ab.log) 3.051 -> 3.077 member-steps/s there (the chip
    // is at 2.39 GHz / 1190 W: idle time is lost time) and nothing in the split mode (dual QKV 99.7 vs 100.3 us, chunk 109.5 vs 109.6 ms: at
    // the power cap the gap's energy comes back as clock), where it also costs the 256-row instance 79 more spilled registers
    if constexpr (!CONV && TERMS == 0) {
#if !defined(LDC_GEMM_DIAG_NODMA) && !defined(LDC_GEMM_NO_PREFETCH)
      if (u_next < u_end) {
        prefetch_next(nxt);
        prefetched = true;
      }
#endif
    }
    if (seg_ == 0) { LDC_STAMP_CLK(14) }
    LDC_STAMP(2 + 4 * seg_)

    if (k0 == 0 && k1 == P.kt) {
      tile_epilogue<BM, !CONV, LinearRows<BM>, true, TERMS == 0>(P, b, bm, bn, acc, wave, lane);
    } else {
      // ---- publish this piece (write-through slab, drained, ONE ticket per workgroup); the piece whose ticket is
      // the last re-reads all slabs of the tile in workgroup order and applies the epilogue ----
      const int lane_h = fresh_lane();  // slab addresses are formed here, not hoisted to the kernel entry and spilled across the main loop
      float* slot = a.ws + (static_cast<long long>(2 * g) + (k0 > 0 ? 0 : 1)) * SLOT_FLOATS + lane_h;
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_store(slot + ((wave * NACC + i) * 4 + r) * 64, acc[i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const long long f = P.unit0 + static_cast<long long>(tile) * P.kt;
      const long long l = f + P.kt;
      long long g_first = g;
      while (g_first > 0 && range_start(g_first, a) > f) --g_first;
      long long g_last = g;
      while (g_last + 1 < a.G && range_start(g_last + 1, a) < l) ++g_last;
      const unsigned pieces = static_cast<unsigned>(g_last - g_first + 1);
      unsigned* cnt = a.counters + (P.tile0 + tile);
      // (the ring may already hold the next segment's first k-steps: the flag lives in wave 0's dump slot behind the ring.  Thread 0
      // writes it after its own vmcnt(0) above, so none of wave 0's clamped DMAs can land on it afterwards; the other waves' clamped
      // DMAs go to their own slots)
      unsigned* flag = reinterpret_cast<unsigned*>(smem + NSTAGE * STAGE_B);
      if (wave == 0 && lane_h == 0) {
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
      __syncthreads();  // everyone has read the flag word before wave 0's next clamped DMA can land on it
      if (is_last) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long long gp = g_first; gp <= g_last; ++gp) {
          const long long s = range_start(gp, a);
          const float* sl = a.ws + (2 * gp + (s > f ? 0 : 1)) * SLOT_FLOATS;
#pragma unroll
          for (int i = 0; i < NACC; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] += sl[((wave * NACC + i) * 4 + r) * 64 + lane_h];
          }
        }
        tile_epilogue<BM, !CONV, LinearRows<BM>, true, TERMS == 0>(P, b, bm, bn, acc, wave, lane);
      }
    }
    u = u_next;
    cur = nxt;
    LDC_STAMP(4 + 4 * seg_)
#ifdef LDC_GEMM_STAMPS
    if (seg_ < 2) ++seg_;
#endif
  }
  LDC_STAMP(15)
}

struct ConvParams {
  int H, W, cin, ks, kshift;
};

constexpr int LDC_SPLIT_GROUP = -100;  // internal: "launch this group's problems one by one" (gemm_v3_dispatch)

template <int BM, int TERMS, bool CONV = false>
int launch_v3(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace, long long workspace_bytes,
              void* stream, const ConvParams* cp = nullptr, bool may_split_group = false) {
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
    if (d.K % (TERMS == 1 ? 2 * BK : BK)) return LDC_ERR_UNSUPPORTED;  // a 128-byte k-step: 64 plain-bf16 / 32 split / 32 fp32 values
    if constexpr (TERMS == 0) {  // exact fp32: plain fp32 rows in, fp32 rows out, a contiguous [N][K] weight
      if (d.flags != 0 || d.ldw != d.K) return LDC_ERR_UNSUPPORTED;
    } else {
      if (!(d.flags & LDC_GEMM_A_SPLIT)) return LDC_ERR_UNSUPPORTED;
    }
    LDC_CHECK_ALIGN16(q.A);
    LDC_CHECK_ALIGN16(q.W);
    if constexpr (TERMS == 0) {
      if ((d.lda & 3) || (d.a_bs & 3)) return LDC_ERR_UNSUPPORTED;  // the caller falls back to the register-staged kernel
    } else {
      if ((d.lda & 7) || (d.a_bs & 7) || (reinterpret_cast<unsigned long long>(q.A) & 31ull)) return LDC_ERR_ALIGN;
    }
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
      P.c_split = (TERMS != 0 && (d.flags & LDC_GEMM_C_SPLIT)) ? (TERMS == 3 ? LDC_FMT_SPLIT : LDC_FMT_BF16) : 0;
      // operand rows out: N % 4 == 0 (v4), the pad half of a last half-filled 8-column group is zeroed
      if (P.c_split && !(v4 && d.ldc % 8 == 0 && d.ldc >= ((d.N + 7) & ~7) &&
                         d.c_bs % 8 == 0 && (reinterpret_cast<unsigned long long>(q.C) & 31ull) == 0))
        return LDC_ERR_ALIGN;
    }
    if (epi != nullptr && epi[i].heads > 0) {
      const ldc_qkv_epilogue& e = epi[i];
      auto al16 = [](const void* q_) { return (reinterpret_cast<unsigned long long>(q_) & 15ull) == 0; };
      if (d.N != 3 * e.heads * BN || d.act != LDC_ACT_NONE || q.gate != nullptr || q.R != nullptr) return LDC_ERR_ARG;
      if ((e.wq == nullptr) != (e.wk == nullptr) || e.reserved != nullptr || e.rope_row0 < 0) return LDC_ERR_ARG;
      if constexpr (TERMS == 0) {  // plain fp32 rows out
        if ((d.ldc & 3) || (d.c_bs & 3) || !al16(q.C)) return LDC_ERR_ALIGN;
      } else {
        if ((d.ldc & 7) || (d.c_bs & 7) || (reinterpret_cast<unsigned long long>(q.C) & 31ull)) return LDC_ERR_ALIGN;
      }
      if (!al16(e.wq) || !al16(e.wk) || !al16(e.rope) || (q.bias && !al16(q.bias))) return LDC_ERR_ALIGN;
      P.qkv_heads = e.heads;
      P.rope_row0 = e.rope_row0;
      P.qk_w[0] = e.wq;
      P.qk_w[1] = e.wk;
      P.rope_cs = e.rope;
      P.eps = e.eps;
      P.qscale = e.qscale != 0.f ? e.qscale : 0.08838834764831845f * 1.4426950408889634f;
    }
    if constexpr (CONV) {
      P.cH = cp->H; P.cW = cp->W; P.cin = cp->cin; P.ks = cp->ks; P.kshift = cp->kshift;
      // 16 zero bytes: the tail of the counter block (zeroed by ldc_gemm_grouped_workspace_init, written by nobody)
      P.zero16 = static_cast<const unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES - 64;
    }
    P.tm = ldc_cdiv(d.M, BM);
    P.tn = ldc_cdiv(d.N, BN);
    P.kt = d.K / (TERMS == 1 ? 2 * BK : BK);
    P.unit0 = U;
    P.tile0 = tiles;
    {
      // super-row height: an XCD's share is ~T8 = tiles / 8 consecutive tiles = rm row panels x T8 / rm column panels; per k-step
      // it pulls rm * BM + (T8 / rm) * BN operand rows through the fabric -> minimum at rm = sqrt(T8 * BN / BM); super-rows of
      // (nearly) equal height.  Only when the activation panel is what the one-super-row order over-fetches (every XCD streams all
      // of A): measured (profiles/r02_c_gemm_tile_order_ab.log) 2250 x 1536 x 7680 -3.4 % time / -31 % fabric
      // bytes and 18000 x 1536 x 7680 -10 %, but +3 % on 2250 x 4608 / 10752 x 1536, whose 14 MB A panel stays cached anyway.
      // LDC_BF16X3_RM (measurement aid, read per call): force it; 0 = one super-row
      P.rm = P.tm;
      // (a conv's taps re-read the same pixels: its panel is M x ldx, and neighbouring row tiles share their halo rows)
      const double a_bytes = CONV ? static_cast<double>(d.M) * d.lda * 4.0 : static_cast<double>(d.batch) * d.M * d.K * 4.0;
      if (a_bytes >= 48e6) {
        const double t8 = static_cast<double>(d.batch) * P.tm * P.tn / 8.0;
        const double want = sqrt(t8 * BN / BM);
        int nsr = static_cast<int>(P.tm / (want > 1.0 ? want : 1.0) + 0.5);
        if (nsr < 1) nsr = 1;
        if (nsr > P.tm) nsr = P.tm;
        P.rm = ldc_cdiv(P.tm, nsr);
      }
      if (const char* e = LDC_AB_GETENV("LDC_BF16X3_RM")) {
        const int f = atoi(e);
        if (f >= 0) P.rm = (f == 0 || f > P.tm) ? P.tm : f;
      }
    }
    const long long t = static_cast<long long>(d.batch) * P.tm * P.tn;
    tiles += t;
    U += t * P.kt;
  }
  if (tiles > LDC_GEMM_COUNTER_BYTES / 4 - 16) return LDC_ERR_UNSUPPORTED;  // in-launch reduction only; last 64 B: zero page
  const long long slot_bytes = SLOT_FLOATS * static_cast<long long>(sizeof(float));
  // grid size: (phase-aligned divisor of tiles * s when every problem has the same k-depth)
  long long G = CUS;
  {
    bool same_kt = true;
    for (int i = 1; i < n; ++i) same_kt = same_kt && (a.pr[i].kt == a.pr[0].kt);
    long long best = 0;
    if (same_kt) {
      const int kt = a.pr[0].kt;
      // more workgroups win unless they are bought with many more pieces per tile (a hand-off each): 2 % per extra split factor.
      // 288 tiles x 320 k-steps (the 1.6B model's single-block out projection): 192 ranges at s = 2, 240 at s = 5 - 279 -> 269 us
      double best_score = 0.0;
      for (int sfac = 1; sfac <= 8; ++sfac) {
        if (kt % sfac || (sfac > 1 && kt / sfac < 8)) continue;  // whole tiles (sfac 1) at any depth
        const long long items = tiles * sfac;
        long long gd = items < CUS ? items : CUS;
        while (gd > 1 && items % gd) --gd;
        const double score = static_cast<double>(gd) * (1.0 - 0.02 * (sfac - 1));
        if (score > best_score) {
          best_score = score;
          best = gd;
        }
      }
    }
    long long few = 0;  // few tiles (the 84-column output head: 15 tiles): aligned split-K over more workgroups
    if (same_kt && best < 160) {
      const int kt = a.pr[0].kt;
      for (int sfac = 8; sfac >= 2; --sfac)
        if (kt % sfac == 0 && kt / sfac >= 8 && tiles * sfac <= CUS) {
          few = tiles * sfac;
          break;
        }
    }
    // Round 4: a GROUP whose tile count has no k-aligned cut onto >= 160 workgroups (1.6B dual blocks: (15 + 4) x 16 = 304 tiles of
    // k-depth 2^j - every aligned grid is a multiple of 19 that divides 304 s: 152) used to take the unaligned ranges below, which cost
    // the split modes 40 % (out / down projection of the 1.6B dual block: 205 / 213 TFLOP/s).  Its problems one by one cut well
    // (pred: 240 tiles = 240 workgroups; cond: 64 tiles x 4 k-pieces = 256): the dispatcher launches them separately.
    if (may_split_group && TERMS != 0 && n > 1 && best < 160 && few == 0) return LDC_SPLIT_GROUP;
    if (best >= 160) {
      G = best;
    } else if (few > 0) {
      G = few;  // measured on 1800 x 84 x 1536: 24.7 us at the stream-K default (36 ranges), 18.0 at 90
    } else {
      const double kt_avg = static_cast<double>(U) / static_cast<double>(tiles);
      long long umin = static_cast<long long>(sqrt(kt_avg * 8.0) + 0.5);
      if (umin < 1) umin = 1;
      const long long gmax = U / umin > 0 ? U / umin : 1;
      if (gmax < G) G = gmax;
    }
  }
  if constexpr (TERMS == 0) {  // (the DCAE's fp32 convs too: 8.09 / 8.83 -> 7.73 / 8.41 ms per frame, profiles/r04_l_dcae_unit_ranges_ab.txt; the split modes lose 10 % with it)
    // exact fp32 (round 4): ALWAYS one unit range per CU, cut wherever U / 256 falls.  The fp32 k-step is matrix-pipe bound (5x the split
    // kernel's MFMA time per k-step), so the L2 lockstep that made unaligned ranges 40 % slower in the split mode does not matter here and
    // the chip is not at its power cap: idle CUs cost their full share.  375M in fp32 mode, same box: 111.3 -> 118.5 TFLOP/s per GEMM launch,
    // 2.82 -> 2.96 member-steps/s (aligned 240 / 248 ranges: 111.6 / 115.0; profiles/r04_k_fp32_unit_ranges.txt)
    if (U >= 4 * CUS) G = CUS;
  }
  static const char* const force_g = LDC_AB_GETENV("LDC_BF16X3_G");  // measurement aid (read once): force the number of unit ranges
  if (force_g) {
    const long long g_ = atoll(force_g);
    if (g_ > 0 && g_ <= CUS) G = g_;
  }
  if (U < G) G = U;
  if (workspace == nullptr) return LDC_ERR_ARG;
  if (workspace_bytes < LDC_GEMM_COUNTER_BYTES + 2 * slot_bytes) return LDC_ERR_ARG;
  {
    const long long fit = (workspace_bytes - LDC_GEMM_COUNTER_BYTES) / (2 * slot_bytes);
    if (fit < G) G = fit;
  }
  LDC_CHECK_ALIGN16(workspace);
  if (U >= (1LL << 31)) return LDC_ERR_UNSUPPORTED;  // 32-bit unit arithmetic in the kernel
#ifdef LDC_AB_BUILD
  {  // A/B build only: LDC_GEMM_DEBUG_G=1 prints every launch's cut (n problems, tiles, k-depth, tile rows, unit ranges, units per range)
    static const bool dbg = getenv("LDC_GEMM_DEBUG_G") != nullptr;
    if (dbg)
      fprintf(stderr, "ldc_gemm cut: terms %d conv %d n %d tiles %lld kt0 %d BM %d G %lld U %lld %s\n", TERMS, int(CONV), n, tiles, a.pr[0].kt, BM, G, U,
              (U % G == 0 && (U / G) % a.pr[0].kt == 0) ? "whole tiles" : (U % G == 0 ? "equal ranges" : "UNALIGNED"));
  }
#endif
  a.G = static_cast<int>(G);
  a.U = U;
  a.upg = static_cast<unsigned>(U / G);
  a.urem = static_cast<unsigned>(U % G);
  a.tiles = tiles;
  a.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES);
  a.counters = static_cast<unsigned*>(workspace);
  const size_t lds = NSTAGE * STAGE_B + 8 * 1024;  // ring + one 1 KiB dump slot per wave
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_v3_kernel<BM, TERMS, CONV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((gemm_bf16x3_v3_kernel<BM, TERMS, CONV>), dim3(a.G), dim3(512), lds, static_cast<hipStream_t>(stream), a);
  return ldc_launch_status();
}

}  // namespace

// returns LDC_ERR_UNSUPPORTED when a problem does not fit this kernel (caller falls back to the register-staged kernel of gemm_streamk.hip)
static int gemm_v3_dispatch(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace,
                            long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(problems);
  if (n <= 0 || n > MAXP) return LDC_ERR_ARG;
  long long tiles256 = 0;
  for (int i = 0; i < n; ++i) {
    const ldc_gemm_desc& d = problems[i].d;
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return LDC_ERR_ARG;
    if (!(d.flags & LDC_GEMM_A_SPLIT)) return LDC_ERR_UNSUPPORTED;
    if (((d.flags ^ problems[0].d.flags) & LDC_GEMM_BF16_1TERM) != 0) return LDC_ERR_ARG;  // one arithmetic per launch
    tiles256 += static_cast<long long>(d.batch) * ldc_cdiv(d.M, 256) * ldc_cdiv(d.N, BN);
  }
  // half-height tiles while 256-row tiles would not fill the chip twice over (both heights run 8 waves here, so the
  // half-height tile costs no MFMA efficiency, only twice the W traffic per FLOP); LDC_BF16X3_BM forces one
  static const char* const force_thr = LDC_AB_GETENV("LDC_BF16X3_SMALL_TILES");  // measurement aid, read once: the cross-over below
  bool small = tiles256 < (force_thr ? atoll(force_thr) : 400);  // measured cross-over on the 375M model launches (tools/gemm_sustained.py, both heights forced)
  static const char* const force_bm = LDC_AB_GETENV("LDC_BF16X3_BM");  // measurement aid, read once
  if (force_bm) small = (atoi(force_bm) == 128);
  const bool one = (problems[0].d.flags & LDC_GEMM_BF16_1TERM) != 0;
  auto launch = [&](const ldc_gemm_problem* pr, const ldc_qkv_epilogue* ep, int cnt, bool sm, bool may_split) {
    if (one)
      return sm ? launch_v3<128, 1>(pr, ep, cnt, workspace, workspace_bytes, stream, nullptr, may_split)
                : launch_v3<256, 1>(pr, ep, cnt, workspace, workspace_bytes, stream, nullptr, may_split);
    return sm ? launch_v3<128, 3>(pr, ep, cnt, workspace, workspace_bytes, stream, nullptr, may_split)
              : launch_v3<256, 3>(pr, ep, cnt, workspace, workspace_bytes, stream, nullptr, may_split);
  };
  const int st = launch(problems, epi, n, small, true);
  if (st != LDC_SPLIT_GROUP) return st;
  // (no partial output on a bad argument: launch_v3 returns LDC_SPLIT_GROUP only after it has validated EVERY problem of the group - the
  // per-problem checks run before the cut is chosen - and the workspace checks that follow it fail on the first single launch, before
  // anything has been queued)
  for (int i = 0; i < n; ++i) {  // the group has no good aligned cut: its problems one after the other (independent; same stream)
    const ldc_gemm_desc& d = problems[i].d;
    bool sm = static_cast<long long>(d.batch) * ldc_cdiv(d.M, 256) * ldc_cdiv(d.N, BN) < (force_thr ? atoll(force_thr) : 400);
    if (force_bm) sm = (atoi(force_bm) == 128);
    const int sti = launch(problems + i, epi ? epi + i : nullptr, 1, sm, false);
    if (sti != LDC_OK) return sti;
  }
  return LDC_OK;
}

// exact-fp32 grouped GEMM on the ring kernel (TERMS = 0); LDC_ERR_UNSUPPORTED -> the caller (ldc_gemm_grouped, gemm_streamk.hip) runs
// its register-staged stream-K kernel instead (K % 32 != 0, strided weights, unaligned rows)
static int gemm_f32_ring(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace, long long workspace_bytes,
                         void* stream) {
  LDC_CHECK_PTR(problems);
  if (n <= 0 || n > MAXP) return LDC_ERR_ARG;
  for (int i = 0; i < n; ++i)  // the caller's choice (include/ladcast_hip.h): this launch stays on the register-staged kernel
    if (problems[i].d.flags & LDC_GEMM_F32_REGSTAGE) return LDC_ERR_UNSUPPORTED;
  auto al16 = [](const void* q_) { return q_ != nullptr && (reinterpret_cast<unsigned long long>(q_) & 15ull) == 0; };
  // what the register-staged kernel accepts and this one does not goes there, not back to the caller as an error
  if (!al16(workspace) || workspace_bytes < LDC_GEMM_COUNTER_BYTES + 2ll * 256 * BN * static_cast<long long>(sizeof(float)))
    return LDC_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i) {
    const ldc_gemm_desc& d = problems[i].d;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.batch <= 0) return LDC_ERR_ARG;
    if (!al16(problems[i].A) || !al16(problems[i].W)) return LDC_ERR_UNSUPPORTED;
  }
  // 128-row tiles only: a tile is 2.7x the split kernel's MFMA time, so balance matters more than the halved W traffic per FLOP of
  // the 256-row tile (375M model in fp32 mode, same box: 113.6 TFLOP/s against 111.0 with 256 rows forced; profiles/r03_j_*)
#ifdef LDC_AB_BUILD
  static const char* const force_bm = getenv("LDC_F32_RING_BM");  // A/B build only: 256-row tiles for the exact-fp32 GEMMs
  if (force_bm && atoi(force_bm) == 256) return launch_v3<256, 0>(problems, epi, n, workspace, workspace_bytes, stream);
#endif
  return launch_v3<128, 0>(problems, epi, n, workspace, workspace_bytes, stream);
}

int ldc_gemm_grouped_f32_ring(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes, void* stream) {
  return gemm_f32_ring(problems, nullptr, n, workspace, workspace_bytes, stream);
}

// exact-fp32 QKV projection with per-head RMSNorm + rotary embedding as its epilogue, plain fp32 rows out (include/ladcast_hip.h);
// LDC_ERR_UNSUPPORTED (K % 32 != 0, strided weights, unaligned rows): run ldc_gemm_grouped and ldc_qk_rmsnorm_rope instead
extern "C" int ldc_gemm_grouped_qkv_f32(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace,
                                        long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(epi);
  return gemm_f32_ring(problems, epi, n, workspace, workspace_bytes, stream);
}

int ldc_gemm_grouped_bf16x3_v3(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                               void* stream) {
  return gemm_v3_dispatch(problems, nullptr, n, workspace, workspace_bytes, stream);
}

// QKV projection with the attention operand rows as output (include/ladcast_hip.h); only this kernel has that epilogue:
// LDC_ERR_UNSUPPORTED (K % 32 != 0, fp32 activations) means "run the plain GEMM and ldc_attn_qkv_prepare_split instead"
extern "C" int ldc_gemm_grouped_bf16x3_qkv(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace,
                                           long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(epi);
  return gemm_v3_dispatch(problems, epi, n, workspace, workspace_bytes, stream);
}

// SphereConv2d (dense, stride 1, k = 3 / 5; k = 1: pointwise conv / Linear over pixel rows) as an implicit GEMM on the pre-split kernel.
// in_fmt = LDC_FMT_SPLIT: X is NHWC split rows [B*H*W][ldx] (pad columns behind cin up to the next multiple of 8 zero), Wp =
// ldc_pack_weight_bf16x2 of the [cout][k*k][cin rounded up to 32 * 2^j] tap-major weight (zero behind cin) - the weight format of
// ldc_sphere_conv_nhwc_bf16x3; Y: fp32 rows or (out_fmt = LDC_FMT_SPLIT) split rows for the next conv.
// in_fmt = LDC_FMT_BF16 (the single-term `bf16` mode): X is plain bf16 rows (same row stride ldx in floats), Wp = ldc_pack_weight_bf16
// of the [cout][k*k][cin rounded up to 64 * 2^j] weight; Y: fp32 rows or (out_fmt = LDC_FMT_BF16) plain bf16 rows.
// in_fmt = LDC_FMT_F32 (round 4, the exact-fp32 mode): X is plain fp32 rows (ldx % 4 == 0, cin % 4 == 0), Wp the plain fp32
// [cout][k*k][cin rounded up to 32 * 2^j] tap-major weight (zeros behind cin), Y fp32 rows - the TERMS = 0 instantiation.
extern "C" int ldc_sphere_conv_nhwc_split(const float* X, const void* Wp, const float* bias, const float* R, float* Y, int B, int H,
                                          int W, int cin, int ldx, int cout, int ldy, int ldr, int ksize, int act, int in_fmt,
                                          int out_fmt, void* workspace, long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(X);
  LDC_CHECK_PTR(Wp);
  LDC_CHECK_PTR(Y);
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return LDC_ERR_ARG;
  if (ksize != 1 && ksize != 3 && ksize != 5) return LDC_ERR_UNSUPPORTED;
  if (ksize > 1 && ((W & 1) || H < 2 || H < ksize / 2 || W / 2 < ksize / 2 || H > 32767 || W > 65535)) return LDC_ERR_UNSUPPORTED;
  if (in_fmt != LDC_FMT_SPLIT && in_fmt != LDC_FMT_BF16 && in_fmt != LDC_FMT_F32) return LDC_ERR_UNSUPPORTED;
  if (out_fmt != LDC_FMT_F32 && out_fmt != in_fmt) return LDC_ERR_UNSUPPORTED;
  if (in_fmt == LDC_FMT_F32) {  // exact fp32 (round 4): plain fp32 pixel rows, 16-byte chunks of 4 channels
    if ((ldx & 3) || (cin & 3) || ldx < cin || ldy < cout) return LDC_ERR_ALIGN;
  } else if ((ldx & 7) || ldx < ((cin + 7) & ~7) || ldy < cout) {
    return LDC_ERR_ALIGN;
  }
  const long long M = static_cast<long long>(B) * H * W;
  if (M > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  const bool one = in_fmt == LDC_FMT_BF16;
  const int cpk = one ? 2 * BK : BK;  // channels per k-step
  ConvParams cp{H, W, cin, ksize, 0};
  int ktpt = ldc_cdiv(cin, cpk);  // k-steps per tap, rounded up to a power of two (the kernel shifts instead of dividing)
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
  q.d.K = ksize * ksize * ktpt * cpk;
  q.d.batch = 1;
  q.d.lda = ldx;
  q.d.ldw = q.d.K;
  q.d.ldc = ldy;
  q.d.ldr = ldr;
  q.d.act = act;
  q.d.flags = LDC_GEMM_A_SPLIT | (out_fmt != LDC_FMT_F32 ? LDC_GEMM_C_SPLIT : 0) | (one ? LDC_GEMM_BF16_1TERM : 0);
  if (in_fmt == LDC_FMT_F32) {
    // the exact-fp32 variant of the ring kernel (TERMS = 0: v_mfma_f32_16x16x4_f32 on plain fp32 operand rows) with the same conv gather;
    // 128-row tiles only, as for the fp32 GEMMs (a tile is 2.7x the split kernel's MFMA time: balance beats the halved W traffic)
    q.d.flags = 0;
    return launch_v3<128, 0, true>(&q, nullptr, 1, workspace, workspace_bytes, stream, &cp);
  }
  if (ksize == 3) {  // round 5: the halo-staged kernel (conv_halo.hip) where it serves the shape
    static const char* const halo_sw = LDC_AB_GETENV("LDC_CONV_HALO");  // measurement aid (A/B build): 0 = never
    if (!(halo_sw && atoi(halo_sw) == 0)) {
      if (workspace == nullptr || workspace_bytes < LDC_GEMM_COUNTER_BYTES) return LDC_ERR_ARG;
      const int hs = ldc_conv_halo_dispatch(X, Wp, bias, R, Y, B, H, W, cin, ldx, cout, ldy, ldr, act, in_fmt, out_fmt, workspace, workspace_bytes, stream);
      if (hs != LDC_ERR_UNSUPPORTED) return hs;
    }
  }
  const long long tiles256 = static_cast<long long>(ldc_cdiv(M, 256)) * ldc_cdiv(cout, BN);
  // tile height: measured cross-over (tools/conv_bench.py) - 252 -> 252 at 120 x 240 (226 tiles of 256 rows): 101 us at 256 rows, 121 at
  // 128; 504 -> 504 at 60 x 120 (116 tiles): 122 / 111; four frames of it (464): 350 / 437.  LDC_CONV_SMALL_TILES: measurement aid
  static const char* const force_thr = LDC_AB_GETENV("LDC_CONV_SMALL_TILES");
  const bool small = tiles256 < (force_thr ? atoll(force_thr) : 200);
  if (one)
    return small ? launch_v3<128, 1, true>(&q, nullptr, 1, workspace, workspace_bytes, stream, &cp)
                 : launch_v3<256, 1, true>(&q, nullptr, 1, workspace, workspace_bytes, stream, &cp);
  return small ? launch_v3<128, 3, true>(&q, nullptr, 1, workspace, workspace_bytes, stream, &cp)
               : launch_v3<256, 3, true>(&q, nullptr, 1, workspace, workspace_bytes, stream, &cp);
}
