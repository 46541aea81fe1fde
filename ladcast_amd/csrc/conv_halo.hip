// Dense 3x3 SphereConv2d (models/sphere_conv.py:62-129,174-192) as an implicit GEMM whose pixel operand is staged ONCE per channel
// chunk: round 5's replacement for the gathered-row conv of gemm_bf16x3_v3.hip (CONV = true) on the DCAE's large layers.
//
// Why (profiles/r05_a_conv_variants_nodma_and_chunk_major.log, r05_a_conv_pmc_gathered_vs_chunk_major.txt): in the gathered form every k-step (tap, 32-channel chunk) pulls its
// own BM x 128 B pixel panel through L2 into LDS - nine DMA'd copies of (nearly) the same pixels per chunk, 48 KB per k-step and CU with
// the weight tile, 571 MB of fabric traffic per 252 -> 252 full-resolution launch against 60 MB algorithmic.  The same kernel with the
// DMAs compiled out runs the launch in 78 us instead of 112; ordering the k-steps chunk-major so that L2 serves the re-reads cut the
// fabric bytes 5.5x and made the launch SLOWER (the per-k-step source arithmetic): what costs is the L2 -> LDS leg itself.
// Here a workgroup owns a TH x TW pixel tile (BM = TH TW) and one 128-column panel of the output.  For every channel chunk it DMAs the
// (TH + 2) x (TW + 2) HALO image of the tile once (1.33x the tile for 8 x 32) into one of two LDS buffers, with the sphere padding rule
// folded into the source addresses (rows past a pole mirror and roll by W / 2, columns wrap), and forms the A fragments of all nine taps
// from that image: tap (ky, kx) of tile pixel (ty, tx) is halo pixel (ty + ky, tx + kx) - at the two pole rows (tx + 2 - kx) for the
// kernel row that reaches over the pole, which is the reference's left-right flip of that kernel row.  Weights stream through the 3-stage
// ring exactly as in the GEMM (k-step = (chunk, tap): byte offset ((tap << kshift) + chunk) * 128 of the packed tap-major row), and the
// k-step body - fragment window, MFMA order, waits - is gemm_bf16x3_v3.hip's.  Per k-step and wave: 2 weight DMAs + 1 halo DMA (a dump
// slot when the chunk's pieces are out) instead of 6 / 4.
// Halo image in LDS: pixel p at byte p * 128; its 16-byte chunk c at slot c ^ f(p), f(p) = ((p >> 1) & 1) | (p & 4).  Found by search
// (tools/lds_swizzle_search.py): with this f the four 16-lane groups of a ds_read_b128 are conflict-free when lanes 0-15 / 16-31 /
// 32-47 / 48-63 read k-group 0 / 1 / 2 / 3 of ANY 16 consecutive pixels - every alignment, which the +-1 pixel tap shifts need (the
// GEMM's swizzle only serves runs that start at a multiple of 16).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

#include "gemm_v3_common.inc"

struct HaloArgs {
  DevProblem P;  // A = X operand rows, W = packed weight, bias / R / C, d.{M, N, lda, ldc, ldr, act}; cH, cW, cin, kshift, zero16
  int TW, tw_shift, TH, HP;  // tile (TH x TW pixels, TW a power of two), halo pitch TW + 2
  int NPX, NP;               // halo pixels (TH + 2) HP, DMA pieces of 8 pixels
  int hp_magic;              // j / HP == (j * hp_magic) >> 16 for j < 8 NP (checked by the host)
  int tiles_x, tiles_y, tn;  // pixel tiles per frame, column panels
  int n_chunks;              // channel chunks of CPK
  int ksplit;                // workgroups per tile: each takes a contiguous share of the channel chunks (1 = whole tiles)
  int G;                     // work items = frames * tiles_y * tiles_x * tn * ksplit = grid size
  float* ws;                 // ksplit > 1: one accumulator slab (256 x 128 floats) per work item
  unsigned* counters;        // ksplit > 1: one arrival counter per tile (zero between launches)
};

// tile row r -> output pixel row of Y (0x7fffffff: outside the image)
struct TileRows {
  int base, y0, x0, tw_shift, tw_mask, H, W;
  __device__ __forceinline__ int operator()(int, int r) const {
    const int y = y0 + (r >> tw_shift), x = x0 + (r & tw_mask);
    return (y < H && x < W) ? base + y * W + x : 0x7fffffff;
  }
};

__device__ __forceinline__ unsigned halo_swz(int p) { return static_cast<unsigned>(((p >> 1) & 1) | (p & 4)); }

template <int BM, int TERMS>
__global__ __launch_bounds__(512) void conv_halo_kernel(HaloArgs a) {
  constexpr int CPK = TERMS == 1 ? 2 * BK : BK;  // channels per 128-byte chunk (plain bf16 | split groups)
  constexpr int RT = BM / 128;
  constexpr int NACC = RT * 8;
  constexpr int WB = BN * ROW_B;  // one weight stage: 16 KiB
  constexpr int TAPS = 9;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const DevProblem& P = a.P;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  int g;
  {  // XCD-aware placement: an XCD gets a contiguous run of work items - the column panels of one pixel tile and the neighbouring
     // tiles (shared halo rows) meet in one L2
    const int bid = blockIdx.x, G = a.G;
    const int q = G >> 3, r = G & 7, xcd = bid & 7;
    g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int bn, y0, x0, frame, c0, c1, item;
  {
    const unsigned ks = static_cast<unsigned>(a.ksplit);
    const unsigned it_ = static_cast<unsigned>(g) / ks, piece = static_cast<unsigned>(g) - it_ * ks;
    item = __builtin_amdgcn_readfirstlane(static_cast<int>(it_));
    c0 = __builtin_amdgcn_readfirstlane(static_cast<int>(piece * static_cast<unsigned>(a.n_chunks) / ks));        // this workgroup's channel chunks
    c1 = __builtin_amdgcn_readfirstlane(static_cast<int>((piece + 1) * static_cast<unsigned>(a.n_chunks) / ks));
    const unsigned gu = it_, tn = static_cast<unsigned>(a.tn);
    unsigned t = gu / tn;
    bn = __builtin_amdgcn_readfirstlane(static_cast<int>(gu - t * tn));
    const unsigned tx_i = t % static_cast<unsigned>(a.tiles_x);
    t /= static_cast<unsigned>(a.tiles_x);
    const unsigned ty_i = t % static_cast<unsigned>(a.tiles_y);
    frame = __builtin_amdgcn_readfirstlane(static_cast<int>(t / static_cast<unsigned>(a.tiles_y)));
    y0 = __builtin_amdgcn_readfirstlane(static_cast<int>(ty_i) * a.TH);
    x0 = __builtin_amdgcn_readfirstlane(static_cast<int>(tx_i) * a.TW);
  }
  const int H = P.cH, W = P.cW, HP = a.HP;
  const int fbase = frame * H * W;
  const unsigned smem_lds = lds_addr(smem);
  const unsigned abuf_b = static_cast<unsigned>(a.NP) * 1024u;      // one halo buffer
  const unsigned abuf0 = NSTAGE * WB;                                // LDS offset of halo buffer 0
  unsigned char* const dump = smem + abuf0 + 2 * abuf_b + wave * 1024;  // per-wave landing zone of the DMAs that carry nothing

  const int lane = fresh_lane();
  const int fr = lane & 15, kg = lane >> 4;
  const int lr = lane >> 3, lp = lane & 7;
  // W fragment addresses: the GEMM's image (rows aligned to 16, swz)
  const unsigned f_r = swz(fr);
  const unsigned w_hi = smem_lds + fr * ROW_B + (((2 * kg) ^ f_r) << 4);
  const unsigned w_lo = smem_lds + fr * ROW_B + (((2 * kg + 1) ^ f_r) << 4);
  // this lane's tile pixels: row tile rt of the wave, fragment row fr
  int p0[RT], pole[RT];  // halo pixel of tap (0, 0); bit 0: the pixel is in the northern pole row (kernel row 0 flips), bit 2: southern (row 2)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = 16 * (RT * wave + rt) + fr;
    const int ty = r >> a.tw_shift, tx = r & (a.TW - 1);
    p0[rt] = ty * HP + tx;
    pole[rt] = ((y0 + ty == 0) ? 1 : 0) | ((y0 + ty == H - 1) ? 4 : 0);
  }
  // a wave whose rows all lie outside the image issues no MFMA (scalar: the wave's first row is its top-left pixel, TW >= 16)
  const bool wave_rows = (y0 + ((16 * RT * wave) >> a.tw_shift) < H) && (x0 + ((16 * RT * wave) & (a.TW - 1)) < W);
  const unsigned kg2 = 2u * static_cast<unsigned>(kg);
  // fragment address (hi chunk; lo = hi ^ 16) of tap (ky, kx) in the halo buffer at LDS offset ab
  auto a_addr = [&](int rt, int ky, int kx, unsigned ab) -> unsigned {
    const int d = 2 - 2 * kx;  // the pole rows read the mirrored tap column: kx -> 2 - kx (branch-free: the k-step body stays one block)
    const int flip = -((pole[rt] >> ky) & ((ky & 1) ^ 1));  // all ones: this kernel row reaches over this pixel's pole
    const int p = p0[rt] + ky * HP + kx + (flip & d);
    return smem_lds + ab + (static_cast<unsigned>(p) << 7) + ((kg2 ^ halo_swz(p)) << 4);
  };

  // weight DMA sources (chunk 0, tap 0): instruction i covers panel rows [8 (wave + 8 i), + 8)
  const long long w_row_bytes = static_cast<long long>(TAPS) * (CPK << P.kshift) * (TERMS == 1 ? 2 : 4);
  const unsigned char* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (wave + 8 * i) + lr;
    int gn = bn * BN + r;
    gn = gn < P.d.N ? gn : P.d.N - 1;
    w_src[i] = P.W + static_cast<long long>(gn) * w_row_bytes + ((lp ^ swz(r)) << 4);
  }
  // (tap, stage, i are literals at every call site: the main loop is unrolled over the nine taps of a chunk)
  auto issue_w = [&](int chunk_, int tap, int stage, int i) {
    const bool live = chunk_ < c1;
    const int chunk = live ? chunk_ : c1 - 1;
    const long long koff = static_cast<long long>((tap << P.kshift) + chunk) * (BK * 4);
#ifdef LDC_GEMM_DIAG_NODMA
    if (chunk >= 0) return;
#endif
    dma16(w_src[i] + koff, live ? smem + stage * WB + (wave + 8 * i) * 1024 : dump);
  };
  // halo pieces: piece q = wave + 8 slot covers halo pixels 8 q .. 8 q + 7; this lane's source pixel row of every slot is formed once
  // (sphere padding: rows past a pole mirror and roll by W / 2, columns wrap); the chunk slot it fetches lands at slot lp of the piece.
  // 64 slot = 0 mod 8 in bits 1, 2 of the halo pixel index, so the swizzle - and with it the channel group this lane fetches - is the
  // same for all of this lane's pieces
  constexpr int MAXSLOT = 7;  // pieces per wave and chunk: slot t is issued in tap t and has landed by the barrier of tap t + 2 <= 8
  const int cs = lp ^ static_cast<int>(halo_swz(8 * wave + lr));
  const int zthr = P.cin - 8 * (TERMS == 3 ? (cs >> 1) : cs);  // this lane's 8-column group holds channels while chunk * CPK < zthr
  const unsigned char* a_row[MAXSLOT];
  {
    const unsigned char* const Xb = reinterpret_cast<const unsigned char*>(P.A);
    const long long row_b = static_cast<long long>(P.d.lda) * 4;
#pragma unroll
    for (int s_ = 0; s_ < MAXSLOT; ++s_) {
      int j = 8 * (wave + 8 * s_) + lr;
      j = j < a.NPX ? j : a.NPX - 1;
      const int hy = (j * a.hp_magic) >> 16, hx = j - hy * HP;
      int r = y0 - 1 + hy, c = x0 - 1 + hx;
      const bool lo = r < 0, hi = r >= H;
      const int rm = lo ? -1 - r : 2 * H - 1 - r;
      r = (lo || hi) ? rm : r;
      c -= (lo || hi) ? (W >> 1) : 0;
      r = r < 0 ? 0 : r;  // (rows of a ragged tile far past the pole: never used)
      c += c < 0 ? W : 0;
      c -= c >= W ? W : 0;
      c = c < 0 ? 0 : (c >= W ? W - 1 : c);
      a_row[s_] = Xb + static_cast<long long>(fbase + r * W + c) * row_b + (cs << 4);
    }
  }
  auto issue_a = [&](int chunk_, int slot) {
    const int q = wave + 8 * slot;
    const bool live = slot < MAXSLOT && chunk_ < c1 && q < a.NP;
    const int chunk = chunk_ < c1 ? chunk_ : c1 - 1;
    const unsigned char* src = a_row[slot < MAXSLOT ? slot : 0] + chunk * (BK * 4);
    src = chunk * CPK >= zthr ? P.zero16 : src;  // 8-column groups behind cin read zeros (their weights are zero too)
#ifdef LDC_GEMM_DIAG_NODMA
    if (chunk >= 0) return;
#endif
    dma16(src, live ? smem + abuf0 + (chunk & 1) * abuf_b + q * 1024 : dump);
  };

  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4v wh0, wl0, wh1, wl1, wh2, wl2, wh3, wl3;  // W fragment window: 4 column tiles
  i32x4v ah0[RT], al0[RT], ah1[RT], al1[RT];      // A fragments of the current / next k-step

#define LDC_SB __builtin_amdgcn_sched_barrier(0)
  // W fragments of column tile CT in ring stage ST: one base register per chunk kind, everything else an immediate
#define LDC_RD_W(WH, WL, ST, CT)                                 \
  {                                                              \
    LDC_DS_READ(WH, w_hi, (ST) * WB + (CT) * (16 * ROW_B));      \
    LDC_DS_READ(WL, w_lo, (ST) * WB + (CT) * (16 * ROW_B));      \
  }
  // A fragments of tap (KY, KX) from the halo buffer at LDS offset AB
#define LDC_RD_A(AH, AL, KY, KX, AB)                                    \
  {                                                                     \
    _Pragma("unroll") for (int rt_ = 0; rt_ < RT; ++rt_) {              \
      const unsigned ad_ = a_addr(rt_, KY, KX, AB);                     \
      LDC_DS_READ(AH[rt_], ad_, 0);                                     \
      LDC_DS_READ(AL[rt_], ad_ ^ 16u, 0);                               \
    }                                                                   \
  }
#define LDC_MM(ACC, WF, AF)                                                                                                    \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WF), __builtin_bit_cast(bf16x8, AF), ACC, 0, 0, 0); \
  LDC_SB;
#define LDC_CT(CT, WH, WL, AH, AL)                 \
  if constexpr (WR) {                              \
    if constexpr (TERMS == 1) {                    \
      LDC_MM(acc[(CT)], WH, AH[0])                 \
      if constexpr (RT == 2) {                     \
        LDC_MM(acc[8 + (CT)], WH, AH[RT - 1])      \
      }                                            \
      LDC_MM(acc[(CT)], WL, AL[0])                 \
      if constexpr (RT == 2) {                     \
        LDC_MM(acc[8 + (CT)], WL, AL[RT - 1])      \
      }                                            \
    } else if constexpr (RT == 2) {                \
      LDC_MM(acc[(CT)], WH, AL[0])                 \
      LDC_MM(acc[8 + (CT)], WH, AL[RT - 1])        \
      LDC_MM(acc[(CT)], WL, AH[0])                 \
      LDC_MM(acc[8 + (CT)], WL, AH[RT - 1])        \
      LDC_MM(acc[(CT)], WH, AH[0])                 \
      LDC_MM(acc[8 + (CT)], WH, AH[RT - 1])        \
    } else {                                       \
      LDC_MM(acc[(CT)], WH, AL[0])                 \
      LDC_MM(acc[(CT)], WL, AH[0])                 \
      LDC_MM(acc[(CT)], WH, AH[0])                 \
    }                                              \
  }
#define LDC_WAIT(X, Y)                                     \
  asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(X), "+v"(Y)); \
  LDC_SB;
  // One k-step = tap TAP (a literal) of chunk ch: (AH, AL) current A fragments, (AHN, ALN) receive those of the next k-step.  Nine k-steps
  // per chunk and a three-stage weight ring: tap t always sits in stage t % 3, so every weight fragment address is base + immediate.
  // DMAs: before the barrier the second weight piece of k-step kt + 2; behind it the first of kt + 3 (into this k-step's stage, free
  // once every wave is past the barrier) and halo piece TAP of chunk ch + 1 - whose buffer was last read for the fragments of k-step
  // (ch - 1, tap 8), waited for before the barrier one k-step ago.  vmcnt(3) at the barrier leaves exactly those three in flight.
#define LDC_KSTEP(TAP, AH, AL, AHN, ALN)                                                                             \
  {                                                                                                                  \
    constexpr int st_ = (TAP) % 3, st1_ = ((TAP) + 1) % 3, st2_ = ((TAP) + 2) % 3;                                   \
    constexpr int tn_ = ((TAP) + 1) % TAPS, cn_ = ((TAP) + 1) / TAPS;                                                \
    LDC_CT(0, wh0, wl0, AH, AL)                                                                                      \
    LDC_RD_W(wh0, wl0, st_, 4)                                                                                       \
    issue_w(ch + ((TAP) + 2) / TAPS, ((TAP) + 2) % TAPS, st2_, 1);                                                   \
    LDC_WAIT(wh1, wl1)                                                                                               \
    LDC_CT(1, wh1, wl1, AH, AL)                                                                                      \
    LDC_RD_W(wh1, wl1, st_, 5)                                                                                       \
    LDC_WAIT(wh2, wl2)                                                                                               \
    LDC_CT(2, wh2, wl2, AH, AL)                                                                                      \
    LDC_RD_W(wh2, wl2, st_, 6)                                                                                       \
    LDC_WAIT(wh3, wl3)                                                                                               \
    LDC_CT(3, wh3, wl3, AH, AL)                                                                                      \
    LDC_RD_W(wh3, wl3, st_, 7)                                                                                       \
    LDC_WAIT(wh0, wl0)                                                                                               \
    LDC_CT(4, wh0, wl0, AH, AL)                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh1), "+v"(wl1), "+v"(wh2), "+v"(wl2), "+v"(wh3), "+v"(wl3));         \
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                    \
    LDC_SB;                                                                                                          \
    LDC_RD_A(AHN, ALN, tn_ / 3, tn_ % 3, abuf0 + ((ch + cn_) & 1) * abuf_b)                                          \
    LDC_RD_W(wh0, wl0, st1_, 0)                                                                                      \
    issue_w(ch + ((TAP) + 3) / TAPS, ((TAP) + 3) % TAPS, st_, 0);                                                    \
    LDC_SB;                                                                                                          \
    LDC_CT(5, wh1, wl1, AH, AL)                                                                                      \
    LDC_RD_W(wh1, wl1, st1_, 1)                                                                                      \
    issue_a(ch + 1, (TAP));                                                                                          \
    LDC_SB;                                                                                                          \
    LDC_CT(6, wh2, wl2, AH, AL)                                                                                      \
    LDC_RD_W(wh2, wl2, st1_, 2)                                                                                      \
    LDC_SB;                                                                                                          \
    LDC_CT(7, wh3, wl3, AH, AL)                                                                                      \
    LDC_RD_W(wh3, wl3, st1_, 3)                                                                                      \
    LDC_SB;                                                                                                          \
  }
  // the k-step's opening wait: the A fragments it consumes and the first weight column tile have landed
#define LDC_OPEN(AH, AL)                                                                                             \
  if constexpr (RT == 2) {                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(AH[0]), "+v"(AL[0]), "+v"(AH[1]), "+v"(AL[1]), "+v"(wh0), "+v"(wl0)); \
  } else {                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(AH[0]), "+v"(AL[0]), "+v"(wh0), "+v"(wl0));                           \
  }                                                                                                                  \
  LDC_SB;

  // prologue: the whole halo of chunk 0, the weights of taps 0 and 1; then (behind the barrier) the post-barrier DMAs of "k-step -1"
#pragma unroll
  for (int s_ = 0; s_ < MAXSLOT; ++s_) issue_a(c0, s_);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_w(c0, 0, 0, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_w(c0, 1, 1, i);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  LDC_RD_A(ah1, al1, 0, 0, abuf0 + (c0 & 1) * abuf_b)  // tap 0 finds its fragments in set 1, where tap 8 of the previous chunk leaves them
  LDC_RD_W(wh0, wl0, 0, 0)
  LDC_RD_W(wh1, wl1, 0, 1)
  LDC_RD_W(wh2, wl2, 0, 2)
  LDC_RD_W(wh3, wl3, 0, 3)
  issue_w(c0, 2, 2, 0);
  issue_a(c1, 0);  // (a dump piece: keeps the barrier's vmcnt(3) exact in k-step 0)
  LDC_SB;

  // the loop in two copies, with and without MFMAs (a wave whose rows all lie outside the image): left to the compiler's loop unswitching the
  // branch stayed inside the k-step and every join copied the accumulator file
  auto main_loop = [&](auto wr_) {
    constexpr bool WR = decltype(wr_)::value;
    for (int ch = c0; ch < c1; ++ch) {
      LDC_OPEN(ah1, al1)
#pragma unroll
      for (int rt_ = 0; rt_ < RT; ++rt_) {  // nine taps alternate between the two fragment sets: an odd count, so one copy per chunk
        ah0[rt_] = ah1[rt_];
        al0[rt_] = al1[rt_];
      }
      LDC_KSTEP(0, ah0, al0, ah1, al1)
      LDC_OPEN(ah1, al1)
      LDC_KSTEP(1, ah1, al1, ah0, al0)
      LDC_OPEN(ah0, al0)
      LDC_KSTEP(2, ah0, al0, ah1, al1)
      LDC_OPEN(ah1, al1)
      LDC_KSTEP(3, ah1, al1, ah0, al0)
      LDC_OPEN(ah0, al0)
      LDC_KSTEP(4, ah0, al0, ah1, al1)
      LDC_OPEN(ah1, al1)
      LDC_KSTEP(5, ah1, al1, ah0, al0)
      LDC_OPEN(ah0, al0)
      LDC_KSTEP(6, ah0, al0, ah1, al1)
      LDC_OPEN(ah1, al1)
      LDC_KSTEP(7, ah1, al1, ah0, al0)
      LDC_OPEN(ah0, al0)
      LDC_KSTEP(8, ah0, al0, ah1, al1)
    }
  };
  if (wave_rows) main_loop(std::true_type{});
  else main_loop(std::false_type{});
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wh0), "+v"(wl0), "+v"(wh1), "+v"(wl1), "+v"(wh2), "+v"(wl2), "+v"(wh3), "+v"(wl3));
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) asm volatile("" : "+v"(ah0[rt]), "+v"(al0[rt]), "+v"(ah1[rt]), "+v"(al1[rt]));
#undef LDC_OPEN
#undef LDC_KSTEP
#undef LDC_WAIT
#undef LDC_CT
#undef LDC_MM
#undef LDC_RD_A
#undef LDC_RD_W
#undef LDC_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped DMAs of the last k-steps land in this wave's dump slot: before the LDS goes back

  if (a.ksplit > 1) {
    // the pieces of a tile meet through write-through slabs and ONE ticket per workgroup; the piece whose ticket is the last re-reads all
    // slabs of the tile in piece order (bitwise reproducible whatever the arrival order) and applies the epilogue - gemm_bf16x3_v3.hip's
    // hand-off, with the channel-chunk share as the k range
    constexpr int SLOT_FLOATS = BM * BN;
    const int lane_h = fresh_lane();
    float* slot = a.ws + static_cast<long long>(g) * SLOT_FLOATS + lane_h;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __hip_atomic_store(slot + ((wave * NACC + i) * 4 + r) * 64, acc[i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned* cnt = a.counters + item;
    unsigned* flag = reinterpret_cast<unsigned*>(smem);  // the ring is idle (every DMA of this workgroup has landed)
    if (wave == 0 && lane_h == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned last = (ticket == static_cast<unsigned>(a.ksplit) - 1) ? 1u : 0u;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    const unsigned is_last = *flag;
    if (!is_last) return;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < a.ksplit; ++p) {
      const float* sl = a.ws + (static_cast<long long>(item) * a.ksplit + p) * SLOT_FLOATS;
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] += sl[((wave * NACC + i) * 4 + r) * 64 + lane_h];
      }
    }
  }
  TileRows rows{fbase, y0, x0, a.tw_shift, a.TW - 1, H, W};
  tile_epilogue<BM, false, TileRows, false>(P, 0, 0, bn, acc, wave, lane, rows);
}

template <int BM, int TERMS>
int launch_halo(const HaloArgs& a, size_t lds, hipStream_t st) {
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<BM, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((conv_halo_kernel<BM, TERMS>), dim3(a.G), dim3(512), lds, st, a);
  return ldc_launch_status();
}

}  // namespace

// Tile choice.  Tiles are 256 pixels (TH x TW, TW a power of two >= 16, at most seven halo pieces per wave and chunk) and a launch is whole
// tiles, one per workgroup, in ceil(tiles / 256) rounds.  Measured against the gathered conv on the DCAE's layer shapes, same box
// (profiles/r05_d_conv_halo_static_taps_ab.log, r05_f_conv_ksplit_ab.log): 252 -> 252 at 120 x 240 (240 tiles) 99 -> 80 us, 504 -> 1008 at
// 60 x 120 (256 tiles) 182 -> 145, eight frames of every 3 x 3 layer 20 - 27 % less.  A launch that cannot give most CUs a tile
// (504 -> 504 at 30 x 60 in one frame: 32 tiles) stays on the gathered kernel, which cuts along K over all of them; between 96 and 128
// tiles two workgroups share a tile (below).  (A 128-pixel tile of this kernel was slower than the gathered conv on every shape - the
// weight stream per FLOP doubles - and is not built: r05_c_conv_tile_sweep.log.)
// Among the tile shapes the one with the fewest tiles wins, then the smallest halo.
static bool halo_plan(int B, int H, int W, int cout, int n_chunks, int* bm_out, int* tw_out, int* ksplit_out) {
  constexpr int bm = 256;
  const int tn = ldc_cdiv(cout, BN);
  int best_tw = 0, best_np = 0;
  long long best_tiles = 0;
  static const char* const force_tw = LDC_AB_GETENV("LDC_CONV_HALO_TW");  // measurement aids (A/B build only)
  for (int tw = 16; tw <= bm; tw *= 2) {
    if (force_tw && atoi(force_tw) != tw) continue;
    const int th = bm / tw;
    if (th > H || tw > W) continue;
    const int np = ldc_cdiv((th + 2) * (tw + 2), 8);
    if (np > 7 * 8) continue;  // seven pieces per wave and chunk
    if (static_cast<long long>(NSTAGE) * BN * ROW_B + 2LL * np * 1024 + 8 * 1024 > 160 * 1024) continue;
    const long long tiles = static_cast<long long>(B) * ldc_cdiv(H, th) * ldc_cdiv(W, tw) * tn;
    if (best_tw == 0 || tiles < best_tiles || (tiles == best_tiles && np < best_np)) {
      best_tw = tw;
      best_np = np;
      best_tiles = tiles;
    }
  }
  static const char* const min_tiles = LDC_AB_GETENV("LDC_CONV_HALO_MIN_TILES");
  static const char* const force_ks = LDC_AB_GETENV("LDC_CONV_HALO_KSPLIT");
  if (best_tw == 0) return false;
  int ks = 1;
  if (best_tiles < (min_tiles ? atoll(min_tiles) : 192)) {
    // too few tiles for the chip: TWO workgroups per tile, each with half of the channel chunks, summed in the launch by the last arriver
    // (504 -> 504 at 60 x 120, one frame: 128 tiles x 2 - 102.6 -> 86.5 us against the gathered kernel, profiles/r05_f_conv_ksplit_ab.log).
    // Three pieces already lost to the gathered kernel (two frames of 504 -> 504 at 30 x 60: 58 -> 63 us), and more would make one
    // workgroup re-read megabytes of slabs: such launches stay on the gathered kernel, which cuts along K over all CUs.
    ks = 0;
    for (int k = 2; k <= 2; ++k)
      if (best_tiles * k >= 192 && best_tiles * k <= 256 && n_chunks / k >= 2) {
        ks = k;
        break;
      }
    if (force_ks && atoi(force_ks) == 1) ks = 0;
    if (ks == 0) return false;
  }
  // useful share of the rounds: a launch just past a multiple of 256 tiles (257 -> two rounds) is better cut along K
  const long long items = best_tiles * ks, rounds = (items + 255) / 256;
  if (static_cast<double>(items) / (256.0 * rounds) < 0.7) return false;
  *bm_out = bm;
  *tw_out = best_tw;
  *ksplit_out = ks;
  return true;
}

// which kernel ldc_sphere_conv_nhwc_split runs for a shape (include/ladcast_hip.h): 1 = the halo-staged kernel, with its tile
extern "C" int ldc_sphere_conv_plan(int B, int H, int W, int cin, int cout, int ksize, int in_fmt, int* tile_rows, int* tile_w) {
  int bm = 0, tw = 0, ks = 1;
  const int n_chunks = ldc_cdiv(cin > 0 ? cin : 1, in_fmt == LDC_FMT_BF16 ? 2 * BK : BK);
  static const char* const halo_sw = LDC_AB_GETENV("LDC_CONV_HALO");
  const bool halo = ksize == 3 && (in_fmt == LDC_FMT_SPLIT || in_fmt == LDC_FMT_BF16) && B > 0 && H > 0 && W > 0 && cout > 0 &&
                    !(halo_sw && atoi(halo_sw) == 0) && cin > 0 && halo_plan(B, H, W, cout, n_chunks, &bm, &tw, &ks);
  if (tile_rows) *tile_rows = halo ? bm : 0;
  if (tile_w) *tile_w = halo ? tw : 0;
  return halo ? 1 : 0;
}

// Launch.  Returns LDC_ERR_UNSUPPORTED when the halo kernel does not serve the shape (the caller then runs the gathered conv of
// gemm_bf16x3_v3.hip): exact-fp32 rows, images too small to fill tiles.
int ldc_conv_halo_dispatch(const float* X, const void* Wp, const float* bias, const float* R, float* Y, int B, int H, int W, int cin,
                           int ldx, int cout, int ldy, int ldr, int act, int in_fmt, int out_fmt, void* workspace, long long workspace_bytes,
                           void* stream) {
  if (in_fmt != LDC_FMT_SPLIT && in_fmt != LDC_FMT_BF16) return LDC_ERR_UNSUPPORTED;
  const bool one = in_fmt == LDC_FMT_BF16;
  const int cpk = one ? 2 * BK : BK;
  const long long M = static_cast<long long>(B) * H * W;
  if (M * ldx * 4 >= (1LL << 40) || M > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  const int tn = ldc_cdiv(cout, BN);
  int best_bm = 0, best_tw = 0, ksplit = 1;
  if (!halo_plan(B, H, W, cout, ldc_cdiv(cin, cpk), &best_bm, &best_tw, &ksplit)) return LDC_ERR_UNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < LDC_GEMM_COUNTER_BYTES) return LDC_ERR_ARG;
  // the argument checks of the gathered kernel's launcher (launch_v3, gemm_bf16x3_v3.hip), with its error codes: this path issues 16-byte
  // global -> LDS DMAs of X and Wp and takes `act` as a table index, so a misaligned view or a bad activation must not get past here
  LDC_CHECK_ALIGN16(Wp);
  LDC_CHECK_ALIGN16(workspace);
  if ((ldx & 7) || (reinterpret_cast<unsigned long long>(X) & 31ull)) return LDC_ERR_ALIGN;
  if (act < LDC_ACT_NONE || act > LDC_ACT_RELU) return LDC_ERR_UNSUPPORTED;
  if (R != nullptr && ldr < cout) return LDC_ERR_ARG;
  HaloArgs a{};
  DevProblem& P = a.P;
  P.A = X;
  P.W = static_cast<const unsigned char*>(Wp);
  P.bias = bias;
  P.R = R;
  P.C = Y;
  P.d.M = static_cast<int>(M);
  P.d.N = cout;
  P.d.batch = 1;
  P.d.lda = ldx;
  P.d.ldc = ldy;
  P.d.ldr = ldr;
  P.d.act = act;
  {
    auto al16 = [](const void* q_) { return (reinterpret_cast<unsigned long long>(q_) & 15ull) == 0; };
    bool v4 = (cout % 4 == 0) && (ldy % 4 == 0) && al16(Y);
    if (bias) v4 = v4 && al16(bias);
    if (R) v4 = v4 && al16(R) && (ldr % 4 == 0);
    P.vec4 = v4 ? 1 : 0;
    P.c_split = out_fmt != LDC_FMT_F32 ? (one ? LDC_FMT_BF16 : LDC_FMT_SPLIT) : 0;
    if (P.c_split && !(v4 && ldy % 8 == 0 && ldy >= ((cout + 7) & ~7) && (reinterpret_cast<unsigned long long>(Y) & 31ull) == 0)) return LDC_ERR_ALIGN;
  }
  P.cH = H;
  P.cW = W;
  P.cin = cin;
  P.ks = 3;
  int ktpt = ldc_cdiv(cin, cpk);
  while ((1 << P.kshift) < ktpt) ++P.kshift;
  P.zero16 = static_cast<const unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES - 64;  // 64 zero bytes behind the counters (written by nobody)
  a.n_chunks = ktpt;  // chunks behind ceil(cin / cpk) hold only zero weights: not visited (the gathered conv walks all 2^kshift)
  a.TW = best_tw;
  a.TH = best_bm / best_tw;
  while ((1 << a.tw_shift) < a.TW) ++a.tw_shift;
  a.HP = a.TW + 2;
  a.NPX = (a.TH + 2) * a.HP;
  a.NP = ldc_cdiv(a.NPX, 8);
  a.hp_magic = 65536 / a.HP + 1;
  for (int j = 0; j < 8 * a.NP; ++j)
    if (((j * a.hp_magic) >> 16) != j / a.HP) return LDC_ERR_UNSUPPORTED;  // (never for HP >= 18 and j < 512)
  a.tiles_x = ldc_cdiv(W, a.TW);
  a.tiles_y = ldc_cdiv(H, a.TH);
  a.tn = tn;
  const long long tiles = static_cast<long long>(B) * a.tiles_y * a.tiles_x * tn;
  const long long G = tiles * ksplit;
  if (G > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  a.G = static_cast<int>(G);
  a.ksplit = ksplit;
  if (ksplit > 1) {  // the GEMM's workspace: [counter block, zero between launches | slabs]
    const long long slab = 256LL * BN * static_cast<long long>(sizeof(float));
    if (tiles > LDC_GEMM_COUNTER_BYTES / 4 - 16 || workspace_bytes < LDC_GEMM_COUNTER_BYTES + G * slab) return LDC_ERR_UNSUPPORTED;
    a.counters = static_cast<unsigned*>(workspace);
    a.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + LDC_GEMM_COUNTER_BYTES);
  }
  const size_t lds = static_cast<size_t>(NSTAGE) * BN * ROW_B + 2u * a.NP * 1024u + 8u * 1024u;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  return one ? launch_halo<256, 1>(a, lds, st) : launch_halo<256, 3>(a, lds, st);
}
