// Split-bf16 ("bf16x3") attention, second generation: operands are split ONCE per attention call by a pack
// kernel (which also applies the per-head q/k RMSNorm and the rotary embedding), and the attention kernel brings
// whole pre-arranged K / V^T tiles into LDS by LDS-DMA (global_load_lds_dwordx4).
//
// Why: in attn_bf16x3.hip every query block re-reads raw fp32 K and V and re-splits them (S/128 = 18 times per
// head at S = 2250): per 32-key tile and wave that is ~112 VALU instructions, 20 global loads (16 of them dword
// gathers for the V transpose) and 8 LDS stores next to 48 MFMAs -- the kernel measured 11 VALU instructions per
// MFMA and 28 % MFMA busy.  Here a tile costs 8 DMA instructions per wave and no VALU; what is left beside the
// MFMAs is the softmax and the split of P.
//
// Packed operand buffer (ldc_attn_packed_bytes), nt = ceil(S / 32) key tiles, Sp = 32 nt:
//   Qp [B][H][Sp]  512 B per query: [hi: 16 chunks x 16 B | lo: 16 chunks], chunk = 8 consecutive head-dim values,
//                  pre-scaled by log2(e) / sqrt(128);
//   Kp [B][H][nt]  17 KiB per tile = the LDS image: 32 rows (keys) x 528 B = [hi 256 B | lo 256 B | 16 B pad] (pitch
//                  132 dwords: conflict-free ds_read_b128 of one chunk column over 32 rows, and every fragment address
//                  is lane base + immediate), padded to 17 DMA pieces of 1 KiB;
//   Vp [B][H][nt]  18 KiB per tile = the LDS image of V^T: 128 rows (head-dim d) x 144 B = [hi 64 B | lo 64 B | 16 B pad],
//                  chunk 2t+h holds keys 16t + 8(j>>2) + 4h + (j&3), j = 0..7 (the order in which a 32x32 accumulator's
//                  registers hold keys, so P needs no shuffle) = 18 DMA pieces.
//   Rows / keys past S are zero.
// Attention workgroup = 8 waves = 128 queries of one (batch, head); waves 0-3 sweep the first half of the key
// tiles, waves 4-7 the second half (two independent 2-stage rings of 32 KiB stages), merged through LDS at the end
// (as attn_bf16x3.hip).  Per iteration: vmcnt(0) + barrier (tile t landed, everyone is done with tile t-1) ->
// issue the DMA of tile t+1 -> S^T = K.Q^T (24 MFMAs) -> online softmax -> O^T += V^T.P^T (24 MFMAs).
// Fragment reads are untracked inline-asm ds_read_b128 with counted lgkmcnt waits (a compiler-visible LDS read would
// make hipcc drain vmcnt(0), i.e. wait for the next tile's DMA, in front of it), four fragment pairs in flight.
#include "common.h"

namespace {

constexpr int HD = 128;
constexpr int QB = 128;
constexpr int KT = 32;
constexpr int QTILE_B = 16384;           // 32 queries x 512 B
constexpr int KPITCH = 528, KIMG_B = KT * KPITCH;      // 16896 B of image ...
constexpr int KTILE_B = 17 * 1024;                     // ... in 17 DMA pieces
constexpr int VPITCH = 144, VTILE_B = HD * VPITCH;     // 18432 B = 18 DMA pieces
constexpr int GROUP_LDS = 2 * KTILE_B + 2 * VTILE_B;   // per key-range group: 2-stage K ring + 2-stage V^T ring = 70 KiB

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

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
// 8 fp32 -> hi / lo packed bf16 (16 B each)
__device__ __forceinline__ void split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7,
                                       uint4& h, uint4& l) {
  float r0, r1, r2, r3, r4, r5, r6, r7;
  h.x = split_pair(x0, x1, r0, r1);
  h.y = split_pair(x2, x3, r2, r3);
  h.z = split_pair(x4, x5, r4, r5);
  h.w = split_pair(x6, x7, r6, r7);
  l.x = pack_pair(r0, r1);
  l.y = pack_pair(r2, r3);
  l.z = pack_pair(r4, r5);
  l.w = pack_pair(r6, r7);
}

// ------------------------------------------------------------------------------------------------------------
// pack: [RMSNorm(128) * weight -> adjacent-pair RoPE] on q and k (same operation order as qk_rmsnorm_rope_kernel),
// split, and lay out as above.  One workgroup per (32-row tile, head, batch).
// ------------------------------------------------------------------------------------------------------------
struct PackArgs {
  const float* Q;
  const float* K;
  const float* V;
  int S, H, ld;
  long long bs;
  int split_row;  // rows [0, split_row): segment 0, rows [split_row, S): segment 1
  const float* wq[2];
  const float* wk[2];
  const float* cs[2];
  const float* sn[2];
  float eps, qscale;
  unsigned char* Qp;
  unsigned char* Kp;
  unsigned char* Vp;
  int nt;
};

__device__ __forceinline__ void norm_rope16(float (&x)[16], const float* __restrict__ w, const float* __restrict__ cs,
                                            const float* __restrict__ sn, int trow, int d0, float eps) {
  if (w) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) ss += x[i] * x[i] + x[i + 1] * x[i + 1];
    // the 8 threads of a row are 8 consecutive lanes
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = x[i] * r * w[d0 + i];
  }
  if (cs) {
    const float* c = cs + static_cast<long long>(trow) * 128 + d0;
    const float* s = sn + static_cast<long long>(trow) * 128 + d0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const float ox = x[i] * c[i] + (-x[i + 1]) * s[i];
      const float oy = x[i + 1] * c[i + 1] + x[i] * s[i + 1];
      x[i] = ox;
      x[i + 1] = oy;
    }
  }
}

__global__ __launch_bounds__(256) void attn_pack_kernel(PackArgs p) {
  // the two tile images are assembled in LDS and leave as linear 1 KiB-per-wave-instruction copies
  __shared__ __attribute__((aligned(16))) float vs[KT][HD + 4];
  __shared__ __attribute__((aligned(16))) unsigned char kimg[KIMG_B];
  __shared__ __attribute__((aligned(16))) unsigned char vimg[VTILE_B];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int r = tid >> 3, sg = tid & 7;
  const int row = t * KT + r;
  const bool ok = row < p.S;
  const int seg = row < p.split_row ? 0 : 1;
  const int trow = ok ? (seg ? row - p.split_row : row) : 0;  // rows past S: any valid table row (their output is zeroed)
  const long long src = static_cast<long long>(b) * p.bs + static_cast<long long>(ok ? row : 0) * p.ld + head * HD + 16 * sg;
  const long long bh = static_cast<long long>(b) * p.H + head;

  // every load of the thread is issued before the first use
  float xq[16], xk[16];
  float4 xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a = reinterpret_cast<const float4*>(p.Q + src)[i];
    xq[4 * i] = a.x; xq[4 * i + 1] = a.y; xq[4 * i + 2] = a.z; xq[4 * i + 3] = a.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a = reinterpret_cast<const float4*>(p.K + src)[i];
    xk[4 * i] = a.x; xk[4 * i + 1] = a.y; xk[4 * i + 2] = a.z; xk[4 * i + 3] = a.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) xv[i] = reinterpret_cast<const float4*>(p.V + src)[i];

  // ---- v: raw tile to LDS for the transpose ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 a = xv[i];
    if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&vs[r][16 * sg + 4 * i]) = a;
  }
  // ---- q: rows are contiguous 512 B in Qp ----
  norm_rope16(xq, p.wq[seg], p.cs[seg], p.sn[seg], trow, 16 * sg, p.eps);
  {
    const float sc = ok ? p.qscale : 0.f;
    unsigned char* dst = p.Qp + (bh * (static_cast<long long>(p.nt) * KT) + row) * 512;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      uint4 h, l;
      split8(xq[8 * e] * sc, xq[8 * e + 1] * sc, xq[8 * e + 2] * sc, xq[8 * e + 3] * sc, xq[8 * e + 4] * sc,
             xq[8 * e + 5] * sc, xq[8 * e + 6] * sc, xq[8 * e + 7] * sc, h, l);
      *reinterpret_cast<uint4*>(dst + (2 * sg + e) * 16) = h;
      *reinterpret_cast<uint4*>(dst + 256 + (2 * sg + e) * 16) = l;
    }
  }
  // ---- k -> LDS image ----
  norm_rope16(xk, p.wk[seg], p.cs[seg], p.sn[seg], trow, 16 * sg, p.eps);
  {
    const float m = ok ? 1.f : 0.f;
    unsigned char* dst = kimg + r * KPITCH;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      uint4 h, l;
      split8(xk[8 * e] * m, xk[8 * e + 1] * m, xk[8 * e + 2] * m, xk[8 * e + 3] * m, xk[8 * e + 4] * m, xk[8 * e + 5] * m,
             xk[8 * e + 6] * m, xk[8 * e + 7] * m, h, l);
      *reinterpret_cast<uint4*>(dst + (2 * sg + e) * 16) = h;
      *reinterpret_cast<uint4*>(dst + 256 + (2 * sg + e) * 16) = l;
    }
    if (sg == 0) *reinterpret_cast<uint4*>(dst + 512) = make_uint4(0u, 0u, 0u, 0u);  // pad
  }
  __syncthreads();
  // ---- v^T -> LDS image ----
  {
    const int d = tid & 127, g = tid >> 7;
    unsigned char* dst = vimg + d * VPITCH;
    if (g == 0) *reinterpret_cast<uint4*>(dst + 128) = make_uint4(0u, 0u, 0u, 0u);  // pad
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vg = g + 2 * i, tt = vg >> 1, hh = vg & 1;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = vs[16 * tt + 8 * (j >> 2) + 4 * hh + (j & 3)][d];
      uint4 h, l;
      split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], h, l);
      *reinterpret_cast<uint4*>(dst + vg * 16) = h;
      *reinterpret_cast<uint4*>(dst + 64 + vg * 16) = l;
    }
  }
  __syncthreads();
  {
    uint4* kd = reinterpret_cast<uint4*>(p.Kp + (bh * p.nt + t) * KTILE_B);
    uint4* vd = reinterpret_cast<uint4*>(p.Vp + (bh * p.nt + t) * VTILE_B);
    for (int i = tid; i < KIMG_B / 16; i += 256) kd[i] = reinterpret_cast<const uint4*>(kimg)[i];  // the piece padding past
    for (int i = tid; i < VTILE_B / 16; i += 256) vd[i] = reinterpret_cast<const uint4*>(vimg)[i];  // the image is never read
  }
}

// ------------------------------------------------------------------------------------------------------------
// attention on packed operands
// ------------------------------------------------------------------------------------------------------------
struct AttnArgs {
  const unsigned char* Qp;
  const unsigned char* Kp;
  const unsigned char* Vp;
  float* O;
  int S, H, ldo;
  long long o_bs;
  int nq, nt;
  int out_split;  // O in the split activation format of LDC_GEMM_A_SPLIT
};

#define LDC_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<unsigned long long>(p));
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// combine with the other 32-lane half (v_permlane32_swap: no LDS crossbar op, so no lgkmcnt(0) drain of the
// fragment reads in flight)
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_add(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#define LDC_SB __builtin_amdgcn_sched_barrier(0)
// fragment addresses are lane base + immediate: K step st at 32 st (+256 for lo); V pair J = 4 tt + dd at
// 32 dd rows x 144 B + 32 tt (+64 for lo)
#define LDC_DS_READ_I(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// TERMS = 3: split-bf16 products (lo.hi + hi.lo + hi.hi); TERMS = 1 (the single-term "bf16" mixed-precision mode): hi.hi only -
// the lo halves of the tile images are staged but never read, P is rounded to bf16 once
#define LDC_RD_K(FH, FL, ST)                                                     \
  {                                                                              \
    LDC_DS_READ_I(FH, ka, 32 * (ST));                                            \
    if constexpr (TERMS == 3) LDC_DS_READ_I(FL, ka, 32 * (ST) + 256);            \
  }
#define LDC_RD_V(FH, FL, J)                                                      \
  {                                                                              \
    LDC_DS_READ_I(FH, va, ((J) & 3) * (32 * VPITCH) + 32 * ((J) >> 2));          \
    if constexpr (TERMS == 3) LDC_DS_READ_I(FL, va, ((J) & 3) * (32 * VPITCH) + 32 * ((J) >> 2) + 64); \
  }
// wait until at most 3 / 2 / 1 / 0 fragment (pairs) are outstanding
#define LDC_W6(FH, FL)                                                           \
  if constexpr (TERMS == 3) {                                                    \
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(FH), "+v"(FL));                   \
  } else {                                                                       \
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(FH));                             \
  }
#define LDC_WN(N, FH, FL)                                                        \
  if constexpr (TERMS == 3) {                                                    \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(FH), "+v"(FL));              \
  } else if constexpr (N == 4) {                                                 \
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(FH));                             \
  } else if constexpr (N == 2) {                                                 \
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(FH));                             \
  } else {                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FH));                             \
  }
// one pair of probabilities -> hi / lo bf16 pair I of the P fragment (6 VALU, anchored where it is written)
#define LDC_SPLIT_PAIR(PH, PL, I, A, B)                                          \
  {                                                                              \
    unsigned u_ = pack_pair(A, B);                                               \
    asm volatile("" : "+v"(u_));                                                 \
    PH[I] = static_cast<int>(u_);                                                \
    if constexpr (TERMS == 3) {                                                  \
      const float t0_ = __uint_as_float(u_ << 16), t1_ = __uint_as_float(u_ & 0xffff0000u); \
      unsigned l_ = pack_pair((A) - t0_, (B) - t1_);                             \
      asm volatile("" : "+v"(l_));                                               \
      PL[I] = static_cast<int>(l_);                                              \
    }                                                                            \
  }
// an MFMA of a lo term: only in the split mode
#define LDC_MFX(DST, A, B, C) if constexpr (TERMS == 3) { DST = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0); }

template <int NGRP, int TERMS>
__global__ __launch_bounds__(256 * NGRP, NGRP == 1 ? 2 : 1) void attn_fwd_packed_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int grp = NGRP == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;  // key-range group 0 / 1 (wave-uniform)
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int S = p.S;
  int head, b, qblk;
  {  // XCD-aware placement as attn_bf16x3.hip: contiguous runs of [batch][head][query block] per XCD
    const int nq = p.nq, T = gridDim.x;
    const int bid = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = bid & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    qblk = lin % nq;
    const int hb = lin / nq;
    head = hb % p.H;
    b = hb / p.H;
  }
  const int q0 = qblk * QB + wave * 32;
  const long long bh = static_cast<long long>(b) * p.H + head;
  const int nt = p.nt;

  // Q fragments: step s, element j <-> d = 16 s + 8 half + j = chunk 2 s + half  
  bf16x8 qh[8], ql[8];
  {
    int qrow = q0 + l31;
    qrow = qrow < nt * KT ? qrow : nt * KT - 1;  // rows in [S, Sp) are zero in Qp; rows past Sp only exist in the last block
    const unsigned char* qp = p.Qp + (bh * (static_cast<long long>(nt) * KT) + qrow) * 512 + 16 * half;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      qh[s] = *reinterpret_cast<const bf16x8*>(qp + 32 * s);
      if constexpr (TERMS == 3) ql[s] = *reinterpret_cast<const bf16x8*>(qp + 256 + 32 * s);
    }
  }

  // per group: K ring (2 x 16 KiB) then V^T ring (2 x 16 KiB); tile t_begin + j lives in stage j & 1 of each
  unsigned char* const ring = smem + grp * GROUP_LDS;
  const unsigned ring_lds = lds_addr(ring);
  const unsigned char* const Kt0 = p.Kp + bh * nt * KTILE_B;  // wave-uniform bases; the lane adds 16 * lane
  const unsigned char* const Vt0 = p.Vp + bh * nt * VTILE_B;
  const unsigned lane_off = lane * 16;
  // K tile = 17 pieces of 1 KiB, V^T tile = 18; wave w of the group issues pieces w, w+4, ... (wave-uniform counts)
  auto issue_k = [&](int t, int stage) {
    const unsigned char* src = Kt0 + static_cast<long long>(t) * KTILE_B + wave * 1024;
    unsigned char* dst = ring + stage * KTILE_B + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(src + j * 4096 + lane_off, dst + j * 4096);
    if (wave == 0) dma16(src + 4 * 4096 + lane_off, dst + 4 * 4096);
  };
  auto issue_v = [&](int t, int stage) {
    const unsigned char* src = Vt0 + static_cast<long long>(t) * VTILE_B + wave * 1024;
    unsigned char* dst = ring + 2 * KTILE_B + stage * VTILE_B + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(src + j * 4096 + lane_off, dst + j * 4096);
    if (wave < 2) dma16(src + 4 * 4096 + lane_off, dst + 4 * 4096);
  };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
  float m_run = -1.0e30f;
  float l_run = 0.f;

  const int nhalf = NGRP == 2 ? (nt + 1) >> 1 : nt;  // iterations of the longer group
  const int t_begin = grp ? nhalf : 0;
  const int t_end = grp ? nt : nhalf;                // group 1 may have one tile fewer (or none)

  // per-lane fragment base inside a stage
  const unsigned k_row = l31 * KPITCH + 16 * half;
  const unsigned v_row = 2 * KTILE_B + l31 * VPITCH + 16 * half;
  i32x4v fh0, fl0, fh1, fl1, fh2, fl2, fh3, fl3;    // four fragment pairs in flight
  if constexpr (TERMS == 1) fl0 = fl1 = fl2 = fl3 = i32x4v{0, 0, 0, 0};  // never loaded in the single-term mode

  // ---- prologue: S of the group's first tile ----
  f32x16 sA, sB;
  if (t_begin < t_end) issue_k(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t_begin < t_end) {
    if (t_begin + 1 < t_end) issue_k(t_begin + 1, 1);
    issue_v(t_begin, 0);
    const unsigned ka = ring_lds + k_row;
    f32x16 sx;
    LDC_RD_K(fh0, fl0, 0)
    LDC_RD_K(fh1, fl1, 1)
    LDC_RD_K(fh2, fl2, 2)
    LDC_RD_K(fh3, fl3, 3)
#define LDC_QK(FH, FL, ST, C)                                                                               \
  if constexpr (TERMS == 3) {                                                                               \
    sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FL), qh[ST], C, 0, 0, 0);       \
    sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FH), ql[ST], sx, 0, 0, 0);      \
    sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FH), qh[ST], sx, 0, 0, 0);      \
  } else {                                                                                                  \
    sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FH), qh[ST], C, 0, 0, 0);       \
  }
    LDC_W6(fh0, fl0);
    LDC_QK(fh0, fl0, 0, zero16)
    LDC_RD_K(fh0, fl0, 4)
    LDC_W6(fh1, fl1);
    LDC_QK(fh1, fl1, 1, sx)
    LDC_RD_K(fh1, fl1, 5)
    LDC_W6(fh2, fl2);
    LDC_QK(fh2, fl2, 2, sx)
    LDC_RD_K(fh2, fl2, 6)
    LDC_W6(fh3, fl3);
    LDC_QK(fh3, fl3, 3, sx)
    LDC_RD_K(fh3, fl3, 7)
    LDC_W6(fh0, fl0);
    LDC_QK(fh0, fl0, 4, sx)
    LDC_WN(4, fh1, fl1);
    LDC_QK(fh1, fl1, 5, sx)
    LDC_WN(2, fh2, fl2);
    LDC_QK(fh2, fl2, 6, sx)
    LDC_WN(0, fh3, fl3);
    LDC_QK(fh3, fl3, 7, sx)
#undef LDC_QK
    sA = sx;
  }

  // ---- one iteration: tile t = t_begin + it.  S_cur = S of tile t (from the previous iteration), S_next = S of t+1 ----
  auto iteration = [&](int it, f32x16& scv, f32x16& sxv) __attribute__((always_inline)) {
    const int t = t_begin + it;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K_{t+1} and V_t landed; everyone is done with K_t and V_{t-1}
    if (t >= t_end) return;
    // the DMA instructions of K_{t+2} (into K_t's stage) and V_{t+1} (into V_{t-1}'s stage) sit one per MFMA gap in
    // the second half of phase A instead of as a burst behind the barrier
    const bool kq = t + 2 < t_end, vq = t + 1 < t_end;
    const unsigned char* const ksrc = Kt0 + static_cast<long long>(t + 2) * KTILE_B + wave * 1024 + lane_off;
    const unsigned char* const vsrc = Vt0 + static_cast<long long>(t + 1) * VTILE_B + wave * 1024 + lane_off;
    unsigned char* const kdst = ring + (it & 1) * KTILE_B + wave * 1024;
    unsigned char* const vdst = ring + 2 * KTILE_B + ((it + 1) & 1) * VTILE_B + wave * 1024;
    auto dma_k = [&](int j) __attribute__((always_inline)) { dma16(ksrc + j * 4096, kdst + j * 4096); };
    auto dma_v = [&](int j) __attribute__((always_inline)) { dma16(vsrc + j * 4096, vdst + j * 4096); };
    const unsigned ka = ring_lds + ((it + 1) & 1) * KTILE_B + k_row;  // K_{t+1} (stale data in the last iteration: S_next is then never used)
    const unsigned va = ring_lds + (it & 1) * VTILE_B + v_row;        // V_t
    float sc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = scv[r];
    if (t == nt - 1) {  // keys past S
      const int key_base = t * KT + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key_base + (r & 3) + 8 * (r >> 2);
        if (key >= S) sc[r] = -1.0e30f;
      }
    }
    f32x16 sx;
    float m_new, alpha;
    float r0, r1, r2, r3, r4, r5, r6, r7;
    i32x4v ph0, pl0, ph1, pl1;
    if constexpr (TERMS == 1) pl0 = pl1 = i32x4v{0, 0, 0, 0};
      // ---- phase A: S_next = K_{t+1} . Q^T (24 MFMAs) with the softmax of S_cur in the MFMA gaps ----
      LDC_RD_K(fh0, fl0, 0)
      LDC_RD_K(fh1, fl1, 1)
      LDC_RD_K(fh2, fl2, 2)
      LDC_RD_K(fh3, fl3, 3)
      LDC_SB;
      {
        float m0 = max3f(sc[0], sc[1], sc[2]), m1 = max3f(sc[3], sc[4], sc[5]), m2 = max3f(sc[6], sc[7], sc[8]), m3 = max3f(sc[9], sc[10], sc[11]);
        const float m4 = max3f(sc[12], sc[13], sc[14]);
        m0 = max3f(m0, m1, m2); m3 = max3f(m3, m4, sc[15]); m0 = fmaxf(m0, m3);
        m0 = xor32_max(m0);
        // lazy running max: it only moves when the tile's max exceeds it by more than 2^8 (scores are in log2 units), so exp2(s - m)
        // stays <= 256 and the 64-multiply rescale of O below really is rare; with the exact max some of a wave's 32 queries move it
        // in nearly every tile.  Softmax does not depend on which reference value is subtracted.
        m_new = (m0 - m_run > 8.0f) ? m0 : m_run;
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        asm volatile("" : "+v"(m_new), "+v"(alpha));
      }
      LDC_SB;
      LDC_W6(fh0, fl0); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl0), qh[0], zero16) LDC_SB;
      sc[0] = __builtin_amdgcn_exp2f(sc[0] - m_new); sc[1] = __builtin_amdgcn_exp2f(sc[1] - m_new); asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh0), ql[0], sx) LDC_SB;
      sc[2] = __builtin_amdgcn_exp2f(sc[2] - m_new); sc[3] = __builtin_amdgcn_exp2f(sc[3] - m_new); asm volatile("" : "+v"(sc[2]), "+v"(sc[3]));
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh0), qh[0], TERMS == 3 ? sx : zero16, 0, 0, 0); LDC_SB;
      LDC_RD_K(fh0, fl0, 4)
      sc[4] = __builtin_amdgcn_exp2f(sc[4] - m_new); sc[5] = __builtin_amdgcn_exp2f(sc[5] - m_new); asm volatile("" : "+v"(sc[4]), "+v"(sc[5]));
      LDC_SB;
      LDC_W6(fh1, fl1); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl1), qh[1], sx) LDC_SB;
      sc[6] = __builtin_amdgcn_exp2f(sc[6] - m_new); sc[7] = __builtin_amdgcn_exp2f(sc[7] - m_new); asm volatile("" : "+v"(sc[6]), "+v"(sc[7]));
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh1), ql[1], sx) LDC_SB;
      sc[8] = __builtin_amdgcn_exp2f(sc[8] - m_new); sc[9] = __builtin_amdgcn_exp2f(sc[9] - m_new); asm volatile("" : "+v"(sc[8]), "+v"(sc[9]));
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh1), qh[1], sx, 0, 0, 0); LDC_SB;
      LDC_RD_K(fh1, fl1, 5)
      sc[10] = __builtin_amdgcn_exp2f(sc[10] - m_new); sc[11] = __builtin_amdgcn_exp2f(sc[11] - m_new); asm volatile("" : "+v"(sc[10]), "+v"(sc[11]));
      LDC_SB;
      LDC_W6(fh2, fl2); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl2), qh[2], sx) LDC_SB;
      sc[12] = __builtin_amdgcn_exp2f(sc[12] - m_new); sc[13] = __builtin_amdgcn_exp2f(sc[13] - m_new); asm volatile("" : "+v"(sc[12]), "+v"(sc[13]));
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh2), ql[2], sx) LDC_SB;
      sc[14] = __builtin_amdgcn_exp2f(sc[14] - m_new); sc[15] = __builtin_amdgcn_exp2f(sc[15] - m_new); asm volatile("" : "+v"(sc[14]), "+v"(sc[15]));
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh2), qh[2], sx, 0, 0, 0); LDC_SB;
      LDC_RD_K(fh2, fl2, 6)
      r0 = sc[0] + sc[1]; r1 = sc[2] + sc[3]; r2 = sc[4] + sc[5]; r3 = sc[6] + sc[7]; asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
      LDC_SB;
      LDC_W6(fh3, fl3); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl3), qh[3], sx) LDC_SB;
      r4 = sc[8] + sc[9]; r5 = sc[10] + sc[11]; r6 = sc[12] + sc[13]; r7 = sc[14] + sc[15]; asm volatile("" : "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh3), ql[3], sx) LDC_SB;
      r0 += r1; r2 += r3; r4 += r5; r6 += r7; asm volatile("" : "+v"(r0), "+v"(r2), "+v"(r4), "+v"(r6));
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh3), qh[3], sx, 0, 0, 0); LDC_SB;
      LDC_RD_K(fh3, fl3, 7)
      r0 += r2; r4 += r6; r0 += r4; r0 = xor32_add(r0); l_run = l_run * alpha + r0; m_run = m_new; asm volatile("" : "+v"(l_run));
      LDC_SB;
      LDC_W6(fh0, fl0); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl0), qh[4], sx) LDC_SB;
      LDC_SPLIT_PAIR(ph0, pl0, 0, sc[0], sc[1])
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh0), ql[4], sx) LDC_SB;
      LDC_SPLIT_PAIR(ph0, pl0, 1, sc[2], sc[3])
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh0), qh[4], sx, 0, 0, 0); LDC_SB;
      LDC_RD_V(fh0, fl0, 0)
      LDC_SPLIT_PAIR(ph0, pl0, 2, sc[4], sc[5])
      LDC_SB;
      LDC_W6(fh1, fl1); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl1), qh[5], sx) LDC_SB;
      LDC_SPLIT_PAIR(ph0, pl0, 3, sc[6], sc[7])
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh1), ql[5], sx) LDC_SB;
      if (kq) dma_k(0);
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh1), qh[5], sx, 0, 0, 0); LDC_SB;
      LDC_RD_V(fh1, fl1, 1)
      if (kq) dma_k(1);
      LDC_SB;
      LDC_W6(fh2, fl2); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl2), qh[6], sx) LDC_SB;
      if (kq) dma_k(2);
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh2), ql[6], sx) LDC_SB;
      if (kq) dma_k(3);
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh2), qh[6], sx, 0, 0, 0); LDC_SB;
      LDC_RD_V(fh2, fl2, 2)
      if (vq) dma_v(0);
      LDC_SB;
      LDC_W6(fh3, fl3); LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fl3), qh[7], sx) LDC_SB;
      if (vq) dma_v(1);
      LDC_SB;
      LDC_MFX(sx, __builtin_bit_cast(bf16x8, fh3), ql[7], sx) LDC_SB;
      if (vq) dma_v(2);
      LDC_SB;
      sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh3), qh[7], sx, 0, 0, 0); LDC_SB;
      LDC_RD_V(fh3, fl3, 3)
      if (vq) { dma_v(3); if (wave < 2) dma_v(4); } if (kq && wave == 0) dma_k(4);
      LDC_SB;
      // ---- rare: the running max moved -> rescale O ----
      if (!__all(alpha == 1.0f)) {
      #pragma unroll
        for (int d = 0; d < 4; ++d)
      #pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
      LDC_SB;
      // ---- phase B: O^T += V_t^T . P^T (24 MFMAs); the second half of P is split in the gaps of the first ----
      LDC_W6(fh0, fl0); LDC_SB;
      LDC_MFX(o[0], __builtin_bit_cast(bf16x8, fl0), __builtin_bit_cast(bf16x8, ph0), o[0]) LDC_SB;
      LDC_SPLIT_PAIR(ph1, pl1, 0, sc[8], sc[9])
      LDC_SB;
      LDC_MFX(o[0], __builtin_bit_cast(bf16x8, fh0), __builtin_bit_cast(bf16x8, pl0), o[0]) LDC_SB;
      LDC_SB;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh0), __builtin_bit_cast(bf16x8, ph0), o[0], 0, 0, 0); LDC_SB;
      LDC_RD_V(fh0, fl0, 4)
      LDC_SPLIT_PAIR(ph1, pl1, 1, sc[10], sc[11])
      LDC_SB;
      LDC_W6(fh1, fl1); LDC_SB;
      LDC_MFX(o[1], __builtin_bit_cast(bf16x8, fl1), __builtin_bit_cast(bf16x8, ph0), o[1]) LDC_SB;
      LDC_SB;
      LDC_MFX(o[1], __builtin_bit_cast(bf16x8, fh1), __builtin_bit_cast(bf16x8, pl0), o[1]) LDC_SB;
      LDC_SPLIT_PAIR(ph1, pl1, 2, sc[12], sc[13])
      LDC_SB;
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh1), __builtin_bit_cast(bf16x8, ph0), o[1], 0, 0, 0); LDC_SB;
      LDC_RD_V(fh1, fl1, 5)
      LDC_SB;
      LDC_W6(fh2, fl2); LDC_SB;
      LDC_MFX(o[2], __builtin_bit_cast(bf16x8, fl2), __builtin_bit_cast(bf16x8, ph0), o[2]) LDC_SB;
      LDC_SPLIT_PAIR(ph1, pl1, 3, sc[14], sc[15])
      LDC_SB;
      LDC_MFX(o[2], __builtin_bit_cast(bf16x8, fh2), __builtin_bit_cast(bf16x8, pl0), o[2]) LDC_SB;
      LDC_SB;
      o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh2), __builtin_bit_cast(bf16x8, ph0), o[2], 0, 0, 0); LDC_SB;
      LDC_RD_V(fh2, fl2, 6)
      LDC_SB;
      LDC_W6(fh3, fl3); LDC_SB;
      LDC_MFX(o[3], __builtin_bit_cast(bf16x8, fl3), __builtin_bit_cast(bf16x8, ph0), o[3]) LDC_SB;
      LDC_SB;
      LDC_MFX(o[3], __builtin_bit_cast(bf16x8, fh3), __builtin_bit_cast(bf16x8, pl0), o[3]) LDC_SB;
      LDC_SB;
      o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh3), __builtin_bit_cast(bf16x8, ph0), o[3], 0, 0, 0); LDC_SB;
      LDC_RD_V(fh3, fl3, 7)
      LDC_SB;
      LDC_W6(fh0, fl0); LDC_SB;
      LDC_MFX(o[0], __builtin_bit_cast(bf16x8, fl0), __builtin_bit_cast(bf16x8, ph1), o[0]) LDC_SB;
      LDC_SB;
      LDC_MFX(o[0], __builtin_bit_cast(bf16x8, fh0), __builtin_bit_cast(bf16x8, pl1), o[0]) LDC_SB;
      LDC_SB;
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh0), __builtin_bit_cast(bf16x8, ph1), o[0], 0, 0, 0); LDC_SB;
      LDC_SB;
      LDC_WN(4, fh1, fl1); LDC_SB;
      LDC_MFX(o[1], __builtin_bit_cast(bf16x8, fl1), __builtin_bit_cast(bf16x8, ph1), o[1]) LDC_SB;
      LDC_SB;
      LDC_MFX(o[1], __builtin_bit_cast(bf16x8, fh1), __builtin_bit_cast(bf16x8, pl1), o[1]) LDC_SB;
      LDC_SB;
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh1), __builtin_bit_cast(bf16x8, ph1), o[1], 0, 0, 0); LDC_SB;
      LDC_SB;
      LDC_WN(2, fh2, fl2); LDC_SB;
      LDC_MFX(o[2], __builtin_bit_cast(bf16x8, fl2), __builtin_bit_cast(bf16x8, ph1), o[2]) LDC_SB;
      LDC_SB;
      LDC_MFX(o[2], __builtin_bit_cast(bf16x8, fh2), __builtin_bit_cast(bf16x8, pl1), o[2]) LDC_SB;
      LDC_SB;
      o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh2), __builtin_bit_cast(bf16x8, ph1), o[2], 0, 0, 0); LDC_SB;
      LDC_SB;
      LDC_WN(0, fh3, fl3); LDC_SB;
      LDC_MFX(o[3], __builtin_bit_cast(bf16x8, fl3), __builtin_bit_cast(bf16x8, ph1), o[3]) LDC_SB;
      LDC_SB;
      LDC_MFX(o[3], __builtin_bit_cast(bf16x8, fh3), __builtin_bit_cast(bf16x8, pl1), o[3]) LDC_SB;
      LDC_SB;
      o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh3), __builtin_bit_cast(bf16x8, ph1), o[3], 0, 0, 0); LDC_SB;
      LDC_SB;

    sxv = sx;
  };

  for (int it = 0; it < nhalf; it += 2) {
    iteration(it, sA, sB);
    if (it + 1 < nhalf) iteration(it + 1, sB, sA);
  }
  __syncthreads();  // everyone is done reading the rings

  // ---- merge the two key halves: group 1 hands (m, l, O) to group 0 through LDS ----
  if constexpr (NGRP == 2) {
    float* xch = reinterpret_cast<float*>(smem) + (wave * 66) * 64 + lane;  // [wave][66][64 lanes]
    if (grp == 1) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 64] = o[d][r];
      xch[64 * 64] = m_run;
      xch[65 * 64] = l_run;
    }
    __syncthreads();
    if (grp == 1) return;
    {
      const float m1 = xch[64 * 64], l1 = xch[65 * 64];
      const float m = fmaxf(m_run, m1);
      const float a0 = exp2f(m_run - m), a1 = exp2f(m1 - m);
      l_run = l_run * a0 + l1 * a1;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * a0 + xch[(d * 16 + r) * 64] * a1;
    }
  }

  const int qrow = q0 + l31;
  if (qrow < S) {
    const float inv = 1.0f / l_run;
    float* op = p.O + static_cast<long long>(b) * p.o_bs + static_cast<long long>(qrow) * p.ldo + head * HD + 4 * half;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        if (p.out_split) {  // this lane has half (4 half .. 4 half + 3) of the 8-column group 32 d + 8 g
          float r0, r1, r2, r3;
          uint2 hi, lo;
          hi.x = split_pair(v.x, v.y, r0, r1);
          hi.y = split_pair(v.z, v.w, r2, r3);
          lo.x = pack_pair(r0, r1);
          lo.y = pack_pair(r2, r3);
          unsigned char* grp = reinterpret_cast<unsigned char*>(op - 4 * half + 32 * d + 8 * g) + 8 * half;
          *reinterpret_cast<uint2*>(grp) = hi;
          *reinterpret_cast<uint2*>(grp + 16) = lo;
        } else {
          *reinterpret_cast<float4*>(op + 32 * d + 8 * g) = v;
        }
      }
  }
}

}  // namespace

extern "C" long long ldc_attn_packed_bytes(int B, int S, int H) {
  if (B <= 0 || S <= 0 || H <= 0) return 0;
  return static_cast<long long>(B) * H * ldc_cdiv(S, KT) * (QTILE_B + KTILE_B + VTILE_B);
}

extern "C" int ldc_attn_pack_bf16x3(const float* Q, const float* K, const float* V, int B, int S, int H, int ld_qkv,
                                    long long qkv_bs, int split_row, const float* wq0, const float* wk0, const float* cos0,
                                    const float* sin0, const float* wq1, const float* wk1, const float* cos1,
                                    const float* sin1, float eps, void* packed, void* stream) {
  LDC_CHECK_PTR(Q);
  LDC_CHECK_PTR(K);
  LDC_CHECK_PTR(V);
  LDC_CHECK_PTR(packed);
  if (B <= 0 || S <= 0 || H <= 0 || split_row < 0) return LDC_ERR_ARG;
  if ((wq0 == nullptr) != (wk0 == nullptr) || (wq1 == nullptr) != (wk1 == nullptr)) return LDC_ERR_ARG;
  if ((cos0 == nullptr) != (sin0 == nullptr) || (cos1 == nullptr) != (sin1 == nullptr)) return LDC_ERR_ARG;
  LDC_CHECK_ALIGN16(Q);
  LDC_CHECK_ALIGN16(K);
  LDC_CHECK_ALIGN16(V);
  LDC_CHECK_ALIGN16(packed);
  if ((ld_qkv & 3) || (qkv_bs & 3)) return LDC_ERR_ALIGN;
  if (H > 65535 || B > 65535) return LDC_ERR_UNSUPPORTED;
  PackArgs p{};
  p.Q = Q; p.K = K; p.V = V;
  p.S = S; p.H = H; p.ld = ld_qkv; p.bs = qkv_bs;
  p.split_row = split_row > S ? S : split_row;
  p.wq[0] = wq0; p.wk[0] = wk0; p.cs[0] = cos0; p.sn[0] = sin0;
  p.wq[1] = wq1; p.wk[1] = wk1; p.cs[1] = cos1; p.sn[1] = sin1;
  p.eps = eps;
  p.qscale = 0.08838834764831845f * 1.4426950408889634f;
  p.nt = ldc_cdiv(S, KT);
  const long long tiles = static_cast<long long>(B) * H * p.nt;
  p.Qp = static_cast<unsigned char*>(packed);
  p.Kp = p.Qp + tiles * QTILE_B;
  p.Vp = p.Kp + tiles * KTILE_B;
  hipLaunchKernelGGL(attn_pack_kernel, dim3(p.nt, H, B), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ldc_launch_status();
}

extern "C" int ldc_attn_fwd_packed_bf16x3(const void* packed, float* O, int B, int S, int H, int ldo, long long o_bs,
                                          int out_split, void* stream) {
  LDC_CHECK_PTR(packed);
  LDC_CHECK_PTR(O);
  if (B <= 0 || S <= 0 || H <= 0) return LDC_ERR_ARG;
  LDC_CHECK_ALIGN16(packed);
  LDC_CHECK_ALIGN16(O);
  if ((ldo & 3) || (o_bs & 3)) return LDC_ERR_ALIGN;
  if (out_split && ((ldo & 7) || (o_bs & 7) || (reinterpret_cast<unsigned long long>(O) & 31ull))) return LDC_ERR_ALIGN;
  if (static_cast<long long>(ldc_cdiv(S, QB)) * H * B > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  AttnArgs p{};
  const bool one_term = (out_split & LDC_ATTN_BF16_1TERM) != 0;
  out_split &= LDC_ATTN_OUT_SPLIT;
  p.out_split = out_split ? 1 : 0;
  p.nt = ldc_cdiv(S, KT);
  const long long tiles = static_cast<long long>(B) * H * p.nt;
  p.Qp = static_cast<const unsigned char*>(packed);
  p.Kp = p.Qp + tiles * QTILE_B;
  p.Vp = p.Kp + tiles * KTILE_B;
  p.O = O;
  p.S = S; p.H = H; p.ldo = ldo; p.o_bs = o_bs;
  p.nq = ldc_cdiv(S, QB);
  dim3 grid(static_cast<unsigned>(p.nq) * H * B);
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_packed_kernel<1, 3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, GROUP_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_packed_kernel<2, 3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GROUP_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_packed_kernel<1, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, GROUP_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_packed_kernel<2, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GROUP_LDS);
    return true;
  }();
  (void)attr_set;
  const long long nwg = static_cast<long long>(p.nq) * H * B;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (one_term) {
    if (nwg <= 256) hipLaunchKernelGGL((attn_fwd_packed_kernel<2, 1>), grid, dim3(512), 2 * GROUP_LDS, st, p);
    else hipLaunchKernelGGL((attn_fwd_packed_kernel<1, 1>), grid, dim3(256), GROUP_LDS, st, p);
  } else {
    if (nwg <= 256) hipLaunchKernelGGL((attn_fwd_packed_kernel<2, 3>), grid, dim3(512), 2 * GROUP_LDS, st, p);
    else hipLaunchKernelGGL((attn_fwd_packed_kernel<1, 3>), grid, dim3(256), GROUP_LDS, st, p);
  }
  return ldc_launch_status();
}
