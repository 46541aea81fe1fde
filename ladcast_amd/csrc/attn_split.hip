// Split-bf16 attention, third generation: consumes q / k / v as ROW-MAJOR token rows in split-bf16 form - exactly what the
// QKV projection's GEMM epilogue writes (gemm_bf16x3_v3.hip, LDC_GEMM_C_QKV: bias -> per-head RMSNorm -> rotary embedding ->
// softmax scale -> hi / lo split) - so no pack pass and no second copy of the operands exists (attn_dma_bf16x3.hip needed one:
// 19 us + a 95 MB round trip per call).  What made the pack pass necessary there was the V^T tile image; here V stays row-major
// and the transposition happens in the LDS read (ds_read_b64_tr_b16).
//
// Operand rows (512 B per token and head, the bytes 128 floats would occupy in the fused [B][S][3 D] buffer):
//   q, k  16 groups of 8 head-dim values, each [hi x8 | lo x8] bf16 (the LDC_GEMM_A_SPLIT activation format); q pre-scaled by
//         log2(e) / sqrt(128)
//   v     [hi x128 | lo x128] bf16 (two planes: a 64-byte run per 32 head-dim values for the DMA below)
// MFMA v_mfma_f32_16x16x32_bf16 throughout (the shape this chip holds a higher clock on; MI355X_MICROARCH.md DVFS item 7).
// Workgroup = 8 waves = 128 queries of one (batch, head): waves 0-3 sweep the first half of the 32-key tiles, waves 4-7 the second
// (merged through LDS at the end), or 4 waves over all tiles when the grid has more than 256 workgroups.  A wave owns 32 queries
// (two 16-query tiles).  Per tile:
//   S^T[key][query] = K . Q^T    A = K fragment (16 keys x 32 d), B = Q^T fragment from registers          48 MFMAs (3 per product)
//   softmax in the accumulator layout: a lane holds 2 queries x 8 keys; row max over the 4 lane groups by permlane16/32 swaps
//   O^T[d][query] += V^T . P^T   A = V^T fragment by two transposed reads, B = P straight from the S accumulators (the key order
//                                inside a 32-key block is the accumulators': k = 8 g + j <-> key 16 (j >> 2) + 4 g + (j & 3))  48 MFMAs
// K tile in LDS: 4 sub-images (32 head-dim values each) of [32 keys][128 B] with the GEMM's conflict-free chunk swizzle, filled by
// LDS-DMA straight from the split rows (swizzle on the per-lane source address).  V tile: [hi | lo][key tile][d tile] blocks of
// [16 keys][16 d] bf16 = 512 B, so that the transposed read of a block is `block + 8 * lane` (every bank exactly once per
// 32-lane half) and every block is an immediate offset from one address register; filled by LDS-DMA, 64 B per key and instruction.
// Software pipeline (two tiles in flight, cdna_hip_programming.md T15): while the MFMAs of S_next = K_(t+1).Q^T issue, the VALU
// does the row sums and the hi / lo split of tile t's probabilities; while O += V_t.P_t issues, it does the running max and the
// exponentials of S_next.  The iteration body is generated (tools/gen_attn_split_body.py -> attn_split_body_t{3,1}.inc): every
// counted lgkmcnt is derived from the issue order of the fragment reads.
#include "common.h"

namespace {

constexpr int HD = 128;
constexpr int QB = 128;  // queries per workgroup
constexpr int KT = 32;   // keys per tile
constexpr int KTILE_B = 16384, VTILE_B = 16384;
constexpr int GROUP_LDS = 2 * KTILE_B + 2 * VTILE_B;  // per key-range group: 2-stage K ring + 2-stage V ring = 64 KiB

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_pair(float a, float b) {
  bf16x2 v;
  v.x = static_cast<__bf16>(a);
  v.y = static_cast<__bf16>(b);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned split_pair(float a, float b, float& ra, float& rb) {
  const unsigned u = pack_pair(a, b);
  ra = a - __uint_as_float(u << 16);
  rb = b - __uint_as_float(u & 0xffff0000u);
  return u;
}
__device__ __forceinline__ int swz(int r) { return ((r >> 1) & 7) ^ ((((r >> 2) ^ (r >> 3)) & 1) << 1); }  // gemm_bf16x3_v3.hip

// ------------------------------------------------------------------------------------------------------------------------
// Stand-alone producer of the operand rows from an fp32 [B][S][3][H][128] buffer, IN PLACE (what the QKV GEMM epilogue does when
// the projection runs through gemm_bf16x3_v3.hip): per-head RMSNorm(128) * weight -> adjacent-pair RoPE on q and k (same operation
// order as qk_rmsnorm_rope_kernel / attn_pack_kernel), q * log2(e)/sqrt(128), hi / lo split.  8 consecutive lanes own one
// (token, head); every lane reads its 16 floats of q, k and v before anything is written.
// ------------------------------------------------------------------------------------------------------------------------
struct PrepArgs {
  float* Q;
  float* K;
  float* V;
  int S, H, ld;
  long long bs;
  int split_row;
  const float* wq[2];
  const float* wk[2];
  const float* cs[2];
  const float* sn[2];
  float eps, qscale;
};

__device__ __forceinline__ void norm_rope16(float (&x)[16], const float* __restrict__ w, const float* __restrict__ cs,
                                            const float* __restrict__ sn, int trow, int d0, float eps) {
  if (w) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) ss += x[i] * x[i] + x[i + 1] * x[i + 1];
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

__global__ __launch_bounds__(256) void qkv_prepare_split_kernel(PrepArgs p) {
  const int tid = threadIdx.x;
  const int sg = tid & 7;
  const long long item = static_cast<long long>(blockIdx.x) * 32 + (tid >> 3);  // (row, head) pairs, head fastest
  const int b = blockIdx.y;
  if (item >= static_cast<long long>(p.S) * p.H) return;
  const int row = static_cast<int>(item / p.H), head = static_cast<int>(item - static_cast<long long>(row) * p.H);
  const int seg = row < p.split_row ? 0 : 1;
  const int trow = seg ? row - p.split_row : row;
  const long long base = static_cast<long long>(b) * p.bs + static_cast<long long>(row) * p.ld + head * HD;
  float xq[16], xk[16], xv[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a = reinterpret_cast<const float4*>(p.Q + base + 16 * sg)[i];
    xq[4 * i] = a.x; xq[4 * i + 1] = a.y; xq[4 * i + 2] = a.z; xq[4 * i + 3] = a.w;
    const float4 c = reinterpret_cast<const float4*>(p.K + base + 16 * sg)[i];
    xk[4 * i] = c.x; xk[4 * i + 1] = c.y; xk[4 * i + 2] = c.z; xk[4 * i + 3] = c.w;
    const float4 d = reinterpret_cast<const float4*>(p.V + base + 16 * sg)[i];
    xv[4 * i] = d.x; xv[4 * i + 1] = d.y; xv[4 * i + 2] = d.z; xv[4 * i + 3] = d.w;
  }
  norm_rope16(xq, p.wq[seg], p.cs[seg], p.sn[seg], trow, 16 * sg, p.eps);
  norm_rope16(xk, p.wk[seg], p.cs[seg], p.sn[seg], trow, 16 * sg, p.eps);
  // all loads of the wave have returned (their values were just used): in-place stores cannot overtake another lane's load
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned char* qd = reinterpret_cast<unsigned char*>(p.Q + base);
  unsigned char* kd = reinterpret_cast<unsigned char*>(p.K + base);
  unsigned char* vd = reinterpret_cast<unsigned char*>(p.V + base);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float r[8];
    uint4 h, l;
    const float sc = p.qscale;
    h.x = split_pair(xq[8 * e] * sc, xq[8 * e + 1] * sc, r[0], r[1]); h.y = split_pair(xq[8 * e + 2] * sc, xq[8 * e + 3] * sc, r[2], r[3]);
    h.z = split_pair(xq[8 * e + 4] * sc, xq[8 * e + 5] * sc, r[4], r[5]); h.w = split_pair(xq[8 * e + 6] * sc, xq[8 * e + 7] * sc, r[6], r[7]);
    l.x = pack_pair(r[0], r[1]); l.y = pack_pair(r[2], r[3]); l.z = pack_pair(r[4], r[5]); l.w = pack_pair(r[6], r[7]);
    *reinterpret_cast<uint4*>(qd + (2 * sg + e) * 32) = h;
    *reinterpret_cast<uint4*>(qd + (2 * sg + e) * 32 + 16) = l;
    h.x = split_pair(xk[8 * e], xk[8 * e + 1], r[0], r[1]); h.y = split_pair(xk[8 * e + 2], xk[8 * e + 3], r[2], r[3]);
    h.z = split_pair(xk[8 * e + 4], xk[8 * e + 5], r[4], r[5]); h.w = split_pair(xk[8 * e + 6], xk[8 * e + 7], r[6], r[7]);
    l.x = pack_pair(r[0], r[1]); l.y = pack_pair(r[2], r[3]); l.z = pack_pair(r[4], r[5]); l.w = pack_pair(r[6], r[7]);
    *reinterpret_cast<uint4*>(kd + (2 * sg + e) * 32) = h;
    *reinterpret_cast<uint4*>(kd + (2 * sg + e) * 32 + 16) = l;
    h.x = split_pair(xv[8 * e], xv[8 * e + 1], r[0], r[1]); h.y = split_pair(xv[8 * e + 2], xv[8 * e + 3], r[2], r[3]);
    h.z = split_pair(xv[8 * e + 4], xv[8 * e + 5], r[4], r[5]); h.w = split_pair(xv[8 * e + 6], xv[8 * e + 7], r[6], r[7]);
    l.x = pack_pair(r[0], r[1]); l.y = pack_pair(r[2], r[3]); l.z = pack_pair(r[4], r[5]); l.w = pack_pair(r[6], r[7]);
    *reinterpret_cast<uint4*>(vd + (2 * sg + e) * 16) = h;        // hi plane
    *reinterpret_cast<uint4*>(vd + 256 + (2 * sg + e) * 16) = l;  // lo plane
  }
}

// ------------------------------------------------------------------------------------------------------------------------
struct AttnArgs {
  const unsigned char* Q;
  const unsigned char* K;
  const unsigned char* V;
  float* O;
  int S, H, ldo;
  long long ldb, bsb;  // row / batch stride of the operand rows, bytes
  long long o_bs;
  int nq, nt;
  int out_split;
  const float* kbias;  // additive per-key score bias (natural-log units, as an SDPA float mask), 32 * nt floats, or nullptr
  // TAIL schedule (more units than CUs): units [0, n_full) run whole, one per workgroup and round; each of the remaining units is cut
  // into `slices` key ranges ("pieces", n_pieces = tail units x slices <= workgroups) whose un-normalised (O, m, l) go to `part` and
  // are combined by attn_tail_merge_kernel
  int n_full, n_pieces, slices;
  float* part;
};
constexpr int PART_FLOATS = 68 * 256;  // per piece: [64 O values + m0, m1, l0, l1][4 waves x 64 lanes]

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<unsigned long long>(p));
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// combine with the lane 16 / 32 further (v_permlane16_swap / v_permlane32_swap: no LDS crossbar op, so no lgkmcnt traffic)
__device__ __forceinline__ float xor16_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_add(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// This lane's index produced where it is asked for (two VALU instructions in a volatile asm: cannot be hoisted or shared).  The TAIL
// kernel's per-item prologue / epilogue geometry derives from it, so that nothing of it is live across the item's key loop - derived
// from threadIdx once, the compiler hoists those values out of the item loop and spills 46 registers (gemm_bf16x3_v3.hip: fresh_lane).
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

#define LDC_SB __builtin_amdgcn_sched_barrier(0)
#define LDC_RDK(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDC_RDV(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDC_CAT(a, b) __builtin_shufflevector(a, b, 0, 1, 2, 3)
#define LDC_BF(x) __builtin_bit_cast(bf16x8, x)
#define LDC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// one pair of probabilities -> hi / lo bf16 pair I of the P fragment (anchored where it is written)
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
// keys of tile t + 1 past S (only its last tile can have them): key = 32 (t + 1) + 16 kt + 4 g4 + r
#define LDC_MASK_TAIL(SN)                                                        \
  {                                                                              \
    const int kb_ = (t + 1) * KT + 4 * g4;                                       \
    _Pragma("unroll") for (int kt_ = 0; kt_ < 2; ++kt_)                          \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                             \
      if (kb_ + 16 * kt_ + r_ >= S) { SN[kt_][0][r_] = -1.0e30f; SN[kt_][1][r_] = -1.0e30f; } \
  }

// ---- epilogue: a lane owns query row 16 qt + c16 and, per d tile, the 4 consecutive columns 16 dt + 4 g4 .. + 3 ----
__device__ __forceinline__ void attn_store_rows(const AttnArgs& p, int b, int head, int q0, int c16, int g4, const f32x4 (&o)[8][2],
                                                const float (&l_run)[2]) {
  const int S = p.S;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = q0 + 16 * qt + c16;
    if (qrow >= S) continue;
    const float inv = 1.0f / l_run[qt];
    float* orow = p.O + static_cast<long long>(b) * p.o_bs + static_cast<long long>(qrow) * p.ldo + head * HD;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const int n = 16 * dt + 4 * g4;
      float4 v = make_float4(o[dt][qt][0] * inv, o[dt][qt][1] * inv, o[dt][qt][2] * inv, o[dt][qt][3] * inv);
      if (p.out_split == LDC_FMT_BF16) {  // plain bf16 row: this lane's 4 columns are 8 bytes at byte offset 2 n
        unsigned char* rowb = reinterpret_cast<unsigned char*>(orow - head * HD);  // bf16 columns are 2 bytes: offset from the ROW start
        *reinterpret_cast<uint2*>(rowb + 2 * (head * HD + n)) = make_uint2(pack_pair(v.x, v.y), pack_pair(v.z, v.w));
      } else if (p.out_split) {
        // LDC_GEMM_A_SPLIT format: columns 8c..8c+7 live in 32 bytes [hi x8 | lo x8]; this lane has half of a group, the lane 16
        // further the other half: v_permlane16_swap gives the even lane group both hi halves, the odd one both lo halves
        float r0, r1, r2, r3;
        const unsigned hx = split_pair(v.x, v.y, r0, r1), hy = split_pair(v.z, v.w, r2, r3);
        const unsigned lx = pack_pair(r0, r1), ly = pack_pair(r2, r3);
        const auto sx = __builtin_amdgcn_permlane16_swap(hx, lx, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(hy, ly, false, false);
        unsigned char* grp8 = reinterpret_cast<unsigned char*>(orow + (n & ~7)) + 4 * (n & 4);
        *reinterpret_cast<uint4*>(grp8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      } else {
        *reinterpret_cast<float4*>(orow + n) = v;
      }
    }
  }
}

// one (query row, 4 columns of d tile dt) of a lane, normalised and stored in the call's output format (attn_store_rows for one d tile)
__device__ __forceinline__ void attn_store_tile(const AttnArgs& p, int b, int head, int qrow, int dt, int g4, f32x4 ov, float l_sum) {
  const float inv = 1.0f / l_sum;
  float* orow = p.O + static_cast<long long>(b) * p.o_bs + static_cast<long long>(qrow) * p.ldo + head * HD;
  const int n = 16 * dt + 4 * g4;
  float4 v = make_float4(ov[0] * inv, ov[1] * inv, ov[2] * inv, ov[3] * inv);
  if (p.out_split == LDC_FMT_BF16) {
    unsigned char* rowb = reinterpret_cast<unsigned char*>(orow - head * HD);
    *reinterpret_cast<uint2*>(rowb + 2 * (head * HD + n)) = make_uint2(pack_pair(v.x, v.y), pack_pair(v.z, v.w));
  } else if (p.out_split) {
    float r0, r1, r2, r3;
    const unsigned hx = split_pair(v.x, v.y, r0, r1), hy = split_pair(v.z, v.w, r2, r3);
    const unsigned lx = pack_pair(r0, r1), ly = pack_pair(r2, r3);
    const auto sx = __builtin_amdgcn_permlane16_swap(hx, lx, false, false);
    const auto sy = __builtin_amdgcn_permlane16_swap(hy, ly, false, false);
    unsigned char* grp8 = reinterpret_cast<unsigned char*>(orow + (n & ~7)) + 4 * (n & 4);
    *reinterpret_cast<uint4*>(grp8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
  } else {
    *reinterpret_cast<float4*>(orow + n) = v;
  }
}

// TAIL schedule, second launch: the `slices` pieces of tail unit u (pieces u * slices ..) -> its output rows.  One 256-thread workgroup per
// (unit, d tile), the same (wave, lane) -> (query, columns) ownership as the attention kernel; pieces are combined in slice order (fixed:
// the result does not depend on which piece finished first).  A thread reads 12 floats per slice, all slices' loads independent.
__global__ __launch_bounds__(256) void attn_tail_merge_kernel(AttnArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int dt = blockIdx.y;
  const int unit = p.n_full + blockIdx.x;
  const int qblk = unit % p.nq, hb = unit / p.nq;
  const int head = hb % p.H, b = hb / p.H;
  const float* base = p.part + static_cast<long long>(blockIdx.x) * p.slices * PART_FLOATS + wave * 64 + lane;
  float m[2] = {-1.0e30f, -1.0e30f};
  for (int s = 0; s < p.slices; ++s) {
    m[0] = fmaxf(m[0], base[static_cast<long long>(s) * PART_FLOATS + 64 * 256]);
    m[1] = fmaxf(m[1], base[static_cast<long long>(s) * PART_FLOATS + 65 * 256]);
  }
  f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float l[2] = {0.f, 0.f};
  for (int s = 0; s < p.slices; ++s) {
    const float* pp = base + static_cast<long long>(s) * PART_FLOATS;
    const float a0 = exp2f(pp[64 * 256] - m[0]), a1 = exp2f(pp[65 * 256] - m[1]);
    l[0] += pp[66 * 256] * a0;
    l[1] += pp[67 * 256] * a1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[0][r] += pp[((dt * 2 + 0) * 4 + r) * 256] * a0;
      o[1][r] += pp[((dt * 2 + 1) * 4 + r) * 256] * a1;
    }
  }
  const int q0 = qblk * QB + wave * 32;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = q0 + 16 * qt + c16;
    if (qrow >= p.S) continue;  // (the 4 lanes of a row share qrow: the lane-group swap of the split format stays among active lanes)
    attn_store_tile(p, b, head, qrow, dt, g4, o[qt], l[qt]);
  }
}

// BIAS: scores += kbias[key] (models/LaDCast_3D_model.py:873-882 `scale_attn_by_lat`: a float attention mask that only depends on
// the key) - it enters as the initial value of the S accumulators, so it costs no instruction in the MFMA stream
// TAIL (NGRP = 2 only): a persistent workgroup per CU walks items item = blockIdx.x, + gridDim.x, ...: first whole units, then at most one
// key slice of a tail unit (see AttnArgs) - 288 units (16 heads x 18 query blocks) on 256 CUs are one round of whole units plus an eighth
// of a unit each instead of two uneven rounds.  TAIL = false compiles to the single-unit kernel (the loop condition is constant).
template <int NGRP, int TERMS, bool BIAS, bool TAIL = false>
__global__ __launch_bounds__(256 * NGRP, NGRP == 1 ? 2 : 1) void attn_fwd_split_kernel(AttnArgs p_arg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int grp = NGRP == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;  // key-range group (wave-uniform)
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15;  // query column of the 16x16 tiles / fragment row
  const int g4 = lane >> 4;   // lane group: keys 4 g4 .. 4 g4 + 3 of a 16-key tile, k-group of an operand fragment
  // XCD-aware placement: contiguous runs of [batch][head][query block] per XCD (its K / V tiles stay in that XCD's L2).  TAIL: the
  // workgroup's items are lin0, lin0 + T, lin0 + 2 T, ... (T = grid size) in that permuted numbering - whole units below n_full, then
  // piece lin - n_full; a workgroup takes part in the last round only if its permuted index is a piece that exists
  int item;
  {
    const int T = gridDim.x, bid = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = bid & 7;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  do {
  const AttnArgs& p = p_arg;
  const int S = p.S;
  int head, b, qblk;
  int t_lo = 0, t_hi = p.nt, piece = -1;
  {
    const int nq = p.nq;
    int lin = item;
    if constexpr (TAIL) {
      if (lin >= p.n_full) {
        piece = lin - p.n_full;
        const int u = piece / p.slices, sl = piece - u * p.slices;
        lin = p.n_full + u;
        t_lo = __builtin_amdgcn_readfirstlane(sl * p.nt / p.slices);
        t_hi = __builtin_amdgcn_readfirstlane((sl + 1) * p.nt / p.slices);
      }
    }
    qblk = lin % nq;
    const int hb = lin / nq;
    head = hb % p.H;
    b = hb / p.H;
  }
  const int q0 = qblk * QB + wave * 32;
  const int nt = p.nt;
  const unsigned ldb = static_cast<unsigned>(p.ldb);  // S * ldb < 2^32 (checked on the host): row offsets are one 32-bit multiply
  const long long bh_off = static_cast<long long>(b) * p.bsb + head * 512;

  // Q fragments (B operand): query 16 qt + c16, d = 32 s + 8 g4 + j <-> split group 4 s + g4
  bf16x8 qh[2][4], ql[2][4];
  const int lane_p = TAIL ? fresh_lane() : lane;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int qrow = q0 + 16 * qt + (lane_p & 15);
    qrow = qrow < S ? qrow : S - 1;  // rows past S: any valid row (never stored)
    const unsigned char* qp = p.Q + bh_off + static_cast<unsigned>(qrow) * ldb + 32 * (lane_p >> 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[qt][s] = *reinterpret_cast<const bf16x8*>(qp + 128 * s);
      if constexpr (TERMS == 3) ql[qt][s] = *reinterpret_cast<const bf16x8*>(qp + 128 * s + 16);
    }
  }

  unsigned char* const ring = smem + grp * GROUP_LDS;
  const unsigned ring_lds = lds_addr(ring);
  // K DMA: wave w fills rows [8 w, 8 w + 8) of the four sub-images; lane -> (row lr, 16-byte slot lp), source chunk lp ^ swz(row)
  const int krow = 8 * wave + (lane >> 3);
  // DMA sources = wave-uniform base (scalar register pair) + a 32-bit per-lane offset: one address register per source instead of
  // two (the <2, 3> kernel sat 4 registers over its budget: a Q fragment lived in scratch and was re-loaded - with a full vmcnt(0)
  // wait - in front of its MFMA in every iteration)
  auto uniform_ptr = [](const unsigned char* q_) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q_);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return reinterpret_cast<const unsigned char*>((static_cast<unsigned long long>(hi) << 32) | lo);
  };
  const unsigned char* const k_base = uniform_ptr(p.K + bh_off);
  const unsigned char* const v_base = uniform_ptr(p.V + bh_off);
  const unsigned k_lane = ((lane & 7) ^ swz(krow)) << 4;
  // V DMA: wave w fills d-tile pair w of the four [hi | lo][key tile] block rows; lane -> (d tile 2 w + (lane >> 5), key (lane & 31) >> 1,
  // 16-byte half lane & 1): 64 contiguous bytes of a plane per key
  const int vkey = (lane & 31) >> 1;
  const unsigned v_lane = (4 * wave + 2 * (lane >> 5) + (lane & 1)) * 16;
  // (TAIL: the per-lane parts are re-formed from a fresh lane index at every use - two tile-DMA offset registers fewer are live across
  // the MFMA phases of the kernel that runs at its 256-register budget; the item loop keeps a few more values live than one unit does)
  auto k_src = [&](int t) {  // 32-bit offset from k_base (S * ldb < 2^32, checked on the host)
    if constexpr (TAIL) {
      const int l_ = fresh_lane();
      const int kr_ = 8 * wave + (l_ >> 3);
      int key = t * KT + kr_;
      key = key < S ? key : S - 1;
      return (static_cast<unsigned>((l_ & 7) ^ swz(kr_)) << 4) + static_cast<unsigned>(key) * ldb;
    }
    int key = t * KT + krow;
    key = key < S ? key : S - 1;  // rows past S are masked in the scores: any finite data
    return k_lane + static_cast<unsigned>(key) * ldb;
  };
  auto v_src = [&](int t, int kt) {
    if constexpr (TAIL) {
      const int l_ = fresh_lane();
      int key = t * KT + 16 * kt + ((l_ & 31) >> 1);
      key = key < S ? key : S - 1;
      return static_cast<unsigned>((4 * wave + 2 * (l_ >> 5) + (l_ & 1)) * 16) + static_cast<unsigned>(key) * ldb;
    }
    int key = t * KT + 16 * kt + vkey;
    key = key < S ? key : S - 1;  // their probabilities are exactly 0
    return v_lane + static_cast<unsigned>(key) * ldb;
  };
  auto issue_k = [&](int t, int stage) {
    const unsigned src = k_src(t);
    unsigned char* dst = ring + stage * KTILE_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(k_base + (src + i * 128), dst + i * 4096);
  };
  auto issue_v = [&](int t, int stage) {
    const unsigned s0 = v_src(t, 0);
    const unsigned s1 = v_src(t, 1);
    unsigned char* dst = ring + 2 * KTILE_B + stage * VTILE_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(v_base + (((i & 1) ? s1 : s0) + (i >> 1) * 256), dst + i * 4096);
  };

  f32x4 o[8][2];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    o[d][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    o[d][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-1.0e30f, -1.0e30f};
  float l_run[2] = {0.f, 0.f};  // per-lane partial row sums (this lane's 8 keys of every tile): reduced over the lane groups at the end
  float alpha[2] = {1.f, 1.f};

  const int nspan = t_hi - t_lo;                               // key tiles of this item (all of them unless it is a tail piece)
  const int nhalf = NGRP == 2 ? (nspan + 1) >> 1 : nspan;    // iterations of the longer group
  const int t_begin = t_lo + (grp ? nhalf : 0);
  const int t_end = grp ? t_hi : t_lo + nhalf;                // group 1 may have one tile fewer (or none)

  // per-lane fragment bases inside a stage
  const unsigned f_r = swz(c16);
  const unsigned k_hi_off = c16 * 128 + (((2 * g4) ^ f_r) << 4);
  const unsigned k_lo_off = c16 * 128 + (((2 * g4 + 1) ^ f_r) << 4);
  const unsigned v_off = 2 * KTILE_B + 8 * lane;
  i32x4v kh0, kl0, kh1, kl1, kh2, kl2, kh3, kl3;  // four K fragment pairs in flight
  i32x2v vh0a, vh0b, vl0a, vl0b, vh1a, vh1b, vl1a, vl1b, vh2a, vh2b, vl2a, vl2b, vh3a, vh3b, vl3a, vl3b;  // four V^T pairs
  if constexpr (TERMS == 1) {  // never loaded in the single-term mode
    kl0 = kl1 = kl2 = kl3 = i32x4v{0, 0, 0, 0};
    vl0a = vl0b = vl1a = vl1b = vl2a = vl2b = vl3a = vl3b = i32x2v{0, 0};
  }

  // ---- prologue: scores of the group's first tile, their running max and exponentials (nothing to overlap with yet) ----
  f32x4 eA[2][2], eB[2][2];
  // key bias of a tile in the accumulator layout (keys 32 t + 16 kt + 4 g4 + r), in log2 units; prefetched one tile ahead
  auto load_bias = [&](int t, int kt) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.kbias + t * KT + 16 * kt + 4 * g4);
    return v * 1.4426950408889634f;
  };
  f32x4 kb_next[2] = {zero4, zero4};
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) eA[kt][qt] = eB[kt][qt] = zero4;
  if constexpr (BIAS) {
    if (t_begin < t_end) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const f32x4 v = load_bias(t_begin, kt);
        eA[kt][0] = v;
        eA[kt][1] = v;
        kb_next[kt] = load_bias(t_begin + 1 < nt ? t_begin + 1 : t_begin, kt);
      }
    }
  }
  if (t_begin < t_end) issue_k(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t_begin < t_end) {
    if (t_begin + 1 < t_end) issue_k(t_begin + 1, 1);
    issue_v(t_begin, 0);
    const unsigned k_hi = ring_lds + k_hi_off, k_lo = ring_lds + k_lo_off;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (s == 0 && kt == 0) { LDC_RDK(kh0, k_hi, 0); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 0); }
        if (s == 0 && kt == 1) { LDC_RDK(kh0, k_hi, 2048); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 2048); }
        if (s == 1 && kt == 0) { LDC_RDK(kh0, k_hi, 4096); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 4096); }
        if (s == 1 && kt == 1) { LDC_RDK(kh0, k_hi, 6144); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 6144); }
        if (s == 2 && kt == 0) { LDC_RDK(kh0, k_hi, 8192); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 8192); }
        if (s == 2 && kt == 1) { LDC_RDK(kh0, k_hi, 10240); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 10240); }
        if (s == 3 && kt == 0) { LDC_RDK(kh0, k_hi, 12288); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 12288); }
        if (s == 3 && kt == 1) { LDC_RDK(kh0, k_hi, 14336); if constexpr (TERMS == 3) LDC_RDK(kl0, k_lo, 14336); }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh0), "+v"(kl0));
        LDC_SB;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          if constexpr (TERMS == 3) {
            eA[kt][qt] = LDC_MFMA(LDC_BF(kl0), qh[qt][s], eA[kt][qt]);
            eA[kt][qt] = LDC_MFMA(LDC_BF(kh0), ql[qt][s], eA[kt][qt]);
          }
          eA[kt][qt] = LDC_MFMA(LDC_BF(kh0), qh[qt][s], eA[kt][qt]);
        }
        LDC_SB;
      }
    }
    if (t_begin == nt - 1) {
      const int t = t_begin - 1;
      LDC_MASK_TAIL(eA)
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = max3f(max3f(eA[0][qt][0], eA[0][qt][1], eA[0][qt][2]), max3f(eA[0][qt][3], eA[1][qt][0], eA[1][qt][1]),
                       fmaxf(eA[1][qt][2], eA[1][qt][3]));
      mx = xor32_max(xor16_max(mx));
      m_run[qt] = mx;  // at least one key of a tile is valid: finite
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) eA[kt][qt][r] = __builtin_amdgcn_exp2f(eA[kt][qt][r] - mx);
    }
  }

  // ---- one iteration: tile t = t_begin + it.  ec = exp2(S_t - m) (from the previous iteration), sn receives S of tile t + 1 ----
  auto iteration = [&](int it, f32x4 (&ec)[2][2], f32x4 (&sn)[2][2]) __attribute__((always_inline)) {
    const int t = t_begin + it;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K_{t+1} and V_t landed; everyone is done with K_t and V_{t-1}
    if (t >= t_end) return;
    // the DMA instructions of K_{t+2} (into K_t's stage) and V_{t+1} (into V_{t-1}'s stage) sit in MFMA gaps of phase A.  They are
    // unconditional (no branch in the MFMA stream): past the group's last tile the tile index is clamped, and what then lands in the
    // two stages is only ever read by iterations that do not exist (V) or whose scores are discarded (K, S_next of the last iteration)
    const bool vq = t + 1 < t_end;
    const bool live_next = vq;  // S_next is a real tile (the last iteration computes it from a stale stage: NaN / inf possible)
    const bool mask_next = (t + 1 == nt - 1);  // ... and may hold keys past S
    const unsigned ksrc = k_src(t + 2 < t_end ? t + 2 : t_end - 1);
    const unsigned vsrc0 = v_src(vq ? t + 1 : t_end - 1, 0);
    const unsigned vsrc1 = v_src(vq ? t + 1 : t_end - 1, 1);
    unsigned char* const kdst = ring + (it & 1) * KTILE_B + wave * 1024;
    unsigned char* const vdst = ring + 2 * KTILE_B + ((it + 1) & 1) * VTILE_B + wave * 1024;
    auto dma_k = [&](int j) __attribute__((always_inline)) { dma16(k_base + (ksrc + j * 128), kdst + j * 4096); };
    auto dma_v = [&](int j) __attribute__((always_inline)) { dma16(v_base + (((j & 1) ? vsrc1 : vsrc0) + (j >> 1) * 256), vdst + j * 4096); };
    const unsigned k_hi = ring_lds + ((it + 1) & 1) * KTILE_B + k_hi_off;  // K_{t+1}
    const unsigned k_lo = ring_lds + ((it + 1) & 1) * KTILE_B + k_lo_off;
    const unsigned v_ad = ring_lds + (it & 1) * VTILE_B + v_off;            // V_t
    float t0, t1, t2, t3, mx[2], m_new[2], alpha_n[2];
    i32x4v ph[2], pl[2];
    const f32x4 c0[2] = {BIAS ? kb_next[0] : zero4, BIAS ? kb_next[1] : zero4};  // S_next accumulators start from the key bias of tile t + 1
    if constexpr (BIAS) {  // prefetch tile t + 2's (consumed after the next barrier)
      const int tb = t + 2 < nt ? t + 2 : nt - 1;
      kb_next[0] = load_bias(tb, 0);
      kb_next[1] = load_bias(tb, 1);
    }
    if constexpr (TERMS == 1) pl[0] = pl[1] = i32x4v{0, 0, 0, 0};
    if constexpr (TERMS == 3) {
#include "attn_split_body_t3.inc"
    } else {
#include "attn_split_body_t1.inc"
    }
    alpha[0] = alpha_n[0];
    alpha[1] = alpha_n[1];
  };

  // the second-dispatched half of an 8-wave workgroup loses VALU arbitration to the older half on every segment: one static
  // priority raise for it, no per-cluster flips (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if constexpr (NGRP == 2) {
    if (grp == 1) __builtin_amdgcn_s_setprio(1);
  }
  for (int it = 0; it < nhalf; it += 2) {
    iteration(it, eA, eB);
    if (it + 1 < nhalf) iteration(it + 1, eB, eA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last iterations' (discarded) tiles have landed: the rings can be reused
  __syncthreads();  // everyone is done reading the rings

  // row sums: the four lane groups of a query each hold the sum over their keys
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) l_run[qt] = xor32_add(xor16_add(l_run[qt]));

  // ---- merge the two key halves: group 1 hands (m, l, O) to group 0 through LDS ----
  if constexpr (NGRP == 2) {
    float* xch = reinterpret_cast<float*>(smem) + (wave * 68) * 64 + (TAIL ? fresh_lane() : lane);  // [wave][68][64 lanes]
    if (grp == 1) {
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) xch[((d * 2 + qt) * 4 + r) * 64] = o[d][qt][r];
      xch[64 * 64] = m_run[0];
      xch[65 * 64] = m_run[1];
      xch[66 * 64] = l_run[0];
      xch[67 * 64] = l_run[1];
    }
    __syncthreads();
    if constexpr (!TAIL) {
      if (grp == 1) return;
    }
    if (grp == 0) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float m1 = xch[(64 + qt) * 64], l1 = xch[(66 + qt) * 64];
        const float m = fmaxf(m_run[qt], m1);
        const float a0 = exp2f(m_run[qt] - m), a1 = exp2f(m1 - m);
        l_run[qt] = l_run[qt] * a0 + l1 * a1;
        m_run[qt] = m;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[d][qt][r] = o[d][qt][r] * a0 + xch[((d * 2 + qt) * 4 + r) * 64] * a1;
      }
    }
    if constexpr (TAIL) {
      __syncthreads();  // the exchange area is ring memory of the next item: group 0 has read it
      if (grp == 1) {
        item += gridDim.x;
        continue;
      }
    }
  }

  if constexpr (TAIL) {
    const int lane_e = fresh_lane();
    // the epilogue's arguments (O, strides, format, workspace) are re-read from the kernarg segment HERE, through a laundered pointer:
    // kept in scalar registers across the item loop they exhaust the SGPR file (105 of 102) and the DMA base pointers fall back to
    // per-lane 64-bit registers that are spilled and re-loaded, behind a vmcnt(0), in every key-loop iteration
    AttnArgs pe{};
#if defined(__HIP_DEVICE_COMPILE__)
    {
      typedef const __attribute__((address_space(4))) AttnArgs* KArgs;
      KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      pe.O = ka->O; pe.S = ka->S; pe.ldo = ka->ldo; pe.o_bs = ka->o_bs; pe.out_split = ka->out_split; pe.part = ka->part;
    }
#endif
    const AttnArgs& p = pe;
    if (piece >= 0) {  // a key slice of a tail unit: un-normalised O and the softmax statistics of the slice, combined by attn_tail_merge_kernel
      float* pp = p.part + static_cast<long long>(piece) * PART_FLOATS + wave * 64 + lane_e;
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) pp[((d * 2 + qt) * 4 + r) * 256] = o[d][qt][r];
      pp[64 * 256] = m_run[0];
      pp[65 * 256] = m_run[1];
      pp[66 * 256] = l_run[0];
      pp[67 * 256] = l_run[1];
    } else {
      attn_store_rows(p, b, head, q0, lane_e & 15, lane_e >> 4, o, l_run);
    }
  } else {
    attn_store_rows(p, b, head, q0, c16, g4, o, l_run);
  }
  item += gridDim.x;
  } while (TAIL && item < p_arg.n_full + p_arg.n_pieces);
}

}  // namespace

extern "C" int ldc_attn_qkv_prepare_split(float* Q, float* K, float* V, int B, int S, int H, int ld_qkv, long long qkv_bs,
                                          int split_row, const float* wq0, const float* wk0, const float* cos0, const float* sin0,
                                          const float* wq1, const float* wk1, const float* cos1, const float* sin1, float eps,
                                          void* stream) {
  LDC_CHECK_PTR(Q);
  LDC_CHECK_PTR(K);
  LDC_CHECK_PTR(V);
  if (B <= 0 || S <= 0 || H <= 0 || split_row < 0) return LDC_ERR_ARG;
  if ((wq0 == nullptr) != (wk0 == nullptr) || (wq1 == nullptr) != (wk1 == nullptr)) return LDC_ERR_ARG;
  if ((cos0 == nullptr) != (sin0 == nullptr) || (cos1 == nullptr) != (sin1 == nullptr)) return LDC_ERR_ARG;
  LDC_CHECK_ALIGN16(Q);
  LDC_CHECK_ALIGN16(K);
  LDC_CHECK_ALIGN16(V);
  if ((ld_qkv & 3) || (qkv_bs & 3)) return LDC_ERR_ALIGN;
  if (B > 65535) return LDC_ERR_UNSUPPORTED;
  PrepArgs p{};
  p.Q = Q; p.K = K; p.V = V;
  p.S = S; p.H = H; p.ld = ld_qkv; p.bs = qkv_bs;
  p.split_row = split_row > S ? S : split_row;
  p.wq[0] = wq0; p.wk[0] = wk0; p.cs[0] = cos0; p.sn[0] = sin0;
  p.wq[1] = wq1; p.wk[1] = wk1; p.cs[1] = cos1; p.sn[1] = sin1;
  p.eps = eps;
  p.qscale = 0.08838834764831845f * 1.4426950408889634f;
  const long long items = static_cast<long long>(S) * H;
  hipLaunchKernelGGL(qkv_prepare_split_kernel, dim3(static_cast<unsigned>((items + 31) / 32), B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return ldc_launch_status();
}

// units of one call = query blocks x heads x batch; the TAIL schedule applies when they exceed the 256 CUs by at most half a round
static bool attn_tail_plan(long long nwg, int nt, int* n_full, int* n_pieces, int* slices) {
  constexpr int CUS = 256;
  if (nwg <= CUS || nwg > 8 * CUS) return false;
  const int rem = static_cast<int>(nwg % CUS);
  if (rem == 0 || rem > CUS / 2) return false;  // a whole number of rounds / a tail that can only be cut once: the plain grids do as well
  int s = CUS / rem;
  if (s > 16) s = 16;
  if (s > nt) s = nt;
  if (s < 2) return false;
  *n_full = static_cast<int>(nwg) - rem;
  *slices = s;
  *n_pieces = rem * s;
  return true;
}

extern "C" long long ldc_attn_fwd_split_workspace_bytes(int B, int S, int H) {
  if (B <= 0 || S <= 0 || H <= 0) return 0;
  int nf, np, sl;
  if (!attn_tail_plan(static_cast<long long>(ldc_cdiv(S, QB)) * H * B, ldc_cdiv(S, KT), &nf, &np, &sl)) return 0;
  return static_cast<long long>(np) * PART_FLOATS * static_cast<long long>(sizeof(float));
}

// the most any call shape can ask for (attn_tail_plan: rem <= 128 left-over units x floor(256 / rem) slices <= 256 pieces): what a caller
// that captures launches into graphs allocates ONCE, so that no later shape ever makes it replace a pointer a captured launch holds
extern "C" long long ldc_attn_fwd_split_workspace_max_bytes(void) {
  return 256LL * PART_FLOATS * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_attn_fwd_split(const float* Q, const float* K, const float* V, float* O, int B, int S, int H, int ld_qkv,
                                  long long qkv_bs, int ldo, long long o_bs, const float* key_bias, int flags, void* workspace,
                                  long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(Q);
  LDC_CHECK_PTR(K);
  LDC_CHECK_PTR(V);
  LDC_CHECK_PTR(O);
  if (B <= 0 || S <= 0 || H <= 0) return LDC_ERR_ARG;
  LDC_CHECK_ALIGN16(Q);
  LDC_CHECK_ALIGN16(K);
  LDC_CHECK_ALIGN16(V);
  LDC_CHECK_ALIGN16(O);
  if ((ld_qkv & 3) || (qkv_bs & 3) || (ldo & 3) || (o_bs & 3)) return LDC_ERR_ALIGN;
  if (key_bias) LDC_CHECK_ALIGN16(key_bias);
  const int out_split = (flags & LDC_ATTN_OUT_BF16) ? LDC_FMT_BF16 : (flags & LDC_ATTN_OUT_SPLIT) ? LDC_FMT_SPLIT : 0;
  const bool one_term = (flags & LDC_ATTN_BF16_1TERM) != 0;
  if (out_split && ((ldo & 7) || (o_bs & 7) || (reinterpret_cast<unsigned long long>(O) & 31ull))) return LDC_ERR_ALIGN;
  if (static_cast<long long>(ldc_cdiv(S, QB)) * H * B > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  if (static_cast<long long>(S) * ld_qkv * 4 >= (1LL << 32)) return LDC_ERR_UNSUPPORTED;  // 32-bit row offsets inside one batch entry
  AttnArgs p{};
  p.Q = reinterpret_cast<const unsigned char*>(Q);
  p.K = reinterpret_cast<const unsigned char*>(K);
  p.V = reinterpret_cast<const unsigned char*>(V);
  p.O = O;
  p.S = S; p.H = H; p.ldo = ldo; p.o_bs = o_bs;
  p.ldb = static_cast<long long>(ld_qkv) * 4;
  p.bsb = qkv_bs * 4;
  p.out_split = out_split;
  p.nt = ldc_cdiv(S, KT);
  p.nq = ldc_cdiv(S, QB);
  dim3 grid(static_cast<unsigned>(p.nq) * H * B);
  p.kbias = key_bias;
  const long long nwg = static_cast<long long>(p.nq) * H * B;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  // more units than CUs by at most half a round (1.6B model: 16 heads x 18 query blocks = 288): one persistent 8-wave workgroup per CU
  // runs its whole units, then one key slice of a tail unit; the slices are merged by a second small launch.  Needs the caller's
  // workspace (ldc_attn_fwd_split_workspace_bytes); without it the 4-wave / two-per-CU form below runs (two uneven rounds).
  if (attn_tail_plan(nwg, p.nt, &p.n_full, &p.n_pieces, &p.slices) && workspace != nullptr &&
      workspace_bytes >= static_cast<long long>(p.n_pieces) * PART_FLOATS * static_cast<long long>(sizeof(float))) {
    LDC_CHECK_ALIGN16(workspace);
    p.part = static_cast<float*>(workspace);
    const dim3 tgrid(256);
#define LDC_ATTN_TAIL(T, BI)                                                                                                    \
  {                                                                                                                             \
    static const bool attr_set = [] {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_split_kernel<2, T, BI, true>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GROUP_LDS);                                     \
      return true;                                                                                                              \
    }();                                                                                                                        \
    (void)attr_set;                                                                                                             \
    hipLaunchKernelGGL((attn_fwd_split_kernel<2, T, BI, true>), tgrid, dim3(512), 2 * GROUP_LDS, st, p);                        \
  }
    if (one_term) {
      if (key_bias) LDC_ATTN_TAIL(1, true) else LDC_ATTN_TAIL(1, false)
    } else {
      if (key_bias) LDC_ATTN_TAIL(3, true) else LDC_ATTN_TAIL(3, false)
    }
#undef LDC_ATTN_TAIL
    const int st1 = ldc_launch_status();
    if (st1 != LDC_OK) return st1;
    hipLaunchKernelGGL(attn_tail_merge_kernel, dim3(static_cast<unsigned>(nwg - p.n_full), 8), dim3(256), 0, st, p);
    return ldc_launch_status();
  }
  const int variant = (nwg <= 256 ? 4 : 0) | (one_term ? 2 : 0) | (key_bias ? 1 : 0);
#define LDC_ATTN_LAUNCH(NG, T, BI)                                                                                              \
  {                                                                                                                             \
    static const bool attr_set = [] {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_split_kernel<NG, T, BI>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, NG * GROUP_LDS);                                    \
      return true;                                                                                                              \
    }();                                                                                                                        \
    (void)attr_set;                                                                                                             \
    hipLaunchKernelGGL((attn_fwd_split_kernel<NG, T, BI>), grid, dim3(256 * NG), NG * GROUP_LDS, st, p);                        \
  }
  switch (variant) {
    case 0: LDC_ATTN_LAUNCH(1, 3, false) break;
    case 1: LDC_ATTN_LAUNCH(1, 3, true) break;
    case 2: LDC_ATTN_LAUNCH(1, 1, false) break;
    case 3: LDC_ATTN_LAUNCH(1, 1, true) break;
    case 4: LDC_ATTN_LAUNCH(2, 3, false) break;
    case 5: LDC_ATTN_LAUNCH(2, 3, true) break;
    case 6: LDC_ATTN_LAUNCH(2, 1, false) break;
    default: LDC_ATTN_LAUNCH(2, 1, true) break;
  }
#undef LDC_ATTN_LAUNCH
  return ldc_launch_status();
}
