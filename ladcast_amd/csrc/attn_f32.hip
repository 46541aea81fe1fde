// Flash-style joint attention on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32).
// Replaces F.scaled_dot_product_attention(q, k, v) + the transpose back to [B, N, C]
// at models/LaDCast_3D_model.py:199-203 for head_dim 128, no mask, non-causal.
//
// Layout: Q/K/V are token-major [B][S][H][128] column slices of one fused QKV buffer
// (row stride ld_qkv), exactly what the QKV projection GEMM writes, so no head transpose
// ever touches HBM; O is written token-major [B][S][H*128].
//
// Work split: workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns
// 32 query rows and sweeps all keys in tiles of 32.  Per tile and wave:
//   S^T = K . Q^T   (64 MFMAs, A = K tile from LDS via ds_read_b128, B = Q held in 64 VGPRs)
//     -> the accumulator has the QUERY on the lane and 16 keys in registers, so the
//        row-wise softmax is 16 in-lane ops + one cross-half shuffle (64-wide wavefront:
//        lanes l and l+32 hold the two key halves of the same query);
//   O^T += V^T . P^T (64 MFMAs, A = V tile from LDS via conflict-free ds_read_b32,
//        B = the probability registers as they stand -- accumulator register r of half h
//        holds key (r&3)+8(r>>2)+4h, which is exactly the k-pair {kappa, kappa+4} one
//        32x32x2 MFMA sums over; no LDS round trip, no lane movement).
// The softmax scale 1/sqrt(128) and log2(e) are folded into Q once; exp2 is used.
// K/V tiles are staged HBM -> VGPR -> LDS double-buffered (66 KiB -> 2 workgroups / CU);
// the K tile uses a 132-float pitch (conflict-free for the b128 lane groups).
// NGRP = 2 (round 3, grids of at most 256 workgroups - one member of the 375M model has 216): the workgroup is 8 waves, waves 0-3 sweep
// the first half of the key tiles and waves 4-7 the second, each group through its own double buffer (133 KiB, one workgroup per CU),
// merged through LDS at the end - as in attn_split.hip.  With one 4-wave workgroup per CU every SIMD held a single wave, whose MFMAs
// waited for its own softmax and LDS traffic (0.59 of the fp32 matrix peak); two waves per SIMD overlap them.
// BAL (round 5, needs the caller's workspace): the launch's (unit, key tile) items are cut into one contiguous range per CU - 216 units of
// 71 key tiles (one member of the 375M model) are 59.9 tiles for each of 256 workgroups instead of 71 for 216 of them.  A range covers
// the end of one unit and the start of the next ("pieces"): each piece runs the same two-group sweep over its tiles, a piece that is not a
// whole unit publishes its un-normalised (O, m, l) in a write-through slab, takes a ticket on the unit's counter, and the piece with the
// last ticket re-reads ALL pieces of the unit in range order (bitwise reproducible whatever the arrival order), normalises and stores -
// the hand-off of gemm_bf16x3_v3.hip's stream-K cut.
#include "common.h"

namespace {

constexpr int HD = 128;      // head dim
constexpr int QB = 128;      // query rows per workgroup
constexpr int KT = 32;       // keys per tile
constexpr int KP = HD + 4;   // K tile pitch (floats)
constexpr int STAGE = KT * KP + KT * HD;  // floats per stage

struct AttnArgs {
  const float* Q;
  const float* K;
  const float* V;
  float* O;
  int S, H, ld_qkv, ldo;
  long long qkv_bs, o_bs;
  float qscale;
  int nq;  // query blocks per (batch, head)
  const float* kbias;  // additive per-key score bias (an SDPA float mask that only depends on the key), [S] or nullptr
  // BAL: range g = tile-items [g * upg + min(g, urem), ...) of the launch's units x nt items; slabs [2 G][SLAB_FLOATS]; one counter per unit
  int nt;
  unsigned upg, urem;
  float* ws;
  unsigned* counters;
};
constexpr int SLAB_FLOATS = 66 * 256;  // [64 O values + m + l][256 threads of key group 0]

__device__ __forceinline__ unsigned range_start(unsigned g, const AttnArgs& p) { return g * p.upg + (g < p.urem ? g : p.urem); }
__device__ __forceinline__ unsigned range_of(unsigned x, const AttnArgs& p) {  // the range that holds tile-item x
  const unsigned big = p.urem * (p.upg + 1);
  return x < big ? x / (p.upg + 1) : p.urem + (x - big) / p.upg;
}

template <int NGRP, bool BAL = false>
__global__ __launch_bounds__(256 * NGRP, NGRP == 1 ? 2 : 1) void attn_fwd_f32_kernel(AttnArgs p) {
  static_assert(!BAL || NGRP == 2, "the balanced schedule runs the 8-wave form");
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int grp = NGRP == 2 ? __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 8) : 0;  // key-range group (wave-uniform)
  float* const smem = smem_all + grp * 2 * STAGE;
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int S = p.S;
  // XCD-aware placement (speed only): the query blocks of one (batch, head) re-read the same K/V (2.3 MB at
  // S = 2250), so lay the 1-D grid out as [batch][head][query block] and give each XCD a contiguous run of it
  // (blocks are dealt round-robin over the 8 XCDs; bijective remap for any grid size).
  // BAL: the same numbering over the ranges - an XCD's workgroups sweep consecutive units
  int lin;
  {
    const int T = gridDim.x;
    const int bid = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = bid & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = (S + KT - 1) / KT;
  unsigned u = 0, u_end = 0, u_first = 0;
  if constexpr (BAL) {
    u_first = u = range_start(static_cast<unsigned>(lin), p);
    u_end = range_start(static_cast<unsigned>(lin) + 1u, p);
  }
  do {  // BAL: the pieces of this workgroup's range; otherwise one pass = one whole unit
  int unit = lin, t_lo = 0, t_hi = nt;
  if constexpr (BAL) {
    unit = static_cast<int>(u / static_cast<unsigned>(nt));
    t_lo = static_cast<int>(u) - unit * nt;
    const int left = static_cast<int>(u_end - u);
    t_hi = t_lo + left < nt ? t_lo + left : nt;
  }
  const int qblk = unit % p.nq;
  const int hb = unit / p.nq;
  const int head = hb % p.H;
  const int b = hb / p.H;
  const int q0 = qblk * QB + wave * 32;

  const long long base = static_cast<long long>(b) * p.qkv_bs + static_cast<long long>(head) * HD;
  const float* __restrict__ Qg = p.Q + base;
  const float* __restrict__ Kg = p.K + base;
  const float* __restrict__ Vg = p.V + base;
  const int ld = p.ld_qkv;

  // Q fragments: qf[c][r] = Q[q0 + l31][8c + 4*half + r] * qscale
  float qf[16][4];
  {
    const int qrow = q0 + l31;
    const bool ok = qrow < S;
    const float* qp = Qg + static_cast<long long>(ok ? qrow : 0) * ld + 4 * half;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float4 v = ok ? *reinterpret_cast<const float4*>(qp + 8 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      qf[c][0] = v.x * p.qscale;
      qf[c][1] = v.y * p.qscale;
      qf[c][2] = v.z * p.qscale;
      qf[c][3] = v.w * p.qscale;
    }
  }

  // staging map: thread t moves float4 column c4 = t&31 of key rows (t>>5) + 8*i
  const int c4 = tid & 31;
  const int kr0 = tid >> 5;
  float4 rk[4], rv[4];
  auto gload = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = key0 + kr0 + 8 * i;
      if (key < S) {
        const long long off = static_cast<long long>(key) * ld + c4 * 4;
        rk[i] = *reinterpret_cast<const float4*>(Kg + off);
        rv[i] = *reinterpret_cast<const float4*>(Vg + off);
      } else {
        rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto sstore = [&](int stage) {
    float* Ks = smem + stage * STAGE;
    float* Vs = Ks + KT * KP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = kr0 + 8 * i;
      *reinterpret_cast<float4*>(Ks + key * KP + c4 * 4) = rk[i];
      *reinterpret_cast<float4*>(Vs + key * HD + c4 * 4) = rv[i];
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -1.0e30f;
  float l_run = 0.f;

  const int nspan = t_hi - t_lo;                             // key tiles of this pass (all of the unit's unless BAL)
  const int nhalf = NGRP == 2 ? (nspan + 1) >> 1 : nspan;  // iterations of the longer group
  const int t_begin = t_lo + (grp ? nhalf : 0);
  const int t_end = grp ? t_hi : t_lo + nhalf;             // group 1 may have one tile fewer (or none)
  if (t_begin < t_end) {
    gload(t_begin * KT);
    sstore(0);
  }
  __syncthreads();

  // (tried in round 5 and dropped, profiles/r05_m_attn_f32_balanced.log: the two key groups half an iteration apart - a second barrier
  // between the score phase and the softmax, group 1 one barrier late, so that one wave of a SIMD is in its softmax while the other issues
  // MFMAs: 270.2 us against 271.3; a second score accumulator to break the 64-MFMA chain: 276 us)
  for (int it = 0; it < nhalf; ++it) {
    const int t = t_begin + it;
    if (t >= t_end) {  // (wave-uniform) the shorter group only keeps the barrier count
      __syncthreads();
      continue;
    }
#if !(defined(LDC_AB_BUILD) && defined(LDC_ATTN_DIAG_NOSTAGE))  // diagnostic builds (make variant_src SRC=attn_f32 ...): results are garbage, timing only
    if (t + 1 < t_end) gload((t + 1) * KT);
#endif
    const float* Ks = smem + (it & 1) * STAGE;
    const float* Vs = Ks + KT * KP;

    // ---- S^T = K . Q^T ------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* kb = Ks + l31 * KP + 4 * half;
    // K fragments three chunks ahead of their MFMAs (the 64 MFMAs are one dependent chain: a fragment read only one chunk ahead - what the
    // compiler schedules from plain loads - leaves its LDS latency exposed every 8 MFMAs, in both waves of the SIMD at once: the barrier
    // keeps the two key groups in step).  Hand-placed reads and counted waits; nothing else may move into this block.
    {
      typedef float f4v __attribute__((ext_vector_type(4)));
      const unsigned ka = static_cast<unsigned>(reinterpret_cast<unsigned long long>(kb));
      f4v k0, k1, k2, k3;
#if defined(LDC_AB_BUILD) && defined(LDC_ATTN_DIAG_NOREADS)
#define LDC_RDK(dst, off) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(ka) : "memory")
#else
#define LDC_RDK(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ka), "n"(off) : "memory")
#endif
#define LDC_MM4(kv, c)                                                          \
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[0], qf[c][0], s, 0, 0, 0);        \
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[1], qf[c][1], s, 0, 0, 0);        \
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[2], qf[c][2], s, 0, 0, 0);        \
  s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[3], qf[c][3], s, 0, 0, 0);
#define LDC_STEP(kv, knext, c, WAIT)                                            \
  if ((c) + 3 < 16) LDC_RDK(knext, 32 * ((c) + 3 < 16 ? (c) + 3 : 15));         \
  asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" : "+v"(kv));                      \
  LDC_MM4(kv, c)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (whatever the compiler had in flight: the counted waits below start from zero)
      LDC_RDK(k0, 0);
      LDC_RDK(k1, 32);
      LDC_RDK(k2, 64);
      LDC_STEP(k0, k3, 0, 3)
      LDC_STEP(k1, k0, 1, 3)
      LDC_STEP(k2, k1, 2, 3)
      LDC_STEP(k3, k2, 3, 3)
      LDC_STEP(k0, k3, 4, 3)
      LDC_STEP(k1, k0, 5, 3)
      LDC_STEP(k2, k1, 6, 3)
      LDC_STEP(k3, k2, 7, 3)
      LDC_STEP(k0, k3, 8, 3)
      LDC_STEP(k1, k0, 9, 3)
      LDC_STEP(k2, k1, 10, 3)
      LDC_STEP(k3, k2, 11, 3)
      LDC_STEP(k0, k3, 12, 3)
      LDC_STEP(k1, k0, 13, 2)
      LDC_STEP(k2, k1, 14, 1)
      LDC_STEP(k3, k2, 15, 0)
      __builtin_amdgcn_sched_barrier(0);
#undef LDC_STEP
#undef LDC_MM4
#undef LDC_RDK
    }
    // the next tile's K / V go to their LDS stage now (it has been free since the barrier; the loads were issued a whole S phase ago):
    // the 32 staging registers are dead through the softmax and the P.V phase
#if !(defined(LDC_AB_BUILD) && defined(LDC_ATTN_DIAG_NOSTAGE))
    if (t + 1 < t_end) sstore((it + 1) & 1);
#endif

#if !(defined(LDC_AB_BUILD) && defined(LDC_ATTN_DIAG_NOSOFTMAX))
    // ---- online softmax over the key axis (registers + the other lane half) --
    const int key_base = t * KT + 4 * half;
    if (p.kbias) {  // scale_attn_by_lat (models/LaDCast_3D_model.py:873-882); scores are in log2 units here
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key_base + (r & 3) + 8 * (r >> 2);
        s[r] += p.kbias[key < S ? key : S - 1] * 1.4426950408889634f;
      }
    }
    if (t == nt - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key_base + (r & 3) + 8 * (r >> 2);
        if (key >= S) s[r] = -1.0e30f;
      }
    }
    float m_t = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m_t = fmaxf(m_t, s[r]);
    m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
    const float m_new = fmaxf(m_run, m_t);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // (v_exp_f32: arguments <= 0; what would be denormal is 0)
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      rs += s[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }

#endif
    // ---- O^T += V^T . P^T ------------------------------------------------------
    // V fragments (one scalar per (key kappa, d tile): V[4 half + kappa][32 d + l31]) three keys ahead of their MFMAs, hand-placed like
    // the K reads: ds_read2st64_b32 fetches the d / d + 2 pair of a key from one base (offsets in units of 256 B, all immediates)
    {
      typedef float f2v __attribute__((ext_vector_type(2)));
      const unsigned va = static_cast<unsigned>(reinterpret_cast<unsigned long long>(Vs + (4 * half) * HD + l31));
      const unsigned vb1 = va + 128u;  // d tiles 1 and 3
      f2v a0, a1, a2, a3, b0, b1, b2, b3;
#define LDC_KAPPA(r) (((r) & 3) + 8 * ((r) >> 2))
#if defined(LDC_AB_BUILD) && defined(LDC_ATTN_DIAG_NOREADS)
#define LDC_RDV(da, db, r)                                               \
  asm volatile("; no read %0 %1" : "=v"(da) : "v"(va) : "memory");       \
  asm volatile("; no read %0 %1" : "=v"(db) : "v"(vb1) : "memory");
#else
#define LDC_RDV(da, db, r)                                                                                                             \
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(da) : "v"(va), "n"(2 * LDC_KAPPA(r)), "n"(2 * LDC_KAPPA(r) + 1) : "memory"); \
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(db) : "v"(vb1), "n"(2 * LDC_KAPPA(r)), "n"(2 * LDC_KAPPA(r) + 1) : "memory");
#endif
#define LDC_PV(da, db, na, nb, r, WAIT)                                          \
  if ((r) + 3 < 16) { LDC_RDV(na, nb, ((r) + 3 < 16 ? (r) + 3 : 15)) }           \
  asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" : "+v"(da), "+v"(db));             \
  o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[0], s[r], o[0], 0, 0, 0);       \
  o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(db[0], s[r], o[1], 0, 0, 0);       \
  o[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[1], s[r], o[2], 0, 0, 0);       \
  o[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(db[1], s[r], o[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging stores / softmax shuffles above: the counted waits start from zero
      LDC_RDV(a0, b0, 0)
      LDC_RDV(a1, b1, 1)
      LDC_RDV(a2, b2, 2)
      LDC_PV(a0, b0, a3, b3, 0, 6)
      LDC_PV(a1, b1, a0, b0, 1, 6)
      LDC_PV(a2, b2, a1, b1, 2, 6)
      LDC_PV(a3, b3, a2, b2, 3, 6)
      LDC_PV(a0, b0, a3, b3, 4, 6)
      LDC_PV(a1, b1, a0, b0, 5, 6)
      LDC_PV(a2, b2, a1, b1, 6, 6)
      LDC_PV(a3, b3, a2, b2, 7, 6)
      LDC_PV(a0, b0, a3, b3, 8, 6)
      LDC_PV(a1, b1, a0, b0, 9, 6)
      LDC_PV(a2, b2, a1, b1, 10, 6)
      LDC_PV(a3, b3, a2, b2, 11, 6)
      LDC_PV(a0, b0, a3, b3, 12, 6)
      LDC_PV(a1, b1, a0, b0, 13, 4)
      LDC_PV(a2, b2, a1, b1, 14, 2)
      LDC_PV(a3, b3, a2, b2, 15, 0)
      __builtin_amdgcn_sched_barrier(0);
#undef LDC_PV
#undef LDC_RDV
#undef LDC_KAPPA
    }

#if !(defined(LDC_AB_BUILD) && defined(LDC_ATTN_DIAG_NOBARRIER))
    __syncthreads();
#endif
  }

  // ---- merge the two key halves: group 1 hands (m, l, O) to group 0 through LDS (the rings are idle: barrier above) ----
  if constexpr (NGRP == 2) {
    float* xch = smem_all + tid;  // [66][256]
    if (grp == 1) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 256] = o[d][r];
      xch[64 * 256] = m_run;
      xch[65 * 256] = l_run;
    }
    __syncthreads();
    if constexpr (!BAL) {
      if (grp == 1) return;
    }
    if (grp == 0) {
      const float m1 = xch[64 * 256], l1 = xch[65 * 256];
      const float m = fmaxf(m_run, m1);
      const float a0 = exp2f(m_run - m), a1 = exp2f(m1 - m);
      l_run = l_run * a0 + l1 * a1;
      m_run = m;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * a0 + xch[(d * 16 + r) * 256] * a1;
    }
  }

  // ---- normalise and store: lane holds O[q0+l31][32d + 8g + 4half + (0..3)] in o[d][4g..4g+3]
  auto store_rows = [&]() {
    const int qrow = q0 + l31;
    if (qrow < S) {
      const float inv = 1.0f / l_run;
      float* op = p.O + static_cast<long long>(b) * p.o_bs + static_cast<long long>(qrow) * p.ldo + head * HD + 4 * half;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
          *reinterpret_cast<float4*>(op + 32 * d + 8 * g) = v;
        }
    }
  };
  if constexpr (!BAL) {
    store_rows();
  } else {
    if (t_lo == 0 && t_hi == nt) {  // (workgroup-uniform) a whole unit inside the range
      if (grp == 0) store_rows();
    } else {
      // publish this piece: write-through slab, drained, ONE ticket per workgroup on the unit's counter
      const unsigned slot_id = 2u * static_cast<unsigned>(lin) + (u != u_first ? 1u : 0u);
      if (grp == 0) {
        float* slot = p.ws + static_cast<long long>(slot_id) * SLAB_FLOATS + tid;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) __hip_atomic_store(slot + (d * 16 + r) * 256, o[d][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 64 * 256, m_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 65 * 256, l_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned f = static_cast<unsigned>(unit) * static_cast<unsigned>(nt);
      const unsigned g_first = range_of(f, p), g_last = range_of(f + static_cast<unsigned>(nt) - 1u, p);
      unsigned* flag = reinterpret_cast<unsigned*>(smem_all + SLAB_FLOATS);  // behind the exchange area, ring memory (idle)
      if (threadIdx.x == 0) {
        unsigned* cnt = p.counters + unit;
        const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (ticket == g_last - g_first) ? 1u : 0u;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
        }
        *flag = last;
      }
      __syncthreads();
      if (*flag != 0u && grp == 0) {  // the last arriver: all pieces of the unit, in range order, from the slabs (its own too)
        m_run = -1.0e30f;
        l_run = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        for (unsigned gp = g_first; gp <= g_last; ++gp) {
          const unsigned s0 = range_start(gp, p);
          const float* sl = p.ws + static_cast<long long>(2u * gp + (s0 < f ? 1u : 0u)) * SLAB_FLOATS + tid;  // (slot 1: the piece does not start its range)
          const float mi = sl[64 * 256], li = sl[65 * 256];
          const float m = fmaxf(m_run, mi);
          const float a0 = exp2f(m_run - m), a1 = exp2f(mi - m);
          l_run = l_run * a0 + li * a1;
          m_run = m;
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * a0 + sl[(d * 16 + r) * 256] * a1;
        }
        store_rows();
      }
    }
    u += static_cast<unsigned>(nspan);
    __syncthreads();  // exchange area / flag word are ring memory of the next piece
  }
  } while (BAL && u < u_end);
}

}  // namespace

// workspace of the balanced schedule: [one counter per unit, zero between launches | 2 x 256 slabs].  The caller zero-fills it ONCE.
constexpr long long BAL_MAX_UNITS = 65536;
constexpr int BAL_G = 256;
constexpr long long BAL_COUNTER_BYTES = BAL_MAX_UNITS * 4;
extern "C" long long ldc_attn_fwd_workspace_bytes(void) {
  return BAL_COUNTER_BYTES + 2LL * BAL_G * SLAB_FLOATS * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_attn_fwd_ws(const float* Q, const float* K, const float* V, float* O, int B, int S, int H, int ld_qkv,
                               long long qkv_bs, int ldo, long long o_bs, const float* key_bias, void* workspace,
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
  if ((ld_qkv & 3) || (ldo & 3) || (qkv_bs & 3) || (o_bs & 3)) return LDC_ERR_ALIGN;
  if (static_cast<long long>(ldc_cdiv(S, QB)) * H * B > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  AttnArgs p{Q, K, V, O, S, H, ld_qkv, ldo, qkv_bs, o_bs,
             0.08838834764831845f * 1.4426950408889634f};  // 1/sqrt(128) * log2(e)
  p.nq = ldc_cdiv(S, QB);
  p.kbias = key_bias;
  p.nt = ldc_cdiv(S, KT);
  const long long units = static_cast<long long>(p.nq) * H * B;
  dim3 grid(static_cast<unsigned>(units));
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_f32_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(2 * STAGE * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_f32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(4 * STAGE * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_f32_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(4 * STAGE * sizeof(float)));
    return true;
  }();
  (void)attr_set;
  // balanced schedule: whenever the units do not fill whole rounds of the 256 CUs and every range still holds two key tiles.  Measured
  // (profiles/r05_m_attn_f32_balanced.log, us plain -> balanced): 216 units (one 375M member) 317 -> 285, 288 (1.6B) 603 -> 376, 432 (two
  // members) 606 -> 549, 864 1196 -> 1084; 48 units of 15 tiles (the refiner's 450 tokens) 74 -> 33, 24 units of 32 tiles 146 -> 42.  Whole
  // rounds (256, 512 units) and long launches (1728 units) are as fast or faster on the plain grids (4 waves, two workgroups per CU).
  const long long items = units * p.nt;
  bool bal = workspace != nullptr && workspace_bytes >= ldc_attn_fwd_workspace_bytes() && units <= BAL_MAX_UNITS && units % BAL_G != 0 &&
             units <= 4 * BAL_G && items >= 2LL * BAL_G && items < (1LL << 31);
  static const char* const force_bal = LDC_AB_GETENV("LDC_ATTN_F32_BAL");  // measurement aid, read once: 0 = never, 1 = whenever possible
  if (force_bal) bal = workspace != nullptr && workspace_bytes >= ldc_attn_fwd_workspace_bytes() && units <= BAL_MAX_UNITS &&
                       items >= 2LL * BAL_G && items < (1LL << 31) && atoi(force_bal) != 0;
  if (bal) {
    LDC_CHECK_ALIGN16(workspace);
    p.counters = static_cast<unsigned*>(workspace);
    p.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + BAL_COUNTER_BYTES);
    p.upg = static_cast<unsigned>(items / BAL_G);
    p.urem = static_cast<unsigned>(items % BAL_G);
    hipLaunchKernelGGL((attn_fwd_f32_kernel<2, true>), dim3(BAL_G), dim3(512), 4 * STAGE * sizeof(float), static_cast<hipStream_t>(stream), p);
    return ldc_launch_status();
  }
  // at most one workgroup per CU anyway -> 8 waves with the keys split over two wave groups; more -> 4 waves, two workgroups per CU
  // NOTE the choice looks at the whole launch (B included), and the forms add the key tiles in different orders: a member's fp32
  // result can differ at rounding level (~1e-7) with how many members share the launch, like the stream-K cut of the GEMMs.  Partition
  // independence of the sharded rollout is therefore "to fp32 rounding", bitwise only for equal per-rank batches (pipelines/distributed.py).
  static const bool one_group_forced = LDC_AB_GETENV("LDC_ATTN_F32_ONE_GROUP") != nullptr;  // measurement aid, read once
  const bool two = grid.x <= 256 && !one_group_forced;
  const size_t lds = (two ? 4 : 2) * STAGE * sizeof(float);
  if (two) hipLaunchKernelGGL(attn_fwd_f32_kernel<2>, grid, dim3(512), lds, static_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(attn_fwd_f32_kernel<1>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return ldc_launch_status();
}

// without a workspace: the one-unit-per-workgroup grids
extern "C" int ldc_attn_fwd(const float* Q, const float* K, const float* V, float* O, int B, int S, int H,
                            int ld_qkv, long long qkv_bs, int ldo, long long o_bs, const float* key_bias, void* stream) {
  return ldc_attn_fwd_ws(Q, K, V, O, B, S, H, ld_qkv, qkv_bs, ldo, o_bs, key_bias, nullptr, 0, stream);
}
