// Flash-style joint attention on the bf16 matrix cores with split-bf16 ("bf16x3") operands:
// both contractions S = Q.K^T and O = P.V are evaluated as  Xh.Yh + Xh.Yl + Xl.Yh  with
// x = hi + lo, hi = bf16(x), lo = bf16(x - hi), fp32 accumulation (v_mfma_f32_32x32x16_bf16).
// Same interface, work split and online softmax as attn_f32.hip (which stays the exact-fp32 path);
// the MFMA time per 32-key tile drops from 128 x 64 to 48 x 32 cycles per wave.
//
// Workgroup = 8 waves = 128 queries of one (batch, head): waves 0-3 sweep the first half of the keys, waves
// 4-7 the second half for the SAME queries (two independent LDS rings), and the two partial results
// (running max, sum, unnormalised O) are merged through LDS at the end.  At one member per GPU the grid
// is only 18 x 12 = 216 workgroups for 256 CUs; the key split gives every SIMD two waves to interleave
// (MFMA of one beside softmax / staging of the other) instead of one.
//
// Data flow per 4-wave group (4 waves x 32 queries):
//   * Q: loaded once per wave, pre-scaled by log2(e)/sqrt(128), split, kept as 8 k16-step B fragments
//     (hi and lo: 64 VGPRs);
//   * K tile (32 keys x 128): split while staged, LDS row = key: [hi 256 B | lo 256 B | 16 B pad]
//     (pitch 132 dwords: conflict-free b128 reads); the A fragment of step s is 16 contiguous bytes;
//   * S^T = K.Q^T puts the QUERY on the lane, so the softmax is in-lane + one shfl_xor 32, and the
//     probability registers are already a B fragment of the next product: registers 8t..8t+7 of lane
//     half h hold keys 16t + 8(j>>2) + 4h + (j&3) (cdna_hip_programming.md section 3, "An accumulator
//     tile as the next MFMA's operand");
//   * V tile is therefore staged TRANSPOSED and in that key order: LDS row = head-dim index d,
//     [hi 64 B | lo 64 B | 16 B pad] with key kappa at position 16t + 8h + j, so the A fragment
//     V^T[d][8 keys] of step t is again 16 contiguous bytes.  The transpose costs nothing extra: each
//     staging thread gathers its 8 keys of one d with lane-coalesced dword loads (64 consecutive d per
//     wave instruction) and writes one b128 per plane.
#include "common.h"

namespace {

constexpr int HD = 128;
constexpr int QB = 128;
constexpr int KT = 32;
constexpr int KPITCH = 528;                      // bytes per K row: 256 hi + 256 lo + 16 pad
constexpr int VPITCH = 144;                      // bytes per V^T row: 64 hi + 64 lo + 16 pad
constexpr int STAGE_BYTES = KT * KPITCH + HD * VPITCH;  // 16896 + 18432 = 35328

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct AttnArgs {
  const float* Q;
  const float* K;
  const float* V;
  float* O;
  int S, H, ld_qkv, ldo;
  long long qkv_bs, o_bs;
  float qscale;
  int nq;  // query blocks per (batch, head)
};

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

template <int NGRP>
__global__ __launch_bounds__(256 * NGRP, NGRP == 1 ? 2 : 1) void attn_fwd_bf16x3_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int grp = NGRP == 2 ? (threadIdx.x >> 8) : 0;  // key-range group 0 / 1
  const int tid = threadIdx.x & 255;  // thread index inside the 4-wave group
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int S = p.S;
  // XCD-aware placement (speed only): the query blocks of one (batch, head) re-read the same K/V (2.3 MB at
  // S = 2250), so lay the 1-D grid out as [batch][head][query block] and give each XCD a contiguous run of it
  // (blocks are dealt round-robin over the 8 XCDs; bijective remap for any grid size).
  int head, b, qblk;
  {
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

  const long long base = static_cast<long long>(b) * p.qkv_bs + static_cast<long long>(head) * HD;
  const float* __restrict__ Qg = p.Q + base;
  const float* __restrict__ Kg = p.K + base;
  const float* __restrict__ Vg = p.V + base;
  const int ld = p.ld_qkv;

  // Q fragments: step s, element j <-> d = 16 s + 8 half + j
  bf16x8 qh[8], ql[8];
  {
    const int qrow = q0 + l31;
    const bool ok = qrow < S;
    const float* qp = Qg + static_cast<long long>(ok ? qrow : 0) * ld + 8 * half;
    const float sc = ok ? p.qscale : 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s);
      const float4 c = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      uint4 h, l;
      split8(a.x * sc, a.y * sc, a.z * sc, a.w * sc, c.x * sc, c.y * sc, c.z * sc, c.w * sc, h, l);
      qh[s] = __builtin_bit_cast(bf16x8, h);
      ql[s] = __builtin_bit_cast(bf16x8, l);
    }
  }

  // ---- staging maps -------------------------------------------------------------------------------
  // K item: (key kk, 8-wide d chunk kc): id = tid + 256 i, kc = id & 15, kk = id >> 4      (i = 0, 1)
  // V item: (d vd, key group vg = 2 t + h): id = tid + 256 i, vd = id & 127, vg = id >> 7 (i = 0, 1)
  const int kc = tid & 15, kk0 = tid >> 4;           // second item: key + 16
  const int vd = tid & 127, vg0 = tid >> 7;          // second item: group + 2
  float4 rk[2][2];
  float rv[2][8];
  // per-thread source offsets (elements), fixed for the whole kernel: K rows kk0, kk0+16; V rows kappa(j) of the
  // thread's two key groups.  Full tiles (all but the last) load unconditionally; only the last tile clamps and zeroes.
  const int k_off0 = kk0 * ld + kc * 8, k_off1 = (kk0 + 16) * ld + kc * 8;
  int v_off[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int vg = vg0 + 2 * i, tt = vg >> 1, hh = vg & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) v_off[i][j] = (16 * tt + 8 * (j >> 2) + 4 * hh + (j & 3)) * ld + vd;
  }
  auto gload = [&](int key0) {
    const float* Kt = Kg + static_cast<long long>(key0) * ld;
    const float* Vt = Vg + static_cast<long long>(key0) * ld;
    if (key0 + KT <= S) {  // wave-uniform: a full tile
      const float4* s0 = reinterpret_cast<const float4*>(Kt + k_off0);
      const float4* s1 = reinterpret_cast<const float4*>(Kt + k_off1);
      rk[0][0] = s0[0];
      rk[0][1] = s0[1];
      rk[1][0] = s1[0];
      rk[1][1] = s1[1];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) rv[i][j] = Vt[v_off[i][j]];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int key = key0 + kk0 + 16 * i;
        const bool ok = key < S;
        const float4* src = reinterpret_cast<const float4*>(Kg + static_cast<long long>(ok ? key : 0) * ld + kc * 8);
        float4 a = src[0], c = src[1];
        const float m = ok ? 1.f : 0.f;
        rk[i][0] = make_float4(a.x * m, a.y * m, a.z * m, a.w * m);
        rk[i][1] = make_float4(c.x * m, c.y * m, c.z * m, c.w * m);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int vg = vg0 + 2 * i;  // = 2 t + h
        const int t = vg >> 1, h = vg & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = key0 + 16 * t + 8 * (j >> 2) + 4 * h + (j & 3);
          const bool ok = key < S;
          const float v = Vg[static_cast<long long>(ok ? key : 0) * ld + vd];
          rv[i][j] = ok ? v : 0.f;
        }
      }
    }
  };
  unsigned char* const ring = smem + grp * (2 * STAGE_BYTES);
  auto sstore = [&](int stage) {
    unsigned char* Ks = ring + stage * STAGE_BYTES;
    unsigned char* Vs = Ks + KT * KPITCH;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 h, l;
      split8(rk[i][0].x, rk[i][0].y, rk[i][0].z, rk[i][0].w, rk[i][1].x, rk[i][1].y, rk[i][1].z, rk[i][1].w, h, l);
      unsigned char* row = Ks + (kk0 + 16 * i) * KPITCH + kc * 16;
      *reinterpret_cast<uint4*>(row) = h;
      *reinterpret_cast<uint4*>(row + 256) = l;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 h, l;
      split8(rv[i][0], rv[i][1], rv[i][2], rv[i][3], rv[i][4], rv[i][5], rv[i][6], rv[i][7], h, l);
      unsigned char* row = Vs + vd * VPITCH + (vg0 + 2 * i) * 16;  // position 16 t + 8 h = 8 (2 t + h) keys = 16 B per group
      *reinterpret_cast<uint4*>(row) = h;
      *reinterpret_cast<uint4*>(row + 64) = l;
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -1.0e30f;
  float l_run = 0.f;

  const int nt = (S + KT - 1) / KT;
  const int nhalf = NGRP == 2 ? (nt + 1) >> 1 : nt;  // iterations of the longer group
  const int t_begin = grp ? nhalf : 0;
  const int t_end = grp ? nt : nhalf;                // group 1 may have one tile fewer (or none)
  if (t_begin < t_end) {
    gload(t_begin * KT);
    sstore(0);
  }
  __syncthreads();

  for (int it = 0; it < nhalf; ++it) {
    const int t = t_begin + it;
    const bool active = t < t_end;
    if (active) {
    if (t + 1 < t_end) gload((t + 1) * KT);
    const unsigned char* Ks = ring + (it & 1) * STAGE_BYTES;
    const unsigned char* Vs = Ks + KT * KPITCH;

    // ---- S^T = K . Q^T  (24 MFMAs) ----------------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const unsigned char* kb = Ks + l31 * KPITCH + 16 * half;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kb + 32 * st);
      const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kb + 256 + 32 * st);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[st], s, 0, 0, 0);
    }

    // ---- online softmax over the key axis (registers + the other lane half) ------------------------
    const int key_base = t * KT + 4 * half;
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
    const float alpha = exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = exp2f(s[r] - m_new);
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

    // ---- O^T += V^T . P^T  (24 MFMAs); P split on the fly ---------------------------------------------
    const unsigned char* vb = Vs + l31 * VPITCH + 16 * half;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      uint4 ph4, pl4;
      split8(s[8 * tt], s[8 * tt + 1], s[8 * tt + 2], s[8 * tt + 3], s[8 * tt + 4], s[8 * tt + 5], s[8 * tt + 6], s[8 * tt + 7], ph4, pl4);
      const bf16x8 ph = __builtin_bit_cast(bf16x8, ph4);
      const bf16x8 pl = __builtin_bit_cast(bf16x8, pl4);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned char* vr = vb + d * 32 * VPITCH + 32 * tt;
        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vr);
        const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vr + 64);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[d], 0, 0, 0);
      }
    }

    if (t + 1 < t_end) sstore((it + 1) & 1);
    }  // active
    __syncthreads();
  }

  // ---- merge the two key halves: group 1 hands (m, l, O) to group 0 through LDS (the rings are idle) ----
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
  }  // NGRP == 2

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
}

}  // namespace

extern "C" int ldc_attn_fwd_bf16x3(const float* Q, const float* K, const float* V, float* O, int B, int S, int H,
                                   int ld_qkv, long long qkv_bs, int ldo, long long o_bs, void* stream) {
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
  AttnArgs p{Q, K, V, O, S, H, ld_qkv, ldo, qkv_bs, o_bs, 0.08838834764831845f * 1.4426950408889634f};
  p.nq = ldc_cdiv(S, QB);
  dim3 grid(static_cast<unsigned>(p.nq) * H * B);
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_bf16x3_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_bf16x3_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES);
    return true;
  }();
  (void)attr_set;
  // few workgroups (one member per GPU): split the keys over two 4-wave groups so every SIMD has two waves;
  // enough workgroups: two 4-wave workgroups per CU do the same job without the merge
  const long long nwg = static_cast<long long>(p.nq) * H * B;
  if (nwg <= 256) {
    hipLaunchKernelGGL(attn_fwd_bf16x3_kernel<2>, grid, dim3(512), 4 * STAGE_BYTES, static_cast<hipStream_t>(stream), p);
  } else {
    hipLaunchKernelGGL(attn_fwd_bf16x3_kernel<1>, grid, dim3(256), 2 * STAGE_BYTES, static_cast<hipStream_t>(stream), p);
  }
  return ldc_launch_status();
}
