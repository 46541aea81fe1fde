// HBM-bound row kernels of the AR transformer: LayerNorm+modulate, q/k RMSNorm+RoPE,
// token mean, gated residual, small-M linear (weight-streaming GEMV).  One 64-lane wave
// per row, float4 (16 B/lane) coalesced accesses, wave-shuffle reductions -- no LDS.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// LayerNorm (biased variance, two-pass in registers) + y = n*mul + add
//   replaces AdaLayerNorm*.norm + modulation and nn.LayerNorm
//   (models/LaDCast_3D_model.py:257,270,441,502,507,524-529,546-555,1044)
// ---------------------------------------------------------------------------
constexpr int LN_MAX_V4 = 8;  // D <= 64 lanes * 8 float4 * 4 = 2048

// pre_y != nullptr (ldc_gate_residual_layernorm): the row is first updated in place, x += gate[b] * pre_y (the refiner's gated
// attention residual, models/LaDCast_3D_model.py:296-299), then normalised - one launch instead of gate_residual + LayerNorm
__global__ __launch_bounds__(256) void layernorm_mod_kernel(float* __restrict__ x, float* __restrict__ y,
                                                            int rows, int D, int ldx, long long x_bs, int ldy,
                                                            long long y_bs, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int split_row,
                                                            const float* __restrict__ scale2,
                                                            const float* __restrict__ shift2, int mod_bs, int mode,
                                                            float eps, int out_split, const float* __restrict__ pre_y,
                                                            const float* __restrict__ pre_gate, int ld_prey,
                                                            long long prey_bs, int gate_bs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (row >= rows) return;
  if (row >= split_row) {  // second row segment (the conditioning stream of a dual block): its own modulation vectors
    scale = scale2;
    shift = shift2;
  }
  float* xr = x + b * x_bs + static_cast<long long>(row) * ldx;
  float* yr = y + b * y_bs + static_cast<long long>(row) * ldy;
  const int nv4 = D >> 2;
  float4 v[LN_MAX_V4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      v[i] = reinterpret_cast<const float4*>(xr)[c];
      if (pre_y) {  // same arithmetic as gate_residual_kernel: one fma per element
        const float4 yv = reinterpret_cast<const float4*>(pre_y + b * prey_bs + static_cast<long long>(row) * ld_prey)[c];
        const float4 g = reinterpret_cast<const float4*>(pre_gate + static_cast<long long>(b) * gate_bs)[c];
        v[i] = make_float4(fmaf(yv.x, g.x, v[i].x), fmaf(yv.y, g.y, v[i].y), fmaf(yv.z, g.z, v[i].z), fmaf(yv.w, g.w, v[i].w));
        reinterpret_cast<float4*>(xr)[c] = v[i];
      }
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(sum) / static_cast<float>(D);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      const float a = v[i].x - mean, bq = v[i].y - mean, cq = v[i].z - mean, dq = v[i].w - mean;
      sq += (a * a + bq * bq) + (cq * cq + dq * dq);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / static_cast<float>(D) + eps);
  const float* sc = scale ? scale + (mode == 0 ? static_cast<long long>(b) * mod_bs : 0) : nullptr;
  const float* sh = shift ? shift + (mode == 0 ? static_cast<long long>(b) * mod_bs : 0) : nullptr;
  const float one = (mode == 0) ? 1.f : 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      float4 m = sc ? reinterpret_cast<const float4*>(sc)[c] : make_float4(1.f - one, 1.f - one, 1.f - one, 1.f - one);
      float4 a = sh ? reinterpret_cast<const float4*>(sh)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = (v[i].x - mean) * rstd * (one + m.x) + a.x;
      o.y = (v[i].y - mean) * rstd * (one + m.y) + a.y;
      o.z = (v[i].z - mean) * rstd * (one + m.z) + a.z;
      o.w = (v[i].w - mean) * rstd * (one + m.w) + a.w;
      if (out_split == LDC_FMT_BF16) {  // plain bf16 row: this lane's 4 columns are 8 bytes at byte offset 8 c
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(yr) + 8 * c) = make_uint2(ldc_pack_pair(o.x, o.y), ldc_pack_pair(o.z, o.w));
      } else if (out_split) {
        // split activation format (ladcast_hip.h, LDC_GEMM_A_SPLIT): columns 8g..8g+7 in 32 bytes [hi x8 | lo x8];
        // this lane has columns 4c..4c+3 = half of group c >> 1
        float r0, r1, r2, r3;
        uint2 hi, lo;
        hi.x = ldc_split_pair(o.x, o.y, r0, r1);
        hi.y = ldc_split_pair(o.z, o.w, r2, r3);
        lo.x = ldc_pack_pair(r0, r1);
        lo.y = ldc_pack_pair(r2, r3);
        unsigned char* grp = reinterpret_cast<unsigned char*>(yr + 8 * (c >> 1)) + 8 * (c & 1);
        *reinterpret_cast<uint2*>(grp) = hi;
        *reinterpret_cast<uint2*>(grp + 16) = lo;
      } else {
        reinterpret_cast<float4*>(yr)[c] = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// per-head RMSNorm(128) * weight, then adjacent-pair RoPE, in place on q and k.
// One wave per (token row, head, q|k); lane l owns the rotary pair (2l, 2l+1).
//   models/LaDCast_3D_model.py:103-169,183-186; diffusers RMSNorm / apply_rotary_emb
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_kernel(float* __restrict__ q, float* __restrict__ k, int row0,
                                                              int rows, int H, int ld, long long bs,
                                                              const float* __restrict__ wq,
                                                              const float* __restrict__ wk, float eps,
                                                              const float* __restrict__ cos_tab,
                                                              const float* __restrict__ sin_tab) {
  const int lane = threadIdx.x & 63;
  const long long item = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);  // (row, head, which)
  const int b = blockIdx.y;
  const long long n_items = static_cast<long long>(rows) * H * 2;
  if (item >= n_items) return;
  const int which = static_cast<int>(item & 1);
  const int head = static_cast<int>((item >> 1) % H);
  const int row = static_cast<int>((item >> 1) / H);
  float* base = (which ? k : q) + b * bs + static_cast<long long>(row0 + row) * ld + head * 128;
  const float* w = which ? wk : wq;
  float2 v = reinterpret_cast<float2*>(base)[lane];
  const float ss = wave_sum(v.x * v.x + v.y * v.y);
  const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
  const float2 wv = reinterpret_cast<const float2*>(w)[lane];
  v.x = v.x * r * wv.x;
  v.y = v.y * r * wv.y;
  if (cos_tab) {
    const float2 c = reinterpret_cast<const float2*>(cos_tab + static_cast<long long>(row) * 128)[lane];
    const float2 s = reinterpret_cast<const float2*>(sin_tab + static_cast<long long>(row) * 128)[lane];
    const float ox = v.x * c.x + (-v.y) * s.x;
    const float oy = v.y * c.y + v.x * s.y;
    v.x = ox;
    v.y = oy;
  }
  reinterpret_cast<float2*>(base)[lane] = v;
}

// ---------------------------------------------------------------------------
// mean over token rows: y[b][c] = mean_r x[b][r][c]
// block = 64 columns x 16 row-groups, 4 loads in flight per thread; LDS combine of the 16 partial sums
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows,
                                                         int D, int ldx, long long x_bs) {
  __shared__ float part[16][64];
  const int c = threadIdx.x & 63;
  const int col = blockIdx.x * 64 + c;
  const int g = threadIdx.x >> 6;
  const int b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four loads in flight per thread
  if (col < D) {
    const float* xb = x + b * x_bs + col;
    int r = g;
    for (; r + 48 < rows; r += 64) {
      s0 += xb[static_cast<long long>(r) * ldx];
      s1 += xb[static_cast<long long>(r + 16) * ldx];
      s2 += xb[static_cast<long long>(r + 32) * ldx];
      s3 += xb[static_cast<long long>(r + 48) * ldx];
    }
    for (; r < rows; r += 16) s0 += xb[static_cast<long long>(r) * ldx];
  }
  part[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && col < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][c];
    y[static_cast<long long>(b) * D + col] = t / static_cast<float>(rows);
  }
}

// the same mean (same summation order per column: bit-identical y) that also writes the rows it reads in the split
// activation format (ladcast_hip.h LDC_GEMM_A_SPLIT), so that the GEMM consuming x does not split it again.
// block = 64 column PAIRS x 16 row-groups
__global__ __launch_bounds__(1024) void mean_rows_split_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               float* __restrict__ xs, int rows, int D, int ldx,
                                                               long long x_bs, int lds, long long s_bs, int fmt) {
  __shared__ float2 part[16][64];
  const int c = threadIdx.x & 63;
  const int col = (blockIdx.x * 64 + c) * 2;
  const int g = threadIdx.x >> 6;
  const int b = blockIdx.y;
  float2 s0 = make_float2(0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (col < D) {
    const float* xb = x + b * x_bs + col;
    unsigned char* sb = fmt == LDC_FMT_BF16 ? reinterpret_cast<unsigned char*>(xs + b * s_bs) + 2 * col  // plain bf16 row
                                            : reinterpret_cast<unsigned char*>(xs + b * s_bs + (col & ~7)) + 2 * (col & 7);
    auto take = [&](int r, float2& acc) {
      const float2 v = *reinterpret_cast<const float2*>(xb + static_cast<long long>(r) * ldx);
      acc.x += v.x;
      acc.y += v.y;
      float r0, r1;
      const unsigned hi = ldc_split_pair(v.x, v.y, r0, r1);
      unsigned char* d = sb + static_cast<long long>(r) * lds * 4;
      *reinterpret_cast<unsigned*>(d) = hi;
      if (fmt != LDC_FMT_BF16) *reinterpret_cast<unsigned*>(d + 16) = ldc_pack_pair(r0, r1);
    };
    int r = g;
    for (; r + 48 < rows; r += 64) {
      take(r, s0);
      take(r + 16, s1);
      take(r + 32, s2);
      take(r + 48, s3);
    }
    for (; r < rows; r += 16) take(r, s0);
  }
  part[g][c] = make_float2((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
  __syncthreads();
  if (g == 0 && col < D) {
    float2 t = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      t.x += part[i][c].x;
      t.y += part[i][c].y;
    }
    y[static_cast<long long>(b) * D + col] = t.x / static_cast<float>(rows);
    y[static_cast<long long>(b) * D + col + 1] = t.y / static_cast<float>(rows);
  }
}

// out = resid + gate[b] * y
__global__ __launch_bounds__(256) void gate_residual_kernel(const float* __restrict__ resid, const float* __restrict__ y,
                                                            const float* __restrict__ gate, float* __restrict__ out,
                                                            int rows, int D, int ld_res, long long res_bs, int ld_y,
                                                            long long y_bs, int gate_bs) {
  const int b = blockIdx.z;
  const int row = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;  // float4 column
  if (c * 4 >= D) return;
  const float4 r = reinterpret_cast<const float4*>(resid + b * res_bs + static_cast<long long>(row) * ld_res)[c];
  const float4 v = reinterpret_cast<const float4*>(y + b * y_bs + static_cast<long long>(row) * ld_y)[c];
  const float4 g = reinterpret_cast<const float4*>(gate + static_cast<long long>(b) * gate_bs)[c];
  float4 o = make_float4(fmaf(v.x, g.x, r.x), fmaf(v.y, g.y, r.y), fmaf(v.z, g.z, r.z), fmaf(v.w, g.w, r.w));
  reinterpret_cast<float4*>(out + b * res_bs + static_cast<long long>(row) * ld_res)[c] = o;
}

// ---------------------------------------------------------------------------
// small-M linear: y[r][n] = act_out(sum_k W[n][k] act_in(x[r % x_rows][k]) + bias[n]) + add[r % add_rows][n]
// HBM-bound weight streaming (the AdaLN modulation GEMVs: one row per member).  A workgroup stages act_in(x) for
// its <= 8 rows in LDS once (K chunks of <= 2048), then each wave streams LS_CPW weight rows at a time: LS_CPW x
// K/256 float4 loads in flight per lane, every W byte read once.
// ---------------------------------------------------------------------------
constexpr int LS_ROWS = 8;
constexpr int LS_CPW_MAX = 4; // output columns per wave: 4 for wide outputs, 1 when that would leave CUs idle
constexpr int LS_KC = 2048;   // K chunk staged in LDS (8 rows x 2048 x 4 B = 64 KiB)

// v * (1 + scale) + shift with separately rounded product and sum (no fma contraction): the bits of temb_modulate_kernel
__device__ __forceinline__ float modulate_nocontract(float v, float scale, float shift) {
#pragma clang fp contract(off)
  const float f = 1.f + scale;
  const float p = v * f;
  return p + shift;
}

// FULL = false compiles the timestep-sinusoid input and the modulation epilogue out: the wide (LS_CPW_MAX) instantiation that streams
// the 38 D x D AdaLN matrix sits at 125 VGPRs = 4 waves per SIMD, and nine more registers would cost it a wave (76 -> 126 us measured)
// Kernels running at the same time (gfx950, measured: profiles/r04_z_gpu_sharing_first_read.log): when a wave of this
// kernel shares a SIMD with a wave that streams MFMAs - another process's 4-wave split attention, or any MFMA kernel on a second stream -
// the FIRST VALU read of the two upper dwords of each 64-bit half of the 128-bit LDS read below (x.y, x.w) can return 0 in lanes 48..63,
// whatever the wait before it (s_waitcnt lgkmcnt(0) plus 32 idle cycles changes nothing), while the second read is right.  One v_mov of
// those two registers before their use takes that first read; reading x.x / x.z instead does not help (control).  Two instructions per
// staged float4, no measurable cost (73.3 / 78.0 -> 72.5 / 77.2 us, tools/gemv_bench.py), bit-identical results; one process on one stream
// never runs two kernels at once and never showed the effect.  -DLDC_LS_NO_FIRST_READ in the A/B build compiles the guard out
// (tools/canary/first_read_repro.hip, built both ways by `make repro`, reproduces the wrong words with it: tests/test_gpu_multistream_victims.py).
__device__ __forceinline__ void ls_first_read(float4& v) {
#if !(defined(LDC_AB_BUILD) && defined(LDC_LS_NO_FIRST_READ))
  float t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(t0), "=&v"(t1), "+v"(v.y), "+v"(v.w));
#endif
}

template <int LS_CPW, bool FULL = true>
__device__ __forceinline__ void linear_small_body(const float* __restrict__ x, int x_rows, const float* __restrict__ W,
                                                  const float* __restrict__ bias, const float* __restrict__ add,
                                                  int add_rows, float* __restrict__ y, int rows, int N, int K, int kc,
                                                  int act_in, int act_out, int iters, const float* __restrict__ mod = nullptr,
                                                  int mod_rows = 1) {
  // iters > 1 (host: only when K <= kc, one staged chunk): the workgroup stages act_in(x) once and walks `iters`
  // consecutive column groups - fewer, longer-lived workgroups for the wide AdaLN modulation GEMV
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [rows_here][kc]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r_base = blockIdx.y * LS_ROWS;
  const int rows_here = rows - r_base < LS_ROWS ? rows - r_base : LS_ROWS;
  for (int it = 0; it < iters; ++it) {
    const int n0 = ((blockIdx.x * iters + it) * 4 + wave) * LS_CPW;
    const float* wr[LS_CPW];
#pragma unroll
    for (int j = 0; j < LS_CPW; ++j) {
      const int n = n0 + j < N ? n0 + j : N - 1;  // clamped: columns past N are computed and dropped
      wr[j] = W + static_cast<long long>(n) * K;
    }
    float acc[LS_CPW][LS_ROWS];
#pragma unroll
    for (int j = 0; j < LS_CPW; ++j)
#pragma unroll
      for (int i = 0; i < LS_ROWS; ++i) acc[j][i] = 0.f;

    for (int k0 = 0; k0 < K; k0 += kc) {
      const int kn = K - k0 < kc ? K - k0 : kc;  // multiple of 4
      const int nv4 = kn >> 2;
      if (it == 0) {
        if (k0) __syncthreads();
        for (int idx = threadIdx.x; idx < rows_here * nv4; idx += 256) {
          const int i = idx / nv4, c = idx - i * nv4;
          float4 xv;
          if (FULL && act_in == LDC_ACT_IN_TIMESTEP_SINCOS) {
            // x holds one timestep per row; the staged row is its 256-wide sinusoidal embedding (same expressions as
            // timestep_embedding_kernel: bit-identical values), K == 256
            const float t = x[(r_base + i) % x_rows];
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int k = k0 + 4 * c + j;
              const float f = expf(-9.210340371976184f * static_cast<float>(k & 127) / 128.0f);
              const float a = t * f;
              e[j] = k < 128 ? cosf(a) : sinf(a);
            }
            xv = make_float4(e[0], e[1], e[2], e[3]);
          } else {
            xv = reinterpret_cast<const float4*>(x + static_cast<long long>((r_base + i) % x_rows) * K + k0)[c];
          }
          if (act_in != LDC_ACT_NONE && !(FULL && act_in == LDC_ACT_IN_TIMESTEP_SINCOS)) {
            xv.x = ldc_apply_act(xv.x, act_in);
            xv.y = ldc_apply_act(xv.y, act_in);
            xv.z = ldc_apply_act(xv.z, act_in);
            xv.w = ldc_apply_act(xv.w, act_in);
          }
          reinterpret_cast<float4*>(xs + i * kc)[c] = xv;
        }
        __syncthreads();
      }
      if (n0 < N) {
        for (int c = lane; c < nv4; c += 64) {  // (not unrolled: the asm in ls_first_read is convergent and the trip count is per lane)
          float4 w[LS_CPW];
#pragma unroll
          for (int j = 0; j < LS_CPW; ++j) {  // weights are streamed once: non-temporal
            typedef float ls_f32x4 __attribute__((ext_vector_type(4)));
            const ls_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const ls_f32x4*>(wr[j] + k0) + c);
            w[j] = make_float4(t.x, t.y, t.z, t.w);
          }
#pragma unroll
          for (int i = 0; i < LS_ROWS; ++i) {
            if (i < rows_here) {
              float4 xv = reinterpret_cast<const float4*>(xs + i * kc)[c];
              ls_first_read(xv);
#pragma unroll
              for (int j = 0; j < LS_CPW; ++j) acc[j][i] += (w[j].x * xv.x + w[j].y * xv.y) + (w[j].z * xv.z + w[j].w * xv.w);
            }
          }
        }
      }
    }
    if (n0 >= N) continue;
#pragma unroll
    for (int j = 0; j < LS_CPW; ++j) {
      const int n = n0 + j;
#pragma unroll
      for (int i = 0; i < LS_ROWS; ++i) {
        const float sum = wave_sum(acc[j][i]);
        const int r = r_base + i;
        if (lane == 0 && i < rows_here && n < N) {
          float v = sum + (bias ? bias[n] : 0.f);
          v = ldc_apply_act(v, act_out);
          if (add) v += add[static_cast<long long>(r % add_rows) * N + n];
          if (FULL && mod) {  // v * (1 + scale) + shift, scale | shift = the two halves of a [mod_rows][2 N] table; no contraction: the
            const float* mr = mod + static_cast<long long>(r % mod_rows) * 2 * N;  // same bits as temb_modulate_kernel
            v = modulate_nocontract(v, mr[n], mr[N + n]);
          }
          y[static_cast<long long>(r) * N + n] = v;
        }
      }
    }
  }
}

// ---- the same product for MORE rows than one workgroup of the kernel above holds (9 .. : the conditioning batch of a sampler chunk is 20 rows
// per member) on the fp32 matrix cores.  The VALU kernel streams W once per 8 rows and is VALU-bound there (20 rows x 58 368 x 1 536: 385 us,
// 0.9 TB/s); here a wave owns 16 output columns and all rows of its row group (RT tiles of 16), W goes HBM -> registers exactly in the
// B-operand layout of v_mfma_f32_16x16x4_f32 (lane = (column, k quarter): one 16-byte load = four k-steps), x is staged per K chunk in LDS
// (row stride KC + 4 floats: the 16 rows of an A read fall on different banks) - exact fp32 products, fp32 accumulation.
constexpr int LM_KC = 512;          // K chunk staged in LDS
constexpr int LM_LD = LM_KC + 4;    // its row stride in floats
typedef float lm_f32x4 __attribute__((ext_vector_type(4)));
template <int RT, int U>
__device__ __forceinline__ void lm_steps(const lm_f32x4* __restrict__ wp, const float* __restrict__ xa, lm_f32x4 (&acc)[RT]) {
  lm_f32x4 w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(wp + 4 * u);
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float4 a = *reinterpret_cast<const float4*>(xa + rt * 16 * LM_LD + 16 * u);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[u].x, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[u].y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[u].z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[u].w, acc[rt], 0, 0, 0);
    }
}

template <int RT, int IT>  // RT row tiles of 16 per workgroup, IT column groups of 64 per workgroup (the staged x is reused IT times)
__global__ __launch_bounds__(256) void linear_rows_mfma_kernel(const float* __restrict__ x, int x_rows, const float* __restrict__ W,
                                                               const float* __restrict__ bias, const float* __restrict__ add, int add_rows,
                                                               float* __restrict__ y, int rows, int N, int K, int act_in, int act_out) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [16 RT][LM_LD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int r_base = blockIdx.y * 16 * RT;
  const int rows_here = rows - r_base < 16 * RT ? rows - r_base : 16 * RT;
  lm_f32x4 acc[IT][RT];
#pragma unroll
  for (int it = 0; it < IT; ++it)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[it][rt] = lm_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += LM_KC) {
    const int kn = K - k0 < LM_KC ? K - k0 : LM_KC;  // multiple of 64
    const int nv4 = kn >> 2;
    if (k0) __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * RT * nv4; idx += 256) {
      const int i = idx / nv4, c = idx - i * nv4;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the last one multiply zeros
      if (i < rows_here) {
        xv = reinterpret_cast<const float4*>(x + static_cast<long long>((r_base + i) % x_rows) * K + k0)[c];
        if (act_in != LDC_ACT_NONE) {
          xv.x = ldc_apply_act(xv.x, act_in);
          xv.y = ldc_apply_act(xv.y, act_in);
          xv.z = ldc_apply_act(xv.z, act_in);
          xv.w = ldc_apply_act(xv.w, act_in);
        }
      }
      reinterpret_cast<float4*>(xs + i * LM_LD)[c] = xv;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int n0 = ((blockIdx.x * IT + it) * 4 + wave) * 16;
      if (n0 < N) {  // wave-uniform
        const int n = n0 + col < N ? n0 + col : N - 1;  // clamped: columns past N are computed and dropped
        const lm_f32x4* wp = reinterpret_cast<const lm_f32x4*>(W + static_cast<long long>(n) * K + k0) + kq;
        const float* xa = xs + col * LM_LD + 4 * kq;
        const int ns = kn >> 4;  // 16 k per step: one 16-byte load of W per lane, RT 16-byte LDS reads, 4 RT MFMAs
        int s0 = 0;
        for (; s0 + 8 <= ns; s0 += 8) lm_steps<RT, 8>(wp + 4 * s0, xa + 16 * s0, acc[it]);  // eight loads in flight per lane
        if (s0 < ns) lm_steps<RT, 4>(wp + 4 * s0, xa + 16 * s0, acc[it]);                   // (ns is a multiple of 4)
      }
    }
  }
  // accumulator layout: lane (col, kq) holds rows 16 rt + 4 kq + i (i = 0..3) of output column n0 + col
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int n = ((blockIdx.x * IT + it) * 4 + wave) * 16 + col;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 16 * rt + 4 * kq + i;
        if (rl >= rows_here) continue;
        const int r = r_base + rl;
        float v = acc[it][rt][i] + bv;
        v = ldc_apply_act(v, act_out);
        if (add) v += add[static_cast<long long>(r % add_rows) * N + n];
        y[static_cast<long long>(r) * N + n] = v;
      }
  }
}

template <int LS_CPW>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, int x_rows,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ add, int add_rows,
                                                           float* __restrict__ y, int rows, int N, int K, int kc, int act_in,
                                                           int act_out, int iters, const float* __restrict__ mod, int mod_rows) {
  linear_small_body<LS_CPW, LS_CPW == 1>(x, x_rows, W, bias, add, add_rows, y, rows, N, K, kc, act_in, act_out, iters, mod, mod_rows);
}

// up to LDC_LINEAR_SMALL_MAX_GROUPED independent small linears in one launch: blockIdx.z picks the problem
struct LSGroup {
  ldc_linear_small_problem p[LDC_LINEAR_SMALL_MAX_GROUPED];
};

__global__ __launch_bounds__(256) void linear_small_grouped_kernel(LSGroup g) {
  const ldc_linear_small_problem& q = g.p[blockIdx.z];
  if (static_cast<int>(blockIdx.x) * 4 >= q.N || static_cast<int>(blockIdx.y) * LS_ROWS >= q.rows) return;  // block-uniform
  const int kc = q.K < LS_KC ? q.K : LS_KC;
  linear_small_body<1>(q.x, q.x_rows, q.W, q.bias, q.add, q.add_rows, q.y, q.rows, q.N, q.K, kc, q.act_in, q.act_out, 1);
}

}  // namespace

extern "C" int ldc_layernorm_mod2(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs, int ldy,
                                  long long y_bs, const float* scale, const float* shift, int split_row,
                                  const float* scale2, const float* shift2, int mod_bs, int mode, float eps,
                                  int out_split, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  if (B <= 0 || rows <= 0 || D <= 0 || split_row < 0) return LDC_ERR_ARG;
  if ((D & 3) || D > 64 * 4 * LN_MAX_V4) return LDC_ERR_UNSUPPORTED;
  if ((ldx & 3) || (ldy & 3) || (x_bs & 3) || (y_bs & 3) || (mod_bs & 3)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(y);
  if (scale) LDC_CHECK_ALIGN16(scale);
  if (shift) LDC_CHECK_ALIGN16(shift);
  if (scale2) LDC_CHECK_ALIGN16(scale2);
  if (shift2) LDC_CHECK_ALIGN16(shift2);
  if (mode != 0 && mode != 1) return LDC_ERR_UNSUPPORTED;
  if (out_split && ((D & 7) || (ldy & 7) || (y_bs & 7) || (reinterpret_cast<unsigned long long>(y) & 31ull))) return LDC_ERR_ALIGN;
  dim3 grid(ldc_cdiv(rows, 4), B);
  hipLaunchKernelGGL(layernorm_mod_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), const_cast<float*>(x), y, rows, D,
                     ldx, x_bs, ldy, y_bs, scale, shift, split_row, scale2, shift2, mod_bs, mode, eps, out_split,
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), 0, 0LL, 0);
  return ldc_launch_status();
}

extern "C" int ldc_gate_residual_layernorm(float* resid, const float* yv, const float* gate, float* out, int B, int rows, int D,
                                           int ld_res, long long res_bs, int ld_y, long long y_bs, int gate_bs, int ld_out,
                                           long long out_bs, const float* weight, const float* bias, float eps, int out_split,
                                           void* stream) {
  LDC_CHECK_PTR(resid);
  LDC_CHECK_PTR(yv);
  LDC_CHECK_PTR(gate);
  LDC_CHECK_PTR(out);
  if (B <= 0 || rows <= 0 || D <= 0) return LDC_ERR_ARG;
  if ((D & 3) || D > 64 * 4 * LN_MAX_V4) return LDC_ERR_UNSUPPORTED;
  if ((ld_res & 3) || (ld_y & 3) || (ld_out & 3) || (res_bs & 3) || (y_bs & 3) || (out_bs & 3) || (gate_bs & 3)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(resid);
  LDC_CHECK_ALIGN16(yv);
  LDC_CHECK_ALIGN16(gate);
  LDC_CHECK_ALIGN16(out);
  if (weight) LDC_CHECK_ALIGN16(weight);
  if (bias) LDC_CHECK_ALIGN16(bias);
  if (out_split && ((D & 7) || (ld_out & 7) || (out_bs & 7) || (reinterpret_cast<unsigned long long>(out) & 31ull))) return LDC_ERR_ALIGN;
  dim3 grid(ldc_cdiv(rows, 4), B);
  hipLaunchKernelGGL(layernorm_mod_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), resid, out, rows, D, ld_res, res_bs,
                     ld_out, out_bs, weight, bias, rows, static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), 0, 1,
                     eps, out_split, yv, gate, ld_y, y_bs, gate_bs);
  return ldc_launch_status();
}

extern "C" int ldc_layernorm_mod(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs, int ldy,
                                 long long y_bs, const float* scale, const float* shift, int mod_bs, int mode,
                                 float eps, int out_split, void* stream) {
  return ldc_layernorm_mod2(x, y, B, rows, D, ldx, x_bs, ldy, y_bs, scale, shift, rows, nullptr, nullptr, mod_bs, mode, eps,
                            out_split, stream);
}

extern "C" int ldc_qk_rmsnorm_rope(float* q, float* k, int B, int row0, int rows, int H, int ld, long long bs,
                                   const float* wq, const float* wk, float eps, const float* cos_tab,
                                   const float* sin_tab, void* stream) {
  LDC_CHECK_PTR(q);
  LDC_CHECK_PTR(k);
  LDC_CHECK_PTR(wq);
  LDC_CHECK_PTR(wk);
  if (B <= 0 || rows <= 0 || H <= 0 || row0 < 0) return LDC_ERR_ARG;
  if ((cos_tab == nullptr) != (sin_tab == nullptr)) return LDC_ERR_ARG;
  if ((ld & 1) || (bs & 1)) return LDC_ERR_ALIGN;
  const long long items = static_cast<long long>(rows) * H * 2;
  dim3 grid(ldc_cdiv(items, 4), B);
  hipLaunchKernelGGL(qk_rmsnorm_rope_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), q, k, row0, rows, H,
                     ld, bs, wq, wk, eps, cos_tab, sin_tab);
  return ldc_launch_status();
}

extern "C" int ldc_mean_rows(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  if (B <= 0 || rows <= 0 || D <= 0) return LDC_ERR_ARG;
  dim3 grid(ldc_cdiv(D, 64), B);
  hipLaunchKernelGGL(mean_rows_kernel, grid, dim3(1024), 0, static_cast<hipStream_t>(stream), x, y, rows, D, ldx, x_bs);
  return ldc_launch_status();
}

extern "C" int ldc_mean_rows_split(const float* x, float* y, float* x_split, int B, int rows, int D, int ldx,
                                   long long x_bs, int lds, long long s_bs, int fmt, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  LDC_CHECK_PTR(x_split);
  if (B <= 0 || rows <= 0 || D <= 0) return LDC_ERR_ARG;
  if ((D & 7) || (ldx & 1) || (x_bs & 1) || (lds & 7) || (s_bs & 7) || (reinterpret_cast<unsigned long long>(x) & 7ull) ||
      (reinterpret_cast<unsigned long long>(x_split) & 31ull))
    return LDC_ERR_ALIGN;
  if (x == x_split) return LDC_ERR_ARG;  // other workgroups' rows would be read after being overwritten
  if (fmt != LDC_FMT_SPLIT && fmt != LDC_FMT_BF16) return LDC_ERR_ARG;
  dim3 grid(ldc_cdiv(D, 128), B);
  hipLaunchKernelGGL(mean_rows_split_kernel, grid, dim3(1024), 0, static_cast<hipStream_t>(stream), x, y, x_split, rows, D,
                     ldx, x_bs, lds, s_bs, fmt);
  return ldc_launch_status();
}

extern "C" int ldc_gate_residual(const float* resid, const float* y, const float* gate, float* out, int B, int rows,
                                 int D, int ld_res, long long res_bs, int ld_y, long long y_bs, int gate_bs,
                                 void* stream) {
  LDC_CHECK_PTR(resid);
  LDC_CHECK_PTR(y);
  LDC_CHECK_PTR(gate);
  LDC_CHECK_PTR(out);
  if (B <= 0 || rows <= 0 || D <= 0) return LDC_ERR_ARG;
  if ((D & 3) || (ld_res & 3) || (ld_y & 3) || (res_bs & 3) || (y_bs & 3) || (gate_bs & 3)) return LDC_ERR_ALIGN;
  if (rows > 65535 || B > 65535) return LDC_ERR_UNSUPPORTED;
  dim3 grid(ldc_cdiv(D / 4, 256), rows, B);
  hipLaunchKernelGGL(gate_residual_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), resid, y, gate, out,
                     rows, D, ld_res, res_bs, ld_y, y_bs, gate_bs);
  return ldc_launch_status();
}

static int linear_small_impl(const float* x, int x_rows, const float* W, const float* bias, const float* add, int add_rows,
                             const float* mod, int mod_rows, float* y, int rows, int N, int K, int act_in, int act_out,
                             void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(W);
  LDC_CHECK_PTR(y);
  if (rows <= 0 || N <= 0 || K <= 0 || x_rows <= 0) return LDC_ERR_ARG;
  if (add && add_rows <= 0) return LDC_ERR_ARG;
  if (mod && mod_rows <= 0) return LDC_ERR_ARG;
  if (K & 3) return LDC_ERR_ALIGN;
  if (act_in == LDC_ACT_IN_TIMESTEP_SINCOS) {
    if (K != 256) return LDC_ERR_ARG;  // x: one float per row, any 4-byte alignment
  } else {
    LDC_CHECK_ALIGN16(x);
  }
  LDC_CHECK_ALIGN16(W);
  const int kc = K < LS_KC ? K : LS_KC;
  const int rows_max = rows < LS_ROWS ? rows : LS_ROWS;
  const size_t lds = static_cast<size_t>(rows_max) * kc * sizeof(float);
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_small_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LS_ROWS * LS_KC * static_cast<int>(sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_small_kernel<LS_CPW_MAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LS_ROWS * LS_KC * static_cast<int>(sizeof(float)));
    return true;
  }();
  (void)attr_set;
  const bool plain = mod == nullptr && act_in != LDC_ACT_IN_TIMESTEP_SINCOS;  // the wide instantiations have neither feature
  if (plain && rows > LS_ROWS && N >= 4096 && (K & 63) == 0) {  // more rows than one workgroup of the VALU kernel holds: fp32 matrix cores
    const int rt = rows <= 16 ? 1 : 2;
    const int col_groups = ldc_cdiv(N, 64);
    // column groups per workgroup (they share the staged x): the count that needs the fewest rounds of resident workgroups (LDS: 2 per CU at
    // RT = 2, 4 at RT = 1) x groups per workgroup; ties go to the larger count (x is staged less often: 20 x 58 368 x 1 536 takes 120.9 us
    // at 1, 104.6 at 2, 140.5 at 3, 155.0 at 4)
    const int row_groups_lm = ldc_cdiv(rows, 16 * rt);
    const long long slots = (rt == 2 ? 2 : 4) * 256LL;
    int iters = 1;
    long long best_cost = -1;
    for (int c = 1; c <= 4; ++c) {
      const long long wgs = static_cast<long long>(ldc_cdiv(col_groups, c)) * row_groups_lm;
      const long long cost = ((wgs + slots - 1) / slots) * c;
      if (best_cost < 0 || cost <= best_cost) { best_cost = cost; iters = c; }
    }
    static const char* const force_lm_iters = LDC_AB_GETENV("LDC_LINEAR_MFMA_ITERS");  // measurement aid, read once
    if (force_lm_iters && atoi(force_lm_iters) >= 1 && atoi(force_lm_iters) <= 4) iters = atoi(force_lm_iters);
    dim3 grid(ldc_cdiv(col_groups, iters), row_groups_lm);
    const int lm_lds = 16 * rt * LM_LD * static_cast<int>(sizeof(float));
    static const bool lm_attr_set = [] {  // once per process; thread-safe (C++11 static initialisation)
      const int b1 = 16 * LM_LD * static_cast<int>(sizeof(float)), b2 = 2 * b1;
      const auto set = [](auto kern, int bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
      set(linear_rows_mfma_kernel<1, 1>, b1); set(linear_rows_mfma_kernel<1, 2>, b1); set(linear_rows_mfma_kernel<1, 3>, b1); set(linear_rows_mfma_kernel<1, 4>, b1);
      set(linear_rows_mfma_kernel<2, 1>, b2); set(linear_rows_mfma_kernel<2, 2>, b2); set(linear_rows_mfma_kernel<2, 3>, b2); set(linear_rows_mfma_kernel<2, 4>, b2);
      return true;
    }();
    (void)lm_attr_set;
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, grid, dim3(256), lm_lds, static_cast<hipStream_t>(stream), x, x_rows, W, bias, add, add_rows, y, rows, N, K,
                         act_in, act_out);
    };
    if (rt == 1) {
      if (iters == 4) launch(linear_rows_mfma_kernel<1, 4>);
      else if (iters == 3) launch(linear_rows_mfma_kernel<1, 3>);
      else if (iters == 2) launch(linear_rows_mfma_kernel<1, 2>);
      else launch(linear_rows_mfma_kernel<1, 1>);
    } else {
      if (iters == 4) launch(linear_rows_mfma_kernel<2, 4>);
      else if (iters == 3) launch(linear_rows_mfma_kernel<2, 3>);
      else if (iters == 2) launch(linear_rows_mfma_kernel<2, 2>);
      else launch(linear_rows_mfma_kernel<2, 1>);
    }
    return ldc_launch_status();
  }
  const int row_groups = ldc_cdiv(rows, LS_ROWS);
  if (plain && static_cast<long long>(ldc_cdiv(N, 4 * LS_CPW_MAX)) * row_groups >= 1024) {  // >= 4 workgroups per CU
    int iters = 1;
    if (K <= LS_KC) {  // one staged chunk: walk several column groups per workgroup, keep >= 4 workgroups per CU
      const long long groups = static_cast<long long>(ldc_cdiv(N, 4 * LS_CPW_MAX)) * row_groups;
      iters = groups >= 2048 ? 4 : 1;  // measured (tools/gemv_bench.py, 38 D x D): 78 us at 1, 73 at 4, slower at 3 / 8 / 16
      static const char* const force_iters = LDC_AB_GETENV("LDC_LINEAR_SMALL_ITERS");  // measurement aid, read once
      if (force_iters && atoi(force_iters) > 0) iters = atoi(force_iters);
    }
    dim3 grid(ldc_cdiv(ldc_cdiv(N, 4 * LS_CPW_MAX), iters), row_groups);
    hipLaunchKernelGGL(linear_small_kernel<LS_CPW_MAX>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), x, x_rows, W,
                       bias, add, add_rows, y, rows, N, K, kc, act_in, act_out, iters, mod, mod_rows);
  } else {
    dim3 grid(ldc_cdiv(N, 4), row_groups);
    hipLaunchKernelGGL(linear_small_kernel<1>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), x, x_rows, W, bias, add,
                       add_rows, y, rows, N, K, kc, act_in, act_out, 1, mod, mod_rows);
  }
  return ldc_launch_status();
}

extern "C" int ldc_linear_small(const float* x, int x_rows, const float* W, const float* bias, const float* add,
                                int add_rows, float* y, int rows, int N, int K, int act_in, int act_out,
                                void* stream) {
  return linear_small_impl(x, x_rows, W, bias, add, add_rows, nullptr, 1, y, rows, N, K, act_in, act_out, stream);
}

extern "C" int ldc_linear_small_mod(const float* x, int x_rows, const float* W, const float* bias, const float* add,
                                    int add_rows, const float* mod, int mod_rows, float* y, int rows, int N, int K, int act_in,
                                    int act_out, void* stream) {
  LDC_CHECK_PTR(mod);
  return linear_small_impl(x, x_rows, W, bias, add, add_rows, mod, mod_rows, y, rows, N, K, act_in, act_out, stream);
}

extern "C" int ldc_linear_small_grouped(const ldc_linear_small_problem* problems, int n, void* stream) {
  LDC_CHECK_PTR(problems);
  if (n < 1 || n > LDC_LINEAR_SMALL_MAX_GROUPED) return LDC_ERR_ARG;
  LSGroup g{};
  int gx = 0, gy = 0, kc_max = 0, rows_max = 0;
  for (int i = 0; i < n; ++i) {
    const ldc_linear_small_problem& q = problems[i];
    LDC_CHECK_PTR(q.x);
    LDC_CHECK_PTR(q.W);
    LDC_CHECK_PTR(q.y);
    if (q.rows <= 0 || q.N <= 0 || q.K <= 0 || q.x_rows <= 0) return LDC_ERR_ARG;
    if (q.add && q.add_rows <= 0) return LDC_ERR_ARG;
    if (q.K & 3) return LDC_ERR_ALIGN;
    if (q.act_in == LDC_ACT_IN_TIMESTEP_SINCOS) {
      if (q.K != 256) return LDC_ERR_ARG;
    } else {
      LDC_CHECK_ALIGN16(q.x);
    }
    LDC_CHECK_ALIGN16(q.W);
    for (int j = 0; j < i; ++j)  // problems of one launch run concurrently: no output may be another's input
      if (problems[j].y == q.x || problems[j].y == q.add || q.y == problems[j].x || q.y == problems[j].add || q.y == problems[j].y)
        return LDC_ERR_ARG;
    g.p[i] = q;
    const int bx = ldc_cdiv(q.N, 4), by = ldc_cdiv(q.rows, LS_ROWS);
    gx = bx > gx ? bx : gx;
    gy = by > gy ? by : gy;
    const int kc = q.K < LS_KC ? q.K : LS_KC;
    kc_max = kc > kc_max ? kc : kc_max;
    const int r = q.rows < LS_ROWS ? q.rows : LS_ROWS;
    rows_max = r > rows_max ? r : rows_max;
  }
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_small_grouped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LS_ROWS * LS_KC * static_cast<int>(sizeof(float)));
    return true;
  }();
  (void)attr_set;
  const size_t lds = static_cast<size_t>(rows_max) * kc_max * sizeof(float);
  hipLaunchKernelGGL(linear_small_grouped_kernel, dim3(gx, gy, n), dim3(256), lds, static_cast<hipStream_t>(stream), g);
  return ldc_launch_status();
}
