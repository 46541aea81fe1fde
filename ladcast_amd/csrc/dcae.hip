// HBM-bound kernels of the DCAE encoder/decoder in the MI355X-native NHWC layout
// (channels innermost: every per-pixel op is a contiguous, float4-coalesced row).
// Dense 3x3 sphere convolutions and all 1x1 convs / Linears are the MFMA GEMM of gemm_f32.hip.
#include "common.h"

namespace {

// same rule as gemm_f32.hip::sphere_src_pixel (models/sphere_conv.py:62-129,174-192)
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

__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// Depthwise sphere conv (k = 3 or 5), weights [k*k][C], optional bias; GLU form:
//   y[c] = d[c] * silu(d[c + C/2]), c < C/2      (GLUMBConv, models/DCAE.py:311-313)
template <int KS, bool GLU>
__global__ __launch_bounds__(256) void sphere_dwconv_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ y, int H,
                                                            int W, int C, int ldx, int ldy, int out_fmt) {
  const int pix = blockIdx.y;  // b*H*W + h*W + w
  const int hw = H * W;
  const int bimg = pix / hw;
  const int rem = pix - bimg * hw;
  const int h = rem / W, w = rem - h * W;
  const int nvec = (GLU ? C / 2 : C) >> 2;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  const int c0 = v * 4;
  const int half = C / 2;
  float4 a0 = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (GLU && bias) a1 = *reinterpret_cast<const float4*>(bias + half + c0);
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const int src = bimg * hw + sphere_src_pixel(h, w, ky, kx, H, W, KS);
      const float* xp = x + static_cast<long long>(src) * ldx;
      const float* wp = wt + static_cast<long long>(ky * KS + kx) * C;
      a0 = f4_fma(*reinterpret_cast<const float4*>(xp + c0), *reinterpret_cast<const float4*>(wp + c0), a0);
      if (GLU) a1 = f4_fma(*reinterpret_cast<const float4*>(xp + half + c0), *reinterpret_cast<const float4*>(wp + half + c0), a1);
    }
  if (GLU) {
    a0.x *= ldc_silu(a1.x);
    a0.y *= ldc_silu(a1.y);
    a0.z *= ldc_silu(a1.z);
    a0.w *= ldc_silu(a1.w);
  }
  if (out_fmt != LDC_FMT_F32)  // operand rows of the next pointwise conv (gemm_bf16x3_v3.hip)
    ldc_store_fmt4(reinterpret_cast<unsigned char*>(y + static_cast<long long>(pix) * ldy), c0, out_fmt, a0.x, a0.y, a0.z, a0.w);
  else
    *reinterpret_cast<float4*>(y + static_cast<long long>(pix) * ldy + c0) = a0;
}

// The same depthwise conv, register-tiled along the image row: one thread = one float4 of channels x 8 consecutive pixels.  For a
// fixed output row the sphere padding rule is a per-kernel-row (source row, column shift, left-right flip), so kernel row ky needs
// the contiguous (mod W) run of 8 + KS - 1 source pixels starting at w0 - p - shift, and a flipped kernel row reads it with kx
// mirrored: 12 (10) float4 loads per kernel row feed 8 x KS multiply-adds, 4.7x (3.7x) fewer loads per output than one load per
// tap and output - the untiled kernel is bound by exactly those (L1 / L2 request rate).  Needs W >= 8 + KS - 1 (one wrap).
template <int KS, bool GLU>
__global__ __launch_bounds__(256) void sphere_dwconv_row_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                                int C, int ldx, int ldy, int out_fmt, long long total) {
  constexpr int P = KS / 2, PW = 8 + 2 * P;
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int half = C >> 1;
  const int nvec = (GLU ? half : C) >> 2;
  const int c0 = static_cast<int>(idx % nvec) * 4;
  const long long seg = idx / nvec;
  const int nseg = (W + 7) >> 3;
  const int w0 = static_cast<int>(seg % nseg) * 8;
  const long long bh = seg / nseg;
  const int h = static_cast<int>(bh % H);
  const float* img = x + (bh - h) * W * static_cast<long long>(ldx);  // first pixel of this image
  float4 acc[8], gate[8];
#pragma unroll
  for (int gi = 0; gi < (GLU ? 2 : 1); ++gi) {
    const int g = GLU ? 1 - gi : 0;  // GLU: the gate half first
    const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + g * half + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = b4;
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky) {  // not unrolled: one kernel row's patch (12 float4) in registers at a time
      int r = h + ky - P, shift = 0;
      if (r < 0) {
        r = -1 - r;
        shift = W >> 1;
      } else if (r >= H) {
        r = 2 * H - 1 - r;
        shift = W >> 1;
      }
      // a kernel row that reaches over the pole is applied left-right flipped: mirror the WEIGHT index, not the patch index
      const bool flip = (h == 0 && ky < P) || (h == H - 1 && ky >= KS - P);
      int col0 = w0 - P - shift;
      if (col0 < 0) col0 += W;
      const float* rowp = img + static_cast<long long>(r) * W * ldx + g * half + c0;
      float4 pt[PW];
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        int col = col0 + j;
        if (col >= W) col -= W;
        pt[j] = *reinterpret_cast<const float4*>(rowp + static_cast<long long>(col) * ldx);
      }
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int kw = flip ? KS - 1 - kx : kx;
        const float4 w4 = *reinterpret_cast<const float4*>(wt + static_cast<long long>(ky * KS + kw) * C + g * half + c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f4_fma(pt[i + kx], w4, acc[i]);
      }
    }
    if (GLU && gi == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) gate[i] = make_float4(ldc_silu(acc[i].x), ldc_silu(acc[i].y), ldc_silu(acc[i].z), ldc_silu(acc[i].w));
    }
  }
  float* yrow = y + ((bh * W) + w0) * static_cast<long long>(ldy);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (w0 + i >= W) break;
    float4 o = acc[i];
    if (GLU) {
      o.x *= gate[i].x;
      o.y *= gate[i].y;
      o.z *= gate[i].z;
      o.w *= gate[i].w;
    }
    float* yr = yrow + static_cast<long long>(i) * ldy;
    if (out_fmt != LDC_FMT_F32) ldc_store_fmt4(reinterpret_cast<unsigned char*>(yr), c0, out_fmt, o.x, o.y, o.z, o.w);
    else *reinterpret_cast<float4*>(yr + c0) = o;
  }
}

// grouped 1x1 conv, 32 -> 32 channels per group: y[m][g*32+o] = sum_i W[g*32+o][i] * x[m][g*32+i]
// block = 64 pixels x one group: thread (o, pixel phase) keeps its output channel's 32 weights in registers and reads the inputs
// from LDS (one broadcast read per multiply-add) - the weights are fetched once per 64 pixels
__global__ __launch_bounds__(256) void grouped_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                              float* __restrict__ y, int M, int ldx, int ldy) {
  __shared__ float xs[64][32];
  const int g = blockIdx.y;
  const int m0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const int pm = t >> 5, o = t & 31;
  float w[32];
#pragma unroll
  for (int i = 0; i < 32; i += 4) {
    const float4 v = *reinterpret_cast<const float4*>(wt + static_cast<long long>(g * 32 + o) * 32 + i);
    w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int m = m0 + 8 * s + pm;
    xs[8 * s + pm][o] = (m < M) ? x[static_cast<long long>(m) * ldx + g * 32 + o] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int m = m0 + 8 * s + pm;
    if (m >= M) break;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = fmaf(w[i], xs[8 * s + pm][i], acc);
    y[static_cast<long long>(m) * ldy + g * 32 + o] = acc;
  }
}

// ReLU linear attention of one (batch, 96-channel group): q = ch 0..31, k = 32..63, v = 64..95
//   KV[c][j] = sum_p vpad[p][c] relu(k[p][j]), c = 0..32 (row 32 = ones);  out[p][c] = sum_j KV[c][j] relu(q[p][j])
//   y[p][c] = out[p][c] / (out[p][32] + eps)                 (models/DCAE.py:158-175,239-249; fp32)
// ONE launch, one 16-wave workgroup per (batch, group), both contractions on the fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulation - the reference's fp32 island stays fp32):
//   phase 1: wave w takes the pixel pairs w, w + 16, ...: one MFMA per pair, A[c][k] = v[p_k][c], B[k][j] = relu(k[p_k][j]) - lanes
//            0-31 / 32-63 hold the two pixels, one coalesced 128-byte load per operand; the ones row (sum of relu(k)) rides along as
//            a per-lane sum.  The 16 partial 33 x 32 matrices are added in wave order through LDS (deterministic).
//   phase 2: per 32 pixels, out^T[c][p] = sum_j KV[c][j] relu(q[p][j]) as 16 MFMAs: lane (p, half h) loads q[p][16 h .. 16 h + 15]
//            (64 contiguous bytes) and MFMA kk contracts j = kk (h = 0) with j = 16 + kk (h = 1) - any pairing of j with the two
//            k slots works when A uses the same one; the accumulator layout gives a lane 4 x 4 consecutive channels of its pixel
//            -> float4 (or split-format) stores; the denominator is a 16-term dot product per lane + one cross-half add.
// A frame has 30 / 62 groups: 30 - 62 busy CUs, each streaming its group's 0.2 - 0.9 MB once per phase (8 / 6 us at one 30 x 60 /
// 15 x 30 frame, against 70 / 39 us for the earlier slice-partials / apply launches); from 5 - 9 frames on every CU has a workgroup.
constexpr int KV_N = 33 * 32;
constexpr int RLA_WAVES = 16;

__global__ __launch_bounds__(RLA_WAVES * 64) void relu_linear_attn_kernel(const float* __restrict__ qkv, float* __restrict__ y, int P,
                                                                          int ldq, int ldy, float eps, int out_fmt) {
  extern __shared__ __attribute__((aligned(16))) float rla_smem[];
  float* red = rla_smem;                      // [RLA_WAVES][KV_N] partial KV
  float* kv = rla_smem + RLA_WAVES * KV_N;    // [33][32]
  const int g = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, h = lane >> 5;
  const float* base = qkv + static_cast<long long>(b) * P * ldq + g * 96;
  // ---- phase 1 ----
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ksum = 0.f;
  const int npair = (P + 1) >> 1;
  constexpr int UN = 16;  // pairs per trip: their 2 UN loads are in flight together (the loop is bound by memory latency, not by the MFMAs)
  for (int i0 = wave; i0 < npair; i0 += UN * RLA_WAVES) {
    float kk[UN], vv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = 2 * (i0 + u * RLA_WAVES) + h;
      kk[u] = 0.f;
      vv[u] = 0.f;
      if (p < P) {
        const float* row = base + static_cast<long long>(p) * ldq;
        kk[u] = fmaxf(row[32 + n], 0.f);
        vv[u] = row[64 + n];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      ksum += kk[u];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[u], kk[u], acc, 0, 0, 0);
    }
  }
  ksum += __shfl_xor(ksum, 32, 64);
  {
    float* mine = red + wave * KV_N;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r >> 2) * 8 + h * 4 + (r & 3)) * 32 + n] = acc[r];  // KV[c][j = n]
    if (h == 0) mine[32 * 32 + n] = ksum;
  }
  __syncthreads();
  for (int e = tid; e < KV_N; e += RLA_WAVES * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < RLA_WAVES; ++w) s += red[w * KV_N + e];
    kv[e] = s;
  }
  __syncthreads();
  // ---- phase 2 ----
  float ka[16], kd[16];  // A operands: KV[c = n][16 h + kk]; denominator weights KV[32][16 h + kk]
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    ka[kk] = kv[n * 32 + 16 * h + kk];
    kd[kk] = kv[32 * 32 + 16 * h + kk];
  }
  const int ntile = (P + 31) >> 5;
  auto load_q = [&](int t, float4 (&dst)[4]) {  // q[p][16 h .. 16 h + 15] of pixel p = 32 t + n (clamped)
    const int p = 32 * t + n;
    const float* qp = base + static_cast<long long>(p < P ? p : P - 1) * ldq + 16 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const float4*>(qp + 4 * i);
  };
  float4 qn[4];
  if (wave < ntile) load_q(wave, qn);
  for (int t = wave; t < ntile; t += RLA_WAVES) {
    const int p = 32 * t + n;
    float q[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      q[4 * i] = fmaxf(qn[i].x, 0.f);
      q[4 * i + 1] = fmaxf(qn[i].y, 0.f);
      q[4 * i + 2] = fmaxf(qn[i].z, 0.f);
      q[4 * i + 3] = fmaxf(qn[i].w, 0.f);
    }
    if (t + RLA_WAVES < ntile) load_q(t + RLA_WAVES, qn);  // the next tile's loads fly under this tile's MFMAs
    float den = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) den = fmaf(kd[kk], q[kk], den);
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.0f / (den + eps);
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk], q[kk], o, 0, 0, 0);  // out^T[c][p = n]
    if (p < P) {
      float* yrow = y + (static_cast<long long>(b) * P + p) * ldy;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int ch = g * 32 + 8 * r4 + 4 * h;
        const float o0 = o[4 * r4] * inv, o1 = o[4 * r4 + 1] * inv, o2 = o[4 * r4 + 2] * inv, o3 = o[4 * r4 + 3] * inv;
        if (out_fmt != LDC_FMT_F32) ldc_store_fmt4(reinterpret_cast<unsigned char*>(yrow), ch, out_fmt, o0, o1, o2, o3);
        else *reinterpret_cast<float4*>(yrow + ch) = make_float4(o0, o1, o2, o3);
      }
    }
  }
}

// The same attention cut into pixel slices (round 3): a frame has 30 / 62 groups, so the one-workgroup-per-group kernel above keeps
// 30 - 62 CUs busy and every wave walks its pixels through 4 + 4 dependent rounds of memory latency (50 / 15 us per call inside the
// decoder, profiles/r02_m_dcae_decode_1frame_timeline_after.txt).  Here a 4-wave workgroup owns 128 pixels of a group:
//   launch 1: its partial KV (33 x 32) over those pixels - ONE round of 16 pixel pairs per wave - to `part`;
//   launch 2: KV = the group's partials added in slice order (fixed order, independent of the batch size: a frame's result does not
//             depend on what it is batched with), then out / denominator for the same 128 pixels, one 32-pixel tile per wave.
// 450 / 248 workgroups per frame, one memory round trip per phase.  `part`: B x groups x slices x 33 x 32 floats of caller scratch.
constexpr int RLA_SL_WAVES = 4;
constexpr int RLA_SL_PIX = 128;
constexpr int RLA_SLICED_MIN_P = 1024;

__global__ __launch_bounds__(RLA_SL_WAVES * 64) void relu_linear_attn_kv_slice_kernel(const float* __restrict__ qkv, float* __restrict__ part,
                                                                                 int P, int ldq) {
  __shared__ float red[RLA_SL_WAVES * KV_N];
  const int g = blockIdx.x, b = blockIdx.y, sl = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, h = lane >> 5;
  const float* base = qkv + static_cast<long long>(b) * P * ldq + g * 96;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ksum = 0.f;
  constexpr int UN = RLA_SL_PIX / (2 * RLA_SL_WAVES);  // 16 pixel pairs per wave, all 32 loads in flight together
  float kk[UN], vv[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int p = sl * RLA_SL_PIX + 2 * (wave + u * RLA_SL_WAVES) + h;
    kk[u] = 0.f;
    vv[u] = 0.f;
    if (p < P) {
      const float* row = base + static_cast<long long>(p) * ldq;
      kk[u] = fmaxf(row[32 + n], 0.f);
      vv[u] = row[64 + n];
    }
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    ksum += kk[u];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[u], kk[u], acc, 0, 0, 0);
  }
  ksum += __shfl_xor(ksum, 32, 64);
  {
    float* mine = red + wave * KV_N;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r >> 2) * 8 + h * 4 + (r & 3)) * 32 + n] = acc[r];  // KV[c][j = n]
    if (h == 0) mine[32 * 32 + n] = ksum;
  }
  __syncthreads();
  float* out = part + ((static_cast<long long>(b) * gridDim.x + g) * gridDim.z + sl) * KV_N;
  for (int e = tid; e < KV_N; e += RLA_SL_WAVES * 64) {
    float s_ = 0.f;
#pragma unroll
    for (int w = 0; w < RLA_SL_WAVES; ++w) s_ += red[w * KV_N + e];
    out[e] = s_;
  }
}

__global__ __launch_bounds__(RLA_SL_WAVES * 64) void relu_linear_attn_apply_slice_kernel(const float* __restrict__ qkv, const float* __restrict__ part,
                                                                                    float* __restrict__ y, int P, int ldq, int ldy, float eps,
                                                                                    int out_fmt) {
  __shared__ float kv[KV_N];
  const int g = blockIdx.x, b = blockIdx.y, sl = blockIdx.z, slices = gridDim.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, h = lane >> 5;
  const float* base = qkv + static_cast<long long>(b) * P * ldq + g * 96;
  // this wave's 32 pixels: q[p][16 h .. 16 h + 15] (loads issued before the partials are summed)
  const int p = sl * RLA_SL_PIX + 32 * wave + n;
  float4 qn[4];
  {
    const float* qp = base + static_cast<long long>(p < P ? p : P - 1) * ldq + 16 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i) qn[i] = *reinterpret_cast<const float4*>(qp + 4 * i);
  }
  const float* pg = part + (static_cast<long long>(b) * gridDim.x + g) * slices * KV_N;
  {
    // KV = the partials in slice order.  All loads of 16 slices x this thread's 5 elements are in flight together (a dependent
    // load-add chain per element made this the longest part of the launch: 24 us at 15 slices)
    constexpr int NE = (KV_N + RLA_SL_WAVES * 64 - 1) / (RLA_SL_WAVES * 64);  // 5
    float acc_[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) acc_[j] = 0.f;
    for (int i0 = 0; i0 < slices; i0 += 16) {
      float v_[NE][16];
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int e = tid + j * RLA_SL_WAVES * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          v_[j][i] = (e < KV_N && i0 + i < slices) ? pg[static_cast<long long>(i0 + i) * KV_N + e] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NE; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_[j] += v_[j][i];  // fixed order; the padding zeros do not change a sum
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + j * RLA_SL_WAVES * 64;
      if (e < KV_N) kv[e] = acc_[j];
    }
  }
  __syncthreads();
  if (sl * RLA_SL_PIX + 32 * wave >= P) return;  // (after the only barrier)
  float ka[16], kd[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    ka[kk] = kv[n * 32 + 16 * h + kk];
    kd[kk] = kv[32 * 32 + 16 * h + kk];
  }
  float q[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q[4 * i] = fmaxf(qn[i].x, 0.f);
    q[4 * i + 1] = fmaxf(qn[i].y, 0.f);
    q[4 * i + 2] = fmaxf(qn[i].z, 0.f);
    q[4 * i + 3] = fmaxf(qn[i].w, 0.f);
  }
  float den = 0.f;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) den = fmaf(kd[kk], q[kk], den);
  den += __shfl_xor(den, 32, 64);
  const float inv = 1.0f / (den + eps);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk], q[kk], o, 0, 0, 0);  // out^T[c][p = n]
  if (p < P) {
    float* yrow = y + (static_cast<long long>(b) * P + p) * ldy;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int ch = g * 32 + 8 * r4 + 4 * h;
      const float o0 = o[4 * r4] * inv, o1 = o[4 * r4 + 1] * inv, o2 = o[4 * r4 + 2] * inv, o3 = o[4 * r4 + 3] * inv;
      if (out_fmt != LDC_FMT_F32) ldc_store_fmt4(reinterpret_cast<unsigned char*>(yrow), ch, out_fmt, o0, o1, o2, o3);
      else *reinterpret_cast<float4*>(yrow + ch) = make_float4(o0, o1, o2, o3);
    }
  }
}

// RMSNorm over channels of an NHWC row with weight + bias, then optional residual add and activation:
//   y = act(x * rsqrt(mean(x^2) + eps) * w + b (+ resid))          (models/DCAE.py:259-260,317-322,371-377,729-730)
// NV4 float4 per lane hold the row: 4 (C <= 1024: the 84-variable autoencoder's widths) or 8 (C <= 2048: configs/DC_AE_ray_1024.yaml)
template <int NV4>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, const float* __restrict__ resid,
                                                           float* __restrict__ y, float* __restrict__ ys, long long rows, int C,
                                                           int ldx, int ldr, int ldy, int lds, int fmt, float eps, int act) {
  const int lane = threadIdx.x & 63;
  const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const int nv4 = C >> 2;
  float4 v[NV4];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      v[i] = reinterpret_cast<const float4*>(xr)[c];
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float r = rsqrtf(wave_sum(ss) / static_cast<float>(C) + eps);
  const float* rr = resid ? resid + row * ldr : nullptr;
  float* yr = y ? y + row * ldy : nullptr;  // fp32 copy: residual stream / input of a depthwise conv
  unsigned char* sr = ys ? reinterpret_cast<unsigned char*>(ys + row * lds) : nullptr;  // split copy: operand rows of the next conv
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      const float4 wv = reinterpret_cast<const float4*>(w)[c];
      const float4 bv = b ? reinterpret_cast<const float4*>(b)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o = make_float4(v[i].x * r * wv.x + bv.x, v[i].y * r * wv.y + bv.y, v[i].z * r * wv.z + bv.z, v[i].w * r * wv.w + bv.w);
      if (rr) {
        const float4 q = reinterpret_cast<const float4*>(rr)[c];
        o.x += q.x;
        o.y += q.y;
        o.z += q.z;
        o.w += q.w;
      }
      o.x = ldc_apply_act(o.x, act);
      o.y = ldc_apply_act(o.y, act);
      o.z = ldc_apply_act(o.z, act);
      o.w = ldc_apply_act(o.w, act);
      if (yr) reinterpret_cast<float4*>(yr)[c] = o;
      if (sr) {
        ldc_store_fmt4(sr, 4 * c, fmt, o.x, o.y, o.z, o.w);
        if ((C & 4) && c == nv4 - 1) ldc_zero_fmt4(sr, 4 * c + 4, fmt);
      }
    }
  }
}

// fp32 rows -> split rows (the conv_in outputs, whose epilogue writes the fp32 stream)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, float* __restrict__ ys, long long rows, int C,
                                                         int ldx, int lds, int fmt) {
  const int nv4 = C >> 2;
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= rows * nv4) return;
  const long long row = idx / nv4;
  const int c = static_cast<int>(idx - row * nv4);
  const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + 4 * c);
  unsigned char* sr = reinterpret_cast<unsigned char*>(ys + row * lds);
  ldc_store_fmt4(sr, 4 * c, fmt, v.x, v.y, v.z, v.w);
  if ((C & 4) && c == nv4 - 1) ldc_zero_fmt4(sr, 4 * c + 4, fmt);
}

// DCDownBlock2d tail: y[b,h2,w2,co] = cv[b,2h2+i,2w2+j,c] (co = 4c+2i+j)  +  mean_g xs[co*G+g],
// xs = pixel_unshuffle(x): channel q = 4cx+2i'+j' <- x[b,2h2+i',2w2+j',cx]     (models/DCAE.py:477-490)
// One thread = 4 consecutive output channels (one c, the four (i, j)); fp32 rows and / or split rows (ys) for the next conv.
__global__ __launch_bounds__(256) void pixel_unshuffle_shortcut_kernel(const float* __restrict__ cv,
                                                                       const float* __restrict__ x, float* __restrict__ y,
                                                                       float* __restrict__ ys, int H2, int W2, int cout, int cin,
                                                                       int G, int lds, int fmt, long long total4) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int cq = cout >> 2;
  const int c = static_cast<int>(idx % cq);
  const long long pix = idx / cq;  // b*H2*W2 + h2*W2 + w2
  const int w2 = static_cast<int>(pix % W2);
  const long long bh = pix / W2;
  const int h2 = static_cast<int>(bh % H2);
  const long long b = bh / H2;
  const int Wf = 2 * W2, Hf = 2 * H2;
  const long long pbase = (b * Hf + 2 * h2) * Wf + 2 * w2;  // top-left full-res pixel
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int co = 4 * c + e, i = e >> 1, j = e & 1;
    const float v = cv[(pbase + static_cast<long long>(i) * Wf + j) * cq + c];
    if (x == nullptr) {  // ABI 5: a block without the shortcut (the encoder's conv_in when layers_per_block[0] == 0, models/DCAE.py:571-579)
      o[e] = v;
      continue;
    }
    float s = 0.f;
    for (int g = 0; g < G; ++g) {
      const int q = co * G + g;
      const int cx = q >> 2, ii = (q >> 1) & 1, jj = q & 1;
      s += x[(pbase + static_cast<long long>(ii) * Wf + jj) * cin + cx];
    }
    o[e] = v + s / static_cast<float>(G);
  }
  if (y) *reinterpret_cast<float4*>(y + pix * cout + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
  if (ys) {
    unsigned char* sr = reinterpret_cast<unsigned char*>(ys + pix * lds);
    ldc_store_fmt4(sr, 4 * c, fmt, o[0], o[1], o[2], o[3]);
    if ((cout & 4) && c == cq - 1) ldc_zero_fmt4(sr, 4 * c + 4, fmt);
  }
}

// DCUpBlock2d tail: y[b,2h+i,2w+j,c] = cv[b,h,w,4c+2i+j] + x[b,h,w,(4c+2i+j)/rep]   (models/DCAE.py:526-532)
__global__ __launch_bounds__(256) void pixel_shuffle_shortcut_kernel(const float* __restrict__ cv, const float* __restrict__ x,
                                                                     float* __restrict__ y, float* __restrict__ ys, int H, int W,
                                                                     int cout, int cin, int rep, int lds, int fmt, long long total4) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int cq = cout >> 2;
  const int c4 = static_cast<int>(idx % cq);
  const long long pix = idx / cq;  // b*(2H)*(2W) + hf*(2W) + wf
  const int wf = static_cast<int>(pix % (2 * W));
  const long long bh = pix / (2 * W);
  const int hf = static_cast<int>(bh % (2 * H));
  const long long b = bh / (2 * H);
  const long long src = (b * H + (hf >> 1)) * W + (wf >> 1);
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int q = 4 * (4 * c4 + e) + 2 * (hf & 1) + (wf & 1);
    o[e] = (cv ? cv[src * (4LL * cout) + q] : 0.f) + x[src * cin + q / rep];  // cv == NULL: the shortcut alone (interpolate up block)
  }
  if (y) *reinterpret_cast<float4*>(y + pix * cout + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
  if (ys) {
    unsigned char* sr = reinterpret_cast<unsigned char*>(ys + pix * lds);
    ldc_store_fmt4(sr, 4 * c4, fmt, o[0], o[1], o[2], o[3]);
    if ((cout & 4) && c4 == cq - 1) ldc_zero_fmt4(sr, 4 * c4 + 4, fmt);
  }
}

// nearest-neighbour x2 up-sampling of NHWC rows: y[b, 2h + i, 2w + j, :] = x[b, h, w, :] (F.interpolate(scale_factor=2, mode="nearest"),
// DCUpBlock2d with interpolate=True: models/DCAE.py:519-525) - fp32 rows and / or operand rows for the conv that follows
__global__ __launch_bounds__(256) void upsample_nearest2x_rows_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ ys, int H, int W,
                                                                      int C, int ldx, int ldy, int lds, int fmt, long long total4) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int cq = C >> 2;
  const int c4 = static_cast<int>(idx % cq);
  const long long pix = idx / cq;  // b*(2H)*(2W) + hf*(2W) + wf
  const int wf = static_cast<int>(pix % (2 * W));
  const long long bh = pix / (2 * W);
  const int hf = static_cast<int>(bh % (2 * H));
  const long long b = bh / (2 * H);
  const long long src = (b * H + (hf >> 1)) * W + (wf >> 1);
  const float4 v = *reinterpret_cast<const float4*>(x + src * ldx + 4 * c4);
  if (y) *reinterpret_cast<float4*>(y + pix * ldy + 4 * c4) = v;
  if (ys) {
    unsigned char* sr = reinterpret_cast<unsigned char*>(ys + pix * lds);
    ldc_store_fmt4(sr, 4 * c4, fmt, v.x, v.y, v.z, v.w);
    if ((C & 4) && c4 == cq - 1) ldc_zero_fmt4(sr, 4 * c4 + 4, fmt);
  }
}

// channel regroup: cin > cout: y[m][c] = mean_g x[m][c*G+g] (G = cin/cout); else y[m][q] = x[m][q / (cout/cin)]
__global__ __launch_bounds__(256) void chan_regroup_kernel(const float* __restrict__ x, float* __restrict__ y, int cin,
                                                           int cout, long long total) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = static_cast<int>(idx % cout);
  const long long m = idx / cout;
  if (cin >= cout) {
    const int G = cin / cout;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += x[m * cin + c * G + g];
    y[idx] = s / static_cast<float>(G);
  } else {
    y[idx] = x[m * cin + c / (cout / cin)];
  }
}

}  // namespace

extern "C" int ldc_sphere_dwconv_nhwc(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                                      int C, int ldx, int ldy, int ksize, int glu, void* stream) {
  return ldc_sphere_dwconv_nhwc_fmt(x, wt, bias, y, B, H, W, C, ldx, ldy, ksize, glu, LDC_FMT_F32, stream);
}

extern "C" int ldc_sphere_dwconv_nhwc_fmt(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                                          int C, int ldx, int ldy, int ksize, int glu, int out_fmt, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(wt);
  LDC_CHECK_PTR(y);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return LDC_ERR_ARG;
  if ((ksize != 3 && ksize != 5) || (W & 1) || H < 2 || H < ksize / 2) return LDC_ERR_UNSUPPORTED;
  if ((C & 3) || (ldx & 3) || (ldy & 3) || (glu && (C & 7))) return LDC_ERR_ALIGN;
  if (out_fmt != LDC_FMT_F32 && out_fmt != LDC_FMT_SPLIT && out_fmt != LDC_FMT_BF16) return LDC_ERR_UNSUPPORTED;
  if (out_fmt != LDC_FMT_F32 && (((glu ? C / 2 : C) & 7) || (ldy & 7) || (reinterpret_cast<uintptr_t>(y) & 31u))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(wt);
  LDC_CHECK_ALIGN16(y);
  const long long pix = static_cast<long long>(B) * H * W;
  if (pix > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (W >= 8 + ksize - 1) {  // register-tiled kernel: 8 pixels of an image row per thread
    const long long total = static_cast<long long>(B) * H * ldc_cdiv(W, 8) * ((glu ? C / 2 : C) / 4);
    const dim3 tg(static_cast<unsigned>((total + 255) / 256));
    if (ksize == 3 && glu) hipLaunchKernelGGL((sphere_dwconv_row_kernel<3, true>), tg, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt, total);
    else if (ksize == 3) hipLaunchKernelGGL((sphere_dwconv_row_kernel<3, false>), tg, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt, total);
    else if (glu) hipLaunchKernelGGL((sphere_dwconv_row_kernel<5, true>), tg, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt, total);
    else hipLaunchKernelGGL((sphere_dwconv_row_kernel<5, false>), tg, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt, total);
    return ldc_launch_status();
  }
  const int nvec = (glu ? C / 2 : C) / 4;
  dim3 grid(ldc_cdiv(nvec, 256), static_cast<unsigned>(pix));
  if (ksize == 3 && glu) hipLaunchKernelGGL((sphere_dwconv_kernel<3, true>), grid, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt);
  else if (ksize == 3) hipLaunchKernelGGL((sphere_dwconv_kernel<3, false>), grid, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt);
  else if (glu) hipLaunchKernelGGL((sphere_dwconv_kernel<5, true>), grid, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt);
  else hipLaunchKernelGGL((sphere_dwconv_kernel<5, false>), grid, dim3(256), 0, s, x, wt, bias, y, H, W, C, ldx, ldy, out_fmt);
  return ldc_launch_status();
}

extern "C" int ldc_grouped_conv1x1_nhwc(const float* x, const float* wt, float* y, long long M, int groups, int ldx,
                                        int ldy, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(wt);
  LDC_CHECK_PTR(y);
  if (M <= 0 || groups <= 0 || groups > 65535) return LDC_ERR_ARG;
  LDC_CHECK_ALIGN16(wt);
  dim3 grid(ldc_cdiv(M, 64), groups);
  hipLaunchKernelGGL(grouped_conv1x1_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, wt, y,
                     static_cast<int>(M), ldx, ldy);
  return ldc_launch_status();
}

// scratch of the sliced form (B x groups x ceil(P / 128) partial 33 x 32 matrices); a call with less (or NULL) runs the one-launch
// kernel, whose sums are taken in another order (same result up to fp32 rounding)
extern "C" long long ldc_relu_linear_attn_workspace_bytes(int B, int P, int groups) {
  if (B <= 0 || P < RLA_SLICED_MIN_P || groups <= 0) return 0;  // below: the one-launch kernel, no scratch
  return static_cast<long long>(B) * groups * ldc_cdiv(P, RLA_SL_PIX) * KV_N * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_relu_linear_attn_nhwc(const float* qkv, float* y, int B, int P, int groups, int ldq, int ldy, float eps,
                                         void* workspace, long long workspace_bytes, void* stream) {
  return ldc_relu_linear_attn_nhwc_fmt(qkv, y, B, P, groups, ldq, ldy, eps, LDC_FMT_F32, workspace, workspace_bytes, stream);
}

extern "C" int ldc_relu_linear_attn_nhwc_fmt(const float* qkv, float* y, int B, int P, int groups, int ldq, int ldy, float eps,
                                             int out_fmt, void* workspace, long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(qkv);
  LDC_CHECK_PTR(y);
  if (B <= 0 || P <= 0 || groups <= 0) return LDC_ERR_ARG;
  if ((ldq & 3) || (ldy & 3)) return LDC_ERR_ALIGN;
  if (B > 65535) return LDC_ERR_UNSUPPORTED;
  if (out_fmt != LDC_FMT_F32 && out_fmt != LDC_FMT_SPLIT && out_fmt != LDC_FMT_BF16) return LDC_ERR_UNSUPPORTED;
  if (out_fmt != LDC_FMT_F32 && ((ldy & 7) || (reinterpret_cast<uintptr_t>(y) & 31u))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(qkv);
  LDC_CHECK_ALIGN16(y);
  const int slices = ldc_cdiv(P, RLA_SL_PIX);
  // sliced only where a group has enough pixels to pay for the second launch (measured at one frame, back to back: 1800 pixels 40.3 ->
  // 20.5 us, 450 pixels 10.8 -> 13.7 us); the rule looks at P alone, so a frame's result does not depend on its batch
  if (P >= RLA_SLICED_MIN_P && workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0 && slices <= 65535 &&
      workspace_bytes >= ldc_relu_linear_attn_workspace_bytes(B, P, groups)) {
    float* part = static_cast<float*>(workspace);
    const dim3 grid(groups, B, slices);
    hipLaunchKernelGGL(relu_linear_attn_kv_slice_kernel, grid, dim3(RLA_SL_WAVES * 64), 0, static_cast<hipStream_t>(stream), qkv, part, P, ldq);
    hipLaunchKernelGGL(relu_linear_attn_apply_slice_kernel, grid, dim3(RLA_SL_WAVES * 64), 0, static_cast<hipStream_t>(stream), qkv, part, y, P,
                       ldq, ldy, eps, out_fmt);
    return ldc_launch_status();
  }
  const size_t lds = (RLA_WAVES + 1) * KV_N * sizeof(float);  // 71.8 KB
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(relu_linear_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL(relu_linear_attn_kernel, dim3(groups, B), dim3(RLA_WAVES * 64), lds, static_cast<hipStream_t>(stream), qkv, y, P,
                     ldq, ldy, eps, out_fmt);
  return ldc_launch_status();
}

extern "C" int ldc_rmsnorm_rows(const float* x, const float* w, const float* b, const float* resid, float* y,
                                long long rows, int C, int ldx, int ldr, int ldy, float eps, int act, void* stream) {
  LDC_CHECK_PTR(y);
  return ldc_rmsnorm_rows_split(x, w, b, resid, y, nullptr, rows, C, ldx, ldr, ldy, 0, LDC_FMT_SPLIT, eps, act, stream);
}

// y (fp32 rows) and / or ys (operand rows in format fmt = LDC_FMT_SPLIT | LDC_FMT_BF16, lds >= C rounded up to 8, pad columns zeroed)
extern "C" int ldc_rmsnorm_rows_split(const float* x, const float* w, const float* b, const float* resid, float* y, float* ys,
                                      long long rows, int C, int ldx, int ldr, int ldy, int lds, int fmt, float eps, int act,
                                      void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(w);
  if (y == nullptr && ys == nullptr) return LDC_ERR_ARG;
  if (ys && fmt != LDC_FMT_SPLIT && fmt != LDC_FMT_BF16) return LDC_ERR_UNSUPPORTED;
  if (rows <= 0 || C <= 0) return LDC_ERR_ARG;
  if ((C & 3) || C > 2048 || (ldx & 3) || (y && (ldy & 3)) || (resid && (ldr & 3))) return LDC_ERR_UNSUPPORTED;
  if (ys && ((lds & 7) || lds < ((C + 7) & ~7) || (reinterpret_cast<uintptr_t>(ys) & 31u))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(w);
  if (y) LDC_CHECK_ALIGN16(y);
  if (C <= 1024)
    hipLaunchKernelGGL(rmsnorm_rows_kernel<4>, dim3(ldc_cdiv(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, b,
                       resid, y, ys, rows, C, ldx, ldr, ldy, lds, fmt, eps, act);
  else
    hipLaunchKernelGGL(rmsnorm_rows_kernel<8>, dim3(ldc_cdiv(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, b,
                       resid, y, ys, rows, C, ldx, ldr, ldy, lds, fmt, eps, act);
  return ldc_launch_status();
}

extern "C" int ldc_split_rows(const float* x, float* ys, long long rows, int C, int ldx, int lds, int fmt, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(ys);
  if (rows <= 0 || C <= 0) return LDC_ERR_ARG;
  if (fmt != LDC_FMT_SPLIT && fmt != LDC_FMT_BF16) return LDC_ERR_UNSUPPORTED;
  if ((C & 3) || (ldx & 3) || (lds & 7) || lds < ((C + 7) & ~7) || (reinterpret_cast<uintptr_t>(ys) & 31u)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  hipLaunchKernelGGL(split_rows_kernel, dim3(ldc_cdiv(rows * (C >> 2), 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ys,
                     rows, C, ldx, lds, fmt);
  return ldc_launch_status();
}

extern "C" int ldc_pixel_unshuffle_shortcut(const float* cv, const float* x, float* y, int B, int H2, int W2, int cout,
                                            int cin, void* stream) {
  LDC_CHECK_PTR(y);
  return ldc_pixel_unshuffle_shortcut_split(cv, x, y, nullptr, B, H2, W2, cout, cin, 0, LDC_FMT_SPLIT, stream);
}

static int split_out_ok(const float* y, const float* ys, int cout, int lds, int fmt) {
  if (y == nullptr && ys == nullptr) return LDC_ERR_ARG;
  if (ys && fmt != LDC_FMT_SPLIT && fmt != LDC_FMT_BF16) return LDC_ERR_UNSUPPORTED;
  if (y && (reinterpret_cast<uintptr_t>(y) & 15u)) return LDC_ERR_ALIGN;
  if (ys && ((lds & 7) || lds < ((cout + 7) & ~7) || (reinterpret_cast<uintptr_t>(ys) & 31u))) return LDC_ERR_ALIGN;
  return LDC_OK;
}

extern "C" int ldc_pixel_unshuffle_shortcut_split(const float* cv, const float* x, float* y, float* ys, int B, int H2, int W2,
                                                  int cout, int cin, int lds, int fmt, void* stream) {
  LDC_CHECK_PTR(cv);  // x may be NULL (ABI 5): y = pixel_unshuffle(cv) alone, no shortcut term
  if (B <= 0 || H2 <= 0 || W2 <= 0 || cout <= 0 || cin <= 0) return LDC_ERR_ARG;
  if ((cout & 3) || (x != nullptr && ((4 * cin) % cout || 4 * cin < cout))) return LDC_ERR_UNSUPPORTED;
  const int ok = split_out_ok(y, ys, cout, lds, fmt);
  if (ok != LDC_OK) return ok;
  const long long total4 = static_cast<long long>(B) * H2 * W2 * (cout >> 2);
  hipLaunchKernelGGL(pixel_unshuffle_shortcut_kernel, dim3(ldc_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), cv, x, y, ys, H2, W2, cout, cin, 4 * cin / cout, lds, fmt, total4);
  return ldc_launch_status();
}

extern "C" int ldc_pixel_shuffle_shortcut(const float* cv, const float* x, float* y, int B, int H, int W, int cout,
                                          int cin, void* stream) {
  LDC_CHECK_PTR(y);
  return ldc_pixel_shuffle_shortcut_split(cv, x, y, nullptr, B, H, W, cout, cin, 0, LDC_FMT_SPLIT, stream);
}

extern "C" int ldc_pixel_shuffle_shortcut_split(const float* cv, const float* x, float* y, float* ys, int B, int H, int W, int cout,
                                                int cin, int lds, int fmt, void* stream) {
  LDC_CHECK_PTR(x);  // cv may be NULL (ABI 4): y = the shortcut term alone
  if (B <= 0 || H <= 0 || W <= 0 || cout <= 0 || cin <= 0) return LDC_ERR_ARG;
  if ((cout & 3) || (4 * cout) % cin) return LDC_ERR_UNSUPPORTED;
  const int ok = split_out_ok(y, ys, cout, lds, fmt);
  if (ok != LDC_OK) return ok;
  const long long total4 = static_cast<long long>(B) * 4 * H * W * (cout >> 2);
  hipLaunchKernelGGL(pixel_shuffle_shortcut_kernel, dim3(ldc_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), cv, x, y, ys, H, W, cout, cin, 4 * cout / cin, lds, fmt, total4);
  return ldc_launch_status();
}

// DCUpBlock2d WITHOUT shortcut as the decoder's last layer (conv_out when layers_per_block[0] == 0, models/DCAE.py:706-712,526-536 with shortcut=False):
// pixel_shuffle of the conv's NHWC rows written straight as the NCHW result, first `keep` of the cout channels (the static channels behind them are
// dropped there, :1050-1052): out[b][c][2h + i][2w + j] = cv[b][h][w][4 c + 2 i + j].  Any cout (the field count need not be a multiple of 4).
__global__ __launch_bounds__(256) void pixel_shuffle_to_chan_kernel(const float* __restrict__ cv, float* __restrict__ out, int H, int W, int cout, int keep,
                                                                     long long total) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int wf = static_cast<int>(idx % (2 * W));
  const long long r1 = idx / (2 * W);
  const int hf = static_cast<int>(r1 % (2 * H));
  const long long r2 = r1 / (2 * H);
  const int c = static_cast<int>(r2 % keep);
  const long long b = r2 / keep;
  const long long src = (b * H + (hf >> 1)) * W + (wf >> 1);
  out[idx] = cv[src * (4LL * cout) + 4 * c + 2 * (hf & 1) + (wf & 1)];
}

extern "C" int ldc_pixel_shuffle_to_chan(const float* cv, float* out, int B, int H, int W, int cout, int keep, void* stream) {
  LDC_CHECK_PTR(cv);
  LDC_CHECK_PTR(out);
  if (B <= 0 || H <= 0 || W <= 0 || cout <= 0 || keep <= 0 || keep > cout) return LDC_ERR_ARG;
  const long long total = static_cast<long long>(B) * keep * 4 * H * W;
  hipLaunchKernelGGL(pixel_shuffle_to_chan_kernel, dim3(ldc_cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), cv, out, H, W, cout, keep,
                     total);
  return ldc_launch_status();
}

extern "C" int ldc_upsample_nearest2x_rows(const float* x, float* y, float* ys, int B, int H, int W, int C, int ldx, int ldy, int lds, int fmt,
                                           void* stream) {
  LDC_CHECK_PTR(x);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return LDC_ERR_ARG;
  if ((C & 3) || (ldx & 3) || ldx < C || (y && ((ldy & 3) || ldy < C))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  if (y) LDC_CHECK_ALIGN16(y);
  const int ok = split_out_ok(y, ys, C, lds, fmt);
  if (ok != LDC_OK) return ok;
  const long long total4 = static_cast<long long>(B) * 4 * H * W * (C >> 2);
  hipLaunchKernelGGL(upsample_nearest2x_rows_kernel, dim3(ldc_cdiv(total4, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, ys, H, W, C, ldx,
                     ldy, lds, fmt, total4);
  return ldc_launch_status();
}

extern "C" int ldc_chan_regroup(const float* x, float* y, long long M, int cin, int cout, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  if (M <= 0 || cin <= 0 || cout <= 0) return LDC_ERR_ARG;
  if ((cin >= cout) ? (cin % cout) : (cout % cin)) return LDC_ERR_UNSUPPORTED;
  const long long total = M * cout;
  hipLaunchKernelGGL(chan_regroup_kernel, dim3(ldc_cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
                     cin, cout, total);
  return ldc_launch_status();
}
