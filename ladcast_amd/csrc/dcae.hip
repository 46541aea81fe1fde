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
  if (out_fmt == LDC_FMT_SPLIT)  // operand rows of the next pointwise conv (gemm_bf16x3_v3.hip)
    ldc_store_split4(reinterpret_cast<unsigned char*>(y + static_cast<long long>(pix) * ldy), c0, a0.x, a0.y, a0.z, a0.w);
  else
    *reinterpret_cast<float4*>(y + static_cast<long long>(pix) * ldy + c0) = a0;
}

// grouped 1x1 conv, 32 -> 32 channels per group: y[m][g*32+o] = sum_i W[g*32+o][i] * x[m][g*32+i]
// block = 8 pixels x one group; the group's 32x32 weights and the 8x32 inputs sit in LDS.
__global__ __launch_bounds__(256) void grouped_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                              float* __restrict__ y, int M, int ldx, int ldy) {
  __shared__ float ws[32][33];
  __shared__ float xs[8][32];
  const int g = blockIdx.y;
  const int m0 = blockIdx.x * 8;
  const int t = threadIdx.x;
  for (int i = t; i < 1024; i += 256) ws[i >> 5][i & 31] = wt[static_cast<long long>(g * 32 + (i >> 5)) * 32 + (i & 31)];
  {
    const int pm = t >> 5, ci = t & 31;
    xs[pm][ci] = (m0 + pm < M) ? x[static_cast<long long>(m0 + pm) * ldx + g * 32 + ci] : 0.f;
  }
  __syncthreads();
  const int pm = t >> 5, o = t & 31;
  if (m0 + pm >= M) return;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) acc = fmaf(ws[o][i], xs[pm][i], acc);
  y[static_cast<long long>(m0 + pm) * ldy + g * 32 + o] = acc;
}

// ReLU linear attention of one (batch, 96-channel group): q = ch 0..31, k = 32..63, v = 64..95
//   KV[c][j] = sum_p vpad[p][c] relu(k[p][j]), c = 0..32 (row 32 = ones);  out[p][c] = sum_j KV[c][j] relu(q[p][j])
//   y[p][c] = out[p][c] / (out[p][32] + eps)                 (models/DCAE.py:158-175,239-249; fp32)
// Two launches so that one frame (63 groups) still fills the chip: (1) partial KV over a slice of the pixels per
// workgroup, (2) per 64 pixels: add the slices' partials in order (deterministic) and apply.
constexpr int KV_N = 33 * 32;

__global__ __launch_bounds__(256) void relu_linear_attn_kv_kernel(const float* __restrict__ qkv, float* __restrict__ part, int P,
                                                                  int ldq, int pix_per_wg) {
  __shared__ float tile[64][97];  // 64 pixels x 96 channels (+1 pad)
  const int sl = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x;
  const float* base = qkv + static_cast<long long>(b) * P * ldq + g * 96;
  const int p_begin = sl * pix_per_wg;
  const int p_end = min(P, p_begin + pix_per_wg);
  // each thread owns up to 5 (c, j) entries of KV: e = t + 256*i < 1056
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    for (int i = t; i < 64 * 96; i += 256) {
      const int pp = i / 96, ch = i - pp * 96;
      float v = (p0 + pp < p_end) ? base[static_cast<long long>(p0 + pp) * ldq + ch] : 0.f;
      if (ch >= 32 && ch < 64) v = fmaxf(v, 0.f);  // relu(k)
      tile[pp][ch] = v;
    }
    __syncthreads();
    const int np = min(64, p_end - p0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = t + 256 * i;
      if (e < KV_N) {
        const int c = e >> 5, j = e & 31;
        float s = acc[i];
        if (c < 32) {
          for (int pp = 0; pp < np; ++pp) s = fmaf(tile[pp][64 + c], tile[pp][32 + j], s);
        } else {
          for (int pp = 0; pp < np; ++pp) s += tile[pp][32 + j];
        }
        acc[i] = s;
      }
    }
    __syncthreads();
  }
  float* dst = part + ((static_cast<long long>(b) * gridDim.y + g) * gridDim.x + sl) * KV_N;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int e = t + 256 * i;
    if (e < KV_N) dst[e] = acc[i];
  }
}

__global__ __launch_bounds__(256) void relu_linear_attn_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ part,
                                                                     float* __restrict__ y, int P, int ldq, int ldy, float eps,
                                                                     int nslice, int out_fmt) {
  __shared__ float kv[33][32];
  const int g = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x;
  const float* src = part + (static_cast<long long>(b) * gridDim.y + g) * nslice * KV_N;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int e = t + 256 * i;
    if (e < KV_N) {
      float s = 0.f;
      for (int sl = 0; sl < nslice; ++sl) s += src[static_cast<long long>(sl) * KV_N + e];
      kv[e >> 5][e & 31] = s;
    }
  }
  __syncthreads();
  // 64 pixels per workgroup, 4 threads per pixel: thread (pixel, quarter) writes channels [8 quarter, 8 quarter + 8)
  const int p = blockIdx.x * 64 + (t >> 2), qt = t & 3;
  if (p >= P) return;
  const float* qp = qkv + static_cast<long long>(b) * P * ldq + g * 96 + static_cast<long long>(p) * ldq;
  float q[32];
#pragma unroll
  for (int j = 0; j < 32; j += 4) {
    const float4 v = *reinterpret_cast<const float4*>(qp + j);
    q[j] = fmaxf(v.x, 0.f);
    q[j + 1] = fmaxf(v.y, 0.f);
    q[j + 2] = fmaxf(v.z, 0.f);
    q[j + 3] = fmaxf(v.w, 0.f);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) den = fmaf(kv[32][j], q[j], den);
  const float inv = 1.0f / (den + eps);
  float* yrow = y + static_cast<long long>(b) * P * ldy + static_cast<long long>(p) * ldy;
  const int ch = g * 32 + 8 * qt;  // this thread's 8 channels: one group of the split format
#pragma unroll
  for (int c = 0; c < 8; c += 4) {
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) s = fmaf(kv[8 * qt + c + u][j], q[j], s);
      o[u] = s * inv;
    }
    if (out_fmt == LDC_FMT_SPLIT) ldc_store_split4(reinterpret_cast<unsigned char*>(yrow), ch + c, o[0], o[1], o[2], o[3]);
    else *reinterpret_cast<float4*>(yrow + ch + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// RMSNorm over channels of an NHWC row with weight + bias, then optional residual add and activation:
//   y = act(x * rsqrt(mean(x^2) + eps) * w + b (+ resid))          (models/DCAE.py:259-260,317-322,371-377,729-730)
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, const float* __restrict__ resid,
                                                           float* __restrict__ y, float* __restrict__ ys, long long rows, int C,
                                                           int ldx, int ldr, int ldy, int lds, float eps, int act) {
  const int lane = threadIdx.x & 63;
  const long long row = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const int nv4 = C >> 2;
  float4 v[4];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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
  for (int i = 0; i < 4; ++i) {
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
        ldc_store_split4(sr, 4 * c, o.x, o.y, o.z, o.w);
        if ((C & 4) && c == nv4 - 1) ldc_zero_split4(sr, 4 * c + 4);
      }
    }
  }
}

// fp32 rows -> split rows (the conv_in outputs, whose epilogue writes the fp32 stream)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, float* __restrict__ ys, long long rows, int C,
                                                         int ldx, int lds) {
  const int nv4 = C >> 2;
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= rows * nv4) return;
  const long long row = idx / nv4;
  const int c = static_cast<int>(idx - row * nv4);
  const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + 4 * c);
  unsigned char* sr = reinterpret_cast<unsigned char*>(ys + row * lds);
  ldc_store_split4(sr, 4 * c, v.x, v.y, v.z, v.w);
  if ((C & 4) && c == nv4 - 1) ldc_zero_split4(sr, 4 * c + 4);
}

// DCDownBlock2d tail: y[b,h2,w2,co] = cv[b,2h2+i,2w2+j,c] (co = 4c+2i+j)  +  mean_g xs[co*G+g],
// xs = pixel_unshuffle(x): channel q = 4cx+2i'+j' <- x[b,2h2+i',2w2+j',cx]     (models/DCAE.py:477-490)
// One thread = 4 consecutive output channels (one c, the four (i, j)); fp32 rows and / or split rows (ys) for the next conv.
__global__ __launch_bounds__(256) void pixel_unshuffle_shortcut_kernel(const float* __restrict__ cv,
                                                                       const float* __restrict__ x, float* __restrict__ y,
                                                                       float* __restrict__ ys, int H2, int W2, int cout, int cin,
                                                                       int G, int lds, long long total4) {
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
    ldc_store_split4(sr, 4 * c, o[0], o[1], o[2], o[3]);
    if ((cout & 4) && c == cq - 1) ldc_zero_split4(sr, 4 * c + 4);
  }
}

// DCUpBlock2d tail: y[b,2h+i,2w+j,c] = cv[b,h,w,4c+2i+j] + x[b,h,w,(4c+2i+j)/rep]   (models/DCAE.py:526-532)
__global__ __launch_bounds__(256) void pixel_shuffle_shortcut_kernel(const float* __restrict__ cv, const float* __restrict__ x,
                                                                     float* __restrict__ y, float* __restrict__ ys, int H, int W,
                                                                     int cout, int cin, int rep, int lds, long long total4) {
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
    o[e] = cv[src * (4LL * cout) + q] + x[src * cin + q / rep];
  }
  if (y) *reinterpret_cast<float4*>(y + pix * cout + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
  if (ys) {
    unsigned char* sr = reinterpret_cast<unsigned char*>(ys + pix * lds);
    ldc_store_split4(sr, 4 * c4, o[0], o[1], o[2], o[3]);
    if ((cout & 4) && c4 == cq - 1) ldc_zero_split4(sr, 4 * c4 + 4);
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
  if (out_fmt != LDC_FMT_F32 && out_fmt != LDC_FMT_SPLIT) return LDC_ERR_UNSUPPORTED;
  if (out_fmt == LDC_FMT_SPLIT && (((glu ? C / 2 : C) & 7) || (ldy & 7) || (reinterpret_cast<uintptr_t>(y) & 31u))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(wt);
  LDC_CHECK_ALIGN16(y);
  const long long pix = static_cast<long long>(B) * H * W;
  if (pix > 0x7fffffffLL) return LDC_ERR_UNSUPPORTED;
  const int nvec = (glu ? C / 2 : C) / 4;
  dim3 grid(ldc_cdiv(nvec, 256), static_cast<unsigned>(pix));
  hipStream_t s = static_cast<hipStream_t>(stream);
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
  dim3 grid(ldc_cdiv(M, 8), groups);
  hipLaunchKernelGGL(grouped_conv1x1_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, wt, y,
                     static_cast<int>(M), ldx, ldy);
  return ldc_launch_status();
}

// pixel slices per (batch, group) of the KV pass: enough workgroups to fill the chip, at most one per 64 pixels
static int relu_attn_slices(int B, int P, int groups) {
  const int tiles = ldc_cdiv(P, 64);
  long long want = 2048 / (static_cast<long long>(B) * groups);
  if (want < 1) want = 1;
  return static_cast<int>(want < tiles ? want : tiles);
}

extern "C" long long ldc_relu_linear_attn_workspace_bytes(int B, int P, int groups) {
  if (B <= 0 || P <= 0 || groups <= 0) return 0;
  return static_cast<long long>(B) * groups * relu_attn_slices(B, P, groups) * KV_N * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_relu_linear_attn_nhwc(const float* qkv, float* y, int B, int P, int groups, int ldq, int ldy, float eps,
                                         void* workspace, long long workspace_bytes, void* stream) {
  return ldc_relu_linear_attn_nhwc_fmt(qkv, y, B, P, groups, ldq, ldy, eps, LDC_FMT_F32, workspace, workspace_bytes, stream);
}

extern "C" int ldc_relu_linear_attn_nhwc_fmt(const float* qkv, float* y, int B, int P, int groups, int ldq, int ldy, float eps,
                                             int out_fmt, void* workspace, long long workspace_bytes, void* stream) {
  LDC_CHECK_PTR(qkv);
  LDC_CHECK_PTR(y);
  LDC_CHECK_PTR(workspace);
  if (B <= 0 || P <= 0 || groups <= 0) return LDC_ERR_ARG;
  if ((ldq & 3) || (ldy & 3)) return LDC_ERR_ALIGN;
  if (groups > 65535 || B > 65535) return LDC_ERR_UNSUPPORTED;
  if (out_fmt != LDC_FMT_F32 && out_fmt != LDC_FMT_SPLIT) return LDC_ERR_UNSUPPORTED;
  if (out_fmt == LDC_FMT_SPLIT && ((ldy & 7) || (reinterpret_cast<uintptr_t>(y) & 31u))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(qkv);
  LDC_CHECK_ALIGN16(y);
  if (workspace_bytes < ldc_relu_linear_attn_workspace_bytes(B, P, groups)) return LDC_ERR_ARG;
  const int nslice = relu_attn_slices(B, P, groups);
  const int pix_per_wg = ldc_cdiv(ldc_cdiv(P, 64), nslice) * 64;
  const int nsl = ldc_cdiv(P, pix_per_wg);  // <= nslice; the apply pass adds exactly the slices that were written
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(relu_linear_attn_kv_kernel, dim3(nsl, groups, B), dim3(256), 0, s, qkv, part, P, ldq, pix_per_wg);
  int st = ldc_launch_status();
  if (st != LDC_OK) return st;
  hipLaunchKernelGGL(relu_linear_attn_apply_kernel, dim3(ldc_cdiv(P, 64), groups, B), dim3(256), 0, s, qkv, part, y, P, ldq, ldy, eps,
                     nsl, out_fmt);
  return ldc_launch_status();
}

extern "C" int ldc_rmsnorm_rows(const float* x, const float* w, const float* b, const float* resid, float* y,
                                long long rows, int C, int ldx, int ldr, int ldy, float eps, int act, void* stream) {
  LDC_CHECK_PTR(y);
  return ldc_rmsnorm_rows_split(x, w, b, resid, y, nullptr, rows, C, ldx, ldr, ldy, 0, eps, act, stream);
}

// y (fp32 rows) and / or ys (split rows, lds >= C rounded up to 8, pad columns zeroed): at least one
extern "C" int ldc_rmsnorm_rows_split(const float* x, const float* w, const float* b, const float* resid, float* y, float* ys,
                                      long long rows, int C, int ldx, int ldr, int ldy, int lds, float eps, int act, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(w);
  if (y == nullptr && ys == nullptr) return LDC_ERR_ARG;
  if (rows <= 0 || C <= 0) return LDC_ERR_ARG;
  if ((C & 3) || C > 1024 || (ldx & 3) || (y && (ldy & 3)) || (resid && (ldr & 3))) return LDC_ERR_UNSUPPORTED;
  if (ys && ((lds & 7) || lds < ((C + 7) & ~7) || (reinterpret_cast<uintptr_t>(ys) & 31u))) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(w);
  if (y) LDC_CHECK_ALIGN16(y);
  hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(ldc_cdiv(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, b,
                     resid, y, ys, rows, C, ldx, ldr, ldy, lds, eps, act);
  return ldc_launch_status();
}

extern "C" int ldc_split_rows(const float* x, float* ys, long long rows, int C, int ldx, int lds, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(ys);
  if (rows <= 0 || C <= 0) return LDC_ERR_ARG;
  if ((C & 3) || (ldx & 3) || (lds & 7) || lds < ((C + 7) & ~7) || (reinterpret_cast<uintptr_t>(ys) & 31u)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  hipLaunchKernelGGL(split_rows_kernel, dim3(ldc_cdiv(rows * (C >> 2), 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ys,
                     rows, C, ldx, lds);
  return ldc_launch_status();
}

extern "C" int ldc_pixel_unshuffle_shortcut(const float* cv, const float* x, float* y, int B, int H2, int W2, int cout,
                                            int cin, void* stream) {
  LDC_CHECK_PTR(y);
  return ldc_pixel_unshuffle_shortcut_split(cv, x, y, nullptr, B, H2, W2, cout, cin, 0, stream);
}

static int split_out_ok(const float* y, const float* ys, int cout, int lds) {
  if (y == nullptr && ys == nullptr) return LDC_ERR_ARG;
  if (y && (reinterpret_cast<uintptr_t>(y) & 15u)) return LDC_ERR_ALIGN;
  if (ys && ((lds & 7) || lds < ((cout + 7) & ~7) || (reinterpret_cast<uintptr_t>(ys) & 31u))) return LDC_ERR_ALIGN;
  return LDC_OK;
}

extern "C" int ldc_pixel_unshuffle_shortcut_split(const float* cv, const float* x, float* y, float* ys, int B, int H2, int W2,
                                                  int cout, int cin, int lds, void* stream) {
  LDC_CHECK_PTR(cv);
  LDC_CHECK_PTR(x);
  if (B <= 0 || H2 <= 0 || W2 <= 0 || cout <= 0 || cin <= 0) return LDC_ERR_ARG;
  if ((cout & 3) || (4 * cin) % cout) return LDC_ERR_UNSUPPORTED;
  const int ok = split_out_ok(y, ys, cout, lds);
  if (ok != LDC_OK) return ok;
  const long long total4 = static_cast<long long>(B) * H2 * W2 * (cout >> 2);
  hipLaunchKernelGGL(pixel_unshuffle_shortcut_kernel, dim3(ldc_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), cv, x, y, ys, H2, W2, cout, cin, 4 * cin / cout, lds, total4);
  return ldc_launch_status();
}

extern "C" int ldc_pixel_shuffle_shortcut(const float* cv, const float* x, float* y, int B, int H, int W, int cout,
                                          int cin, void* stream) {
  LDC_CHECK_PTR(y);
  return ldc_pixel_shuffle_shortcut_split(cv, x, y, nullptr, B, H, W, cout, cin, 0, stream);
}

extern "C" int ldc_pixel_shuffle_shortcut_split(const float* cv, const float* x, float* y, float* ys, int B, int H, int W, int cout,
                                                int cin, int lds, void* stream) {
  LDC_CHECK_PTR(cv);
  LDC_CHECK_PTR(x);
  if (B <= 0 || H <= 0 || W <= 0 || cout <= 0 || cin <= 0) return LDC_ERR_ARG;
  if ((cout & 3) || (4 * cout) % cin) return LDC_ERR_UNSUPPORTED;
  const int ok = split_out_ok(y, ys, cout, lds);
  if (ok != LDC_OK) return ok;
  const long long total4 = static_cast<long long>(B) * 4 * H * W * (cout >> 2);
  hipLaunchKernelGGL(pixel_shuffle_shortcut_kernel, dim3(ldc_cdiv(total4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), cv, x, y, ys, H, W, cout, cin, 4 * cout / cin, lds, total4);
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
