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

__global__ __launch_bounds__(256) void layernorm_mod_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int rows, int D, int ldx, long long x_bs, int ldy,
                                                            long long y_bs, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int mod_bs, int mode,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (row >= rows) return;
  const float* xr = x + b * x_bs + static_cast<long long>(row) * ldx;
  float* yr = y + b * y_bs + static_cast<long long>(row) * ldy;
  const int nv4 = D >> 2;
  float4 v[LN_MAX_V4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      v[i] = reinterpret_cast<const float4*>(xr)[c];
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
      reinterpret_cast<float4*>(yr)[c] = o;
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
// block = 64 columns x 4 row-groups; LDS combine of the 4 partial sums
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows,
                                                        int D, int ldx, long long x_bs) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  const int b = blockIdx.y;
  float s = 0.f;
  if (col < D) {
    const float* xb = x + b * x_bs + col;
    for (int r = g; r < rows; r += 4) s += xb[static_cast<long long>(r) * ldx];
  }
  part[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && col < D) {
    const int c = threadIdx.x & 63;
    y[static_cast<long long>(b) * D + col] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / static_cast<float>(rows);
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
  float4 o = make_float4(r.x + v.x * g.x, r.y + v.y * g.y, r.z + v.z * g.z, r.w + v.w * g.w);
  reinterpret_cast<float4*>(out + b * res_bs + static_cast<long long>(row) * ld_res)[c] = o;
}

// ---------------------------------------------------------------------------
// small-M linear: y[r][n] = act_out(sum_k W[n][k] act_in(x[r % x_rows][k]) + bias[n]) + add[r % add_rows][n]
// one wave per output column n, 8 rows per pass; W streamed once from HBM (float4/lane).
// ---------------------------------------------------------------------------
constexpr int LS_ROWS = 8;

__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, int x_rows,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ add, int add_rows,
                                                           float* __restrict__ y, int rows, int N, int K, int act_in,
                                                           int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int r_base = blockIdx.y * LS_ROWS;
  if (n >= N) return;
  const float* wr = W + static_cast<long long>(n) * K;
  float acc[LS_ROWS];
#pragma unroll
  for (int i = 0; i < LS_ROWS; ++i) acc[i] = 0.f;
  const int nv4 = K >> 2;
  for (int c = lane; c < nv4; c += 64) {
    const float4 w = reinterpret_cast<const float4*>(wr)[c];
#pragma unroll
    for (int i = 0; i < LS_ROWS; ++i) {
      const int r = r_base + i;
      if (r < rows) {
        float4 xv = reinterpret_cast<const float4*>(x + static_cast<long long>(r % x_rows) * K)[c];
        if (act_in != LDC_ACT_NONE) {
          xv.x = ldc_apply_act(xv.x, act_in);
          xv.y = ldc_apply_act(xv.y, act_in);
          xv.z = ldc_apply_act(xv.z, act_in);
          xv.w = ldc_apply_act(xv.w, act_in);
        }
        acc[i] += (w.x * xv.x + w.y * xv.y) + (w.z * xv.z + w.w * xv.w);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LS_ROWS; ++i) {
    const float s = wave_sum(acc[i]);
    const int r = r_base + i;
    if (lane == 0 && r < rows) {
      float v = s + (bias ? bias[n] : 0.f);
      v = ldc_apply_act(v, act_out);
      if (add) v += add[static_cast<long long>(r % add_rows) * N + n];
      y[static_cast<long long>(r) * N + n] = v;
    }
  }
}

}  // namespace

extern "C" int ldc_layernorm_mod(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs, int ldy,
                                 long long y_bs, const float* scale, const float* shift, int mod_bs, int mode,
                                 float eps, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  if (B <= 0 || rows <= 0 || D <= 0) return LDC_ERR_ARG;
  if ((D & 3) || D > 64 * 4 * LN_MAX_V4) return LDC_ERR_UNSUPPORTED;
  if ((ldx & 3) || (ldy & 3) || (x_bs & 3) || (y_bs & 3) || (mod_bs & 3)) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(y);
  if (scale) LDC_CHECK_ALIGN16(scale);
  if (shift) LDC_CHECK_ALIGN16(shift);
  if (mode != 0 && mode != 1) return LDC_ERR_UNSUPPORTED;
  dim3 grid(ldc_cdiv(rows, 4), B);
  hipLaunchKernelGGL(layernorm_mod_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, rows, D, ldx,
                     x_bs, ldy, y_bs, scale, shift, mod_bs, mode, eps);
  return ldc_launch_status();
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
  hipLaunchKernelGGL(mean_rows_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, rows, D, ldx, x_bs);
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

extern "C" int ldc_linear_small(const float* x, int x_rows, const float* W, const float* bias, const float* add,
                                int add_rows, float* y, int rows, int N, int K, int act_in, int act_out,
                                void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(W);
  LDC_CHECK_PTR(y);
  if (rows <= 0 || N <= 0 || K <= 0 || x_rows <= 0) return LDC_ERR_ARG;
  if (add && add_rows <= 0) return LDC_ERR_ARG;
  if (K & 3) return LDC_ERR_ALIGN;
  LDC_CHECK_ALIGN16(x);
  LDC_CHECK_ALIGN16(W);
  dim3 grid(ldc_cdiv(N, 4), ldc_cdiv(rows, LS_ROWS));
  hipLaunchKernelGGL(linear_small_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, x_rows, W, bias, add,
                     add_rows, y, rows, N, K, act_in, act_out);
  return ldc_launch_status();
}
