// Layout / small elementwise kernels: channel-major <-> token-major transposes,
// sinusoidal timestep embedding, time-elapsed modulation of temb, per-channel affine.
#include "common.h"

namespace {

// in: (B, C, N) channel-major, out: (B, N, ldo) token-major; 32x32 LDS tile transpose
// (33-float pitch: conflict-free column reads).  Columns [C, ldo) are zero-filled.
__global__ __launch_bounds__(256) void chan_to_token_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int C, int N, int ldo, int fill, int out_split) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
  const float* ib = in + static_cast<long long>(b) * C * N;
  float* ob = out + static_cast<long long>(b) * N * ldo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, n = n0 + tx;
    tile[ty + 8 * i][tx] = (c < C && n < N) ? ib[static_cast<long long>(c) * N + n] : 0.f;
  }
  __syncthreads();
  if (out_split) {
    // split activation format (ladcast_hip.h LDC_GEMM_A_SPLIT): columns 8g..8g+7 in 32 bytes [hi x8 | lo x8] bf16;
    // lane tx (even) packs columns c, c+1: one 4-byte hi word and one 4-byte lo word.  fill % 8 == 0, c0 % 32 == 0.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + ty + 8 * i, c = c0 + tx;
      const float v = tile[tx][ty + 8 * i];
      const float w = __shfl_xor(v, 1);
      if (!(tx & 1) && n < N && c < fill) {
        float r0, r1;
        const unsigned hi = ldc_split_pair(v, w, r0, r1);
        if (out_split == LDC_FMT_BF16) {  // plain bf16 row
          *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(ob + static_cast<long long>(n) * ldo) + 2 * c) = hi;
        } else {
          const unsigned lo = ldc_pack_pair(r0, r1);
          unsigned char* grp = reinterpret_cast<unsigned char*>(ob + static_cast<long long>(n) * ldo + (c & ~7)) + 2 * (c & 7);
          *reinterpret_cast<unsigned*>(grp) = hi;
          *reinterpret_cast<unsigned*>(grp + 16) = lo;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, c = c0 + tx;
    if (n < N && c < fill) ob[static_cast<long long>(n) * ldo + c] = tile[tx][ty + 8 * i];
  }
}

// in: (B, N, ldi) token-major (first C columns used), out: (B, C, N)
__global__ __launch_bounds__(256) void token_to_chan_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int C, int N, int ldi) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ib = in + static_cast<long long>(b) * N * ldi;
  float* ob = out + static_cast<long long>(b) * C * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (n < N && c < C) ? ib[static_cast<long long>(n) * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, n = n0 + tx;
    if (c < C && n < N) ob[static_cast<long long>(c) * N + n] = tile[tx][ty + 8 * i];
  }
}

// out[i][0:128] = cos(t_i f_k), out[i][128:256] = sin(t_i f_k), f_k = exp(-ln(1e4) k / 128)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int n) {
  const int i = blockIdx.x;
  const int k = threadIdx.x;  // 0..127
  if (i >= n) return;
  const float f = expf(-9.210340371976184f * static_cast<float>(k) / 128.0f);
  const float a = t[i] * f;
  out[i * 256 + k] = cosf(a);
  out[i * 256 + 128 + k] = sinf(a);
}

__global__ void temb_modulate_kernel(float* __restrict__ temb, const float* __restrict__ te, int B, int D, int te_rows) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * D) return;
  const int b = idx / D, c = idx - b * D;
  const float* tr = te + static_cast<long long>(b % te_rows) * 2 * D;
  temb[idx] = temb[idx] * (1.f + tr[c]) + tr[D + c];
}

__global__ void chan_affine_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                   const float* __restrict__ sd, float target_std, long long total, int C,
                                   long long inner, int inverse) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = static_cast<int>((i / inner) % C);
  const float v = x[i];
  y[i] = inverse ? (v / target_std) * sd[c] + mean[c] : ((v - mean[c]) / sd[c]) * target_std;
}

}  // namespace

extern "C" int ldc_chan_to_token_split(const float* in, float* out, int B, int C, int N, int ldo, int fill_cols,
                                       int out_split, void* stream) {
  LDC_CHECK_PTR(in);
  LDC_CHECK_PTR(out);
  if (B <= 0 || C <= 0 || N <= 0 || fill_cols < C || ldo < fill_cols) return LDC_ERR_ARG;
  if (out_split && ((fill_cols & 7) || (ldo & 7) || (reinterpret_cast<unsigned long long>(out) & 31ull))) return LDC_ERR_ALIGN;
  dim3 grid(ldc_cdiv(N, 32), ldc_cdiv(fill_cols, 32), B);
  hipLaunchKernelGGL(chan_to_token_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, out, C, N, ldo,
                     fill_cols, out_split);
  return ldc_launch_status();
}

extern "C" int ldc_chan_to_token(const float* in, float* out, int B, int C, int N, int ldo, int fill_cols,
                                 void* stream) {
  return ldc_chan_to_token_split(in, out, B, C, N, ldo, fill_cols, 0, stream);
}

extern "C" int ldc_token_to_chan(const float* in, float* out, int B, int C, int N, int ldi, void* stream) {
  LDC_CHECK_PTR(in);
  LDC_CHECK_PTR(out);
  if (B <= 0 || C <= 0 || N <= 0 || ldi < C) return LDC_ERR_ARG;
  dim3 grid(ldc_cdiv(N, 32), ldc_cdiv(C, 32), B);
  hipLaunchKernelGGL(token_to_chan_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, out, C, N, ldi);
  return ldc_launch_status();
}

extern "C" int ldc_timestep_embedding(const float* t, float* out, int n, void* stream) {
  LDC_CHECK_PTR(t);
  LDC_CHECK_PTR(out);
  if (n <= 0) return LDC_ERR_ARG;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(n), dim3(128), 0, static_cast<hipStream_t>(stream), t, out, n);
  return ldc_launch_status();
}

extern "C" int ldc_temb_modulate(float* temb, const float* te, int B, int D, int te_rows, void* stream) {
  LDC_CHECK_PTR(temb);
  LDC_CHECK_PTR(te);
  if (B <= 0 || D <= 0 || te_rows <= 0) return LDC_ERR_ARG;
  hipLaunchKernelGGL(temb_modulate_kernel, dim3(ldc_cdiv(static_cast<long long>(B) * D, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), temb, te, B, D, te_rows);
  return ldc_launch_status();
}

extern "C" int ldc_chan_affine(const float* x, float* y, const float* mean, const float* std_, float target_std,
                               long long outer, int C, long long inner, int inverse, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  LDC_CHECK_PTR(mean);
  LDC_CHECK_PTR(std_);
  if (outer <= 0 || C <= 0 || inner <= 0) return LDC_ERR_ARG;
  const long long total = outer * C * inner;
  hipLaunchKernelGGL(chan_affine_kernel, dim3(ldc_cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
                     mean, std_, target_std, total, C, inner, inverse);
  return ldc_launch_status();
}
