// Batched "NT" contraction C[b] = epilogue(A[b] . W^T) on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per product -- MI355X_MICROARCH §Matrix
// cores).  This is the workhorse that stands in for every nn.Linear on the token streams of
// the AR transformer (models/LaDCast_3D_model.py:92-94,175-177,214,219,422-424,558-563,1045).
//
// Tiling (MI355X-first, not a warp-tiling port):
//   workgroup = 256 threads = 4 waves in a 2x2 arrangement, one wave per SIMD;
//   block tile 128 x 128, K step 32; each wave owns a 64 x 64 tile = 2x2 MFMA tiles
//   (64 accumulator VGPRs).  A and W tiles are staged HBM -> VGPR (float4, coalesced
//   128-B row segments) -> LDS with a 36-float row pitch, which makes both the
//   ds_write_b128 staging stores and the ds_read_b128 fragment reads bank-conflict-free
//   (checked against the b128 lane groups of MI355X_MICROARCH §LDS).  LDS is double
//   buffered (72 KiB) so two workgroups share a CU and the next tile's global loads are
//   in flight under the current tile's 64 MFMAs per wave.
//   The reduction index is permuted inside a K-group of 8 (lane half h supplies
//   k = 4h + r to MFMA r) so that one ds_read_b128 per operand feeds four MFMAs;
//   a sum is invariant under that permutation.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDP = 36;  // LDS row pitch in floats (32 + 4 pad)
constexpr int STAGE_FLOATS = (BM + BN) * LDP;

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* gate;
  const float* R;
  float* C;
  ldc_gemm_desc d;
  // implicit-GEMM sphere convolution (CONV = true): A is an NHWC image [B][H][W][lda], M = B*H*W,
  // the reduction runs over ks*ks taps x cin channels, W is [N][ks*ks*cin] (tap-major, channel-minor)
  int cH, cW, cin, ks;
};

// Source pixel of tap (ky, kx) for output pixel (h, w) under SphereConv2d's padding and pole rule
// (models/sphere_conv.py:62-129,174-192): rows beyond a pole are the first/last p rows mirrored and
// rolled by W/2; columns wrap; output row 0 (H-1) sees the first (last) p kernel rows flipped
// horizontally, which equals gathering the mirrored column with the unflipped weight.
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

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.d.M, N = p.d.N, K = p.d.K;
  const int bm = blockIdx.y, bn = blockIdx.x, b = blockIdx.z;

  const float* __restrict__ A = p.A + static_cast<long long>(b) * p.d.a_bs;
  const float* __restrict__ W = p.W;
  const int lda = p.d.lda, ldw = p.d.ldw;

  // staging map: thread t moves float4 column c4 = t&7 of rows (t>>3) + 32*i
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  float4 ra[4], rb[4];

  // CONV: per staged row, its image base pixel index and (h, w); k-tiles never straddle a tap
  int pix_b[4], pix_h[4], pix_w[4];
  int ktpt = 1;  // k-tiles per tap
  if constexpr (CONV) {
    ktpt = (p.cin + BK - 1) / BK;
    const int hw = p.cH * p.cW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = bm * BM + r0 + 32 * i;
      const int gmc = gm < M ? gm : 0;
      const int bimg = gmc / hw;
      const int rem = gmc - bimg * hw;
      pix_b[i] = bimg * hw;
      pix_h[i] = rem / p.cW;
      pix_w[i] = rem - pix_h[i] * p.cW;
    }
  }

  auto gload = [&](int kt) {
    if constexpr (CONV) {
      const int tap = kt / ktpt;
      const int kc = (kt - tap * ktpt) * BK + c4 * 4;  // channel offset inside the tap
      const bool kin = kc < p.cin;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const int kw = tap * p.cin + kc;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        const int gm = bm * BM + row;
        const int gn = bn * BN + row;
        const int src = pix_b[i] + sphere_src_pixel(pix_h[i], pix_w[i], ky, kx, p.cH, p.cW, p.ks);
        ra[i] = (kin && gm < M) ? *reinterpret_cast<const float4*>(A + static_cast<long long>(src) * lda + kc)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[i] = (kin && gn < N) ? *reinterpret_cast<const float4*>(W + static_cast<long long>(gn) * ldw + kw)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int kk = kt * BK + c4 * 4;
      const bool kin = kk < K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        const int gm = bm * BM + row;
        const int gn = bn * BN + row;
        ra[i] = (kin && gm < M) ? *reinterpret_cast<const float4*>(A + static_cast<long long>(gm) * lda + kk)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[i] = (kin && gn < N) ? *reinterpret_cast<const float4*>(W + static_cast<long long>(gn) * ldw + kk)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto sstore = [&](int stage) {
    float* As = smem + stage * STAGE_FLOATS;
    float* Bs = As + BM * LDP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<float4*>(As + row * LDP + c4 * 4) = ra[i];
      *reinterpret_cast<float4*>(Bs + row * LDP + c4 * 4) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = CONV ? ktpt * p.ks * p.ks : (K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    const float* As = smem + (kt & 1) * STAGE_FLOATS;
    const float* Bs = As + BM * LDP;
    const float* a_base = As + (wm * 64 + frag_row) * LDP + frag_k;
    const float* b_base = Bs + (wn * 64 + frag_row) * LDP + frag_k;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const float4 a0 = *reinterpret_cast<const float4*>(a_base + kg * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDP + kg * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(b_base + kg * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(b_base + 32 * LDP + kg * 8);
      const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
      const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][r], bv[j][r], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore((kt + 1) & 1);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* __restrict__ C = p.C + static_cast<long long>(b) * p.d.c_bs;
  const float* __restrict__ R = p.R ? p.R + static_cast<long long>(b) * p.d.r_bs : nullptr;
  const float* __restrict__ gate = p.gate ? p.gate + static_cast<long long>(b) * p.d.gate_bs : nullptr;
  const int act = p.d.act;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = bn * BN + wn * 64 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const float bias_n = p.bias ? p.bias[n] : 0.f;
    const float gate_n = gate ? gate[n] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        float v = acc[i][j][r] + bias_n;
        v = ldc_apply_act(v, act);
        if (gate) v *= gate_n;
        if (R) v += R[static_cast<long long>(m) * p.d.ldr + n];
        C[static_cast<long long>(m) * p.d.ldc + n] = v;
      }
    }
  }
}

}  // namespace

extern "C" int ldc_sizeof_gemm_desc(void) { return static_cast<int>(sizeof(ldc_gemm_desc)); }

extern "C" int ldc_gemm_bias_act(const float* A, const float* W, const float* bias, const float* gate,
                                 const float* R, float* C, const ldc_gemm_desc* d, void* stream) {
  LDC_CHECK_PTR(A);
  LDC_CHECK_PTR(W);
  LDC_CHECK_PTR(C);
  LDC_CHECK_PTR(d);
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return LDC_ERR_ARG;
  LDC_CHECK_ALIGN16(A);
  LDC_CHECK_ALIGN16(W);
  if ((d->K & 3) || (d->lda & 3) || (d->ldw & 3) || (d->a_bs & 3)) return LDC_ERR_ALIGN;
  if (d->act < LDC_ACT_NONE || d->act > LDC_ACT_RELU) return LDC_ERR_UNSUPPORTED;
  if (d->flags != 0) return LDC_ERR_UNSUPPORTED;  // split activation formats: ldc_gemm_grouped_bf16x3 only
  if (d->batch > 65535 || ldc_cdiv(d->M, BM) > 65535) return LDC_ERR_UNSUPPORTED;
  GemmArgs p{A, W, bias, gate, R, C, *d, 0, 0, 0, 0};
  dim3 grid(ldc_cdiv(d->N, BN), ldc_cdiv(d->M, BM), d->batch);
  const size_t lds = 2 * STAGE_FLOATS * sizeof(float);
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f32_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL(gemm_nt_f32_kernel<false>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return ldc_launch_status();
}

extern "C" int ldc_sphere_conv_nhwc(const float* X, const float* Wt, const float* bias, const float* R, float* Y,
                                    int B, int H, int W, int cin, int ldx, int cout, int ldy, int ldr, int ksize,
                                    int act, void* stream) {
  LDC_CHECK_PTR(X);
  LDC_CHECK_PTR(Wt);
  LDC_CHECK_PTR(Y);
  if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return LDC_ERR_ARG;
  if (ksize != 3 && ksize != 5) return LDC_ERR_UNSUPPORTED;
  if ((W & 1) || H < 2 || H < ksize / 2) return LDC_ERR_UNSUPPORTED;  // reference asserts even width
  LDC_CHECK_ALIGN16(X);
  LDC_CHECK_ALIGN16(Wt);
  if ((cin & 3) || (ldx & 3) || ldx < cin || ldy < cout) return LDC_ERR_ALIGN;
  if (act < LDC_ACT_NONE || act > LDC_ACT_RELU) return LDC_ERR_UNSUPPORTED;
  const long long M = static_cast<long long>(B) * H * W;
  if (M > 0x7fffffffLL || ldc_cdiv(M, BM) > 65535) return LDC_ERR_UNSUPPORTED;
  ldc_gemm_desc d{};
  d.M = static_cast<int>(M);
  d.N = cout;
  d.K = ksize * ksize * cin;
  d.batch = 1;
  d.lda = ldx;
  d.ldw = ksize * ksize * cin;
  d.ldc = ldy;
  d.ldr = ldr;
  d.act = act;
  GemmArgs p{X, Wt, bias, nullptr, R, Y, d, H, W, cin, ksize};
  dim3 grid(ldc_cdiv(cout, BN), ldc_cdiv(M, BM), 1);
  const size_t lds = 2 * STAGE_FLOATS * sizeof(float);
  static const bool attr_set = [&] {  // once per process; thread-safe (C++11 static initialisation)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f32_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL(gemm_nt_f32_kernel<true>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return ldc_launch_status();
}
