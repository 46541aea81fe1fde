// Ensemble scoring of a decoded forecast: CRPS skill / spread / total, ensemble-mean
// latitude-weighted MSE and anomaly correlation, per channel, for one lead time.
//   reference: ladcast/evaluate/utils.py:40-149 (pointwise_crps_skill, pointwise_crps_spread, get_crps, get_acc) and the
//   per-lead-time block of ladcast/evaluate/evaluate_ens_gpu.py:339-425, which runs ~40 torch ops (a sort over the
//   ensemble axis among them) over the (ens, C, H, W) slice = ~15 passes over 0.5 GB at ens = 50.
// Here the slice is read ONCE (HBM-bound: M x C x H x W x 4 bytes): one thread per grid point keeps its M members in
// registers, sorts them with a fully unrolled, pruned odd-even merge network (compile-time register indices), and the workgroup
// reduces the seven weighted sums of its points; a second tiny kernel adds the per-workgroup partials in a fixed
// order (deterministic, no float atomics) and applies the reference's mean / nanmean rules.
//   spread(point) = 2 / (M (M - 1)) * sum_i (2 i - M - 1) x_(i)      (x_(1) <= ... <= x_(M))
//   skill(point)  = mean_i |truth - x_i|,   crps = skill - spread / 2
//   mse = mean_hw[(mean_i x_i - truth)^2 w(lat)],  acc = <fa ta w> / sqrt(<fa^2 w> <ta^2 w>), fa = mean_i x_i - clim
// Points where any member, the truth (or the climatology, for ACC) is NaN are NaN in the reference's maps: channel
// `nan_channel` (SST: NaN over land) averages with nanmean, every other channel with mean (one NaN -> NaN), ACC
// always with nanmean -- reproduced through the valid-point counts.
#include <math.h>

#include "common.h"

namespace {

constexpr int NQ = 8;  // partial sums per workgroup: w*skill, w*spread, w*crps, w*se, w*fa*ta, w*fa^2, w*ta^2 (ACC-valid), counts
constexpr int TPB = 256;

struct ScoreArgs {
  const float* fc;     // [M] x [C] x [HW], strides below (elements)
  const float* truth;  // [C] x [HW]
  const float* clim;   // [C] x [HW] or nullptr
  const float* lat_w;  // [H]
  long long fc_ms, fc_cs, tr_cs, cl_cs;
  int M, C, H, W;
  float* skill_map;   // optional [C][HW]
  float* spread_map;  // optional [C][HW]
  float* part;        // [C][nblk][NQ + 1]
  int nblk;
};

// Batcher's odd-even merge sort for NP = 2^k registers, fully unrolled (compile-time register indices).  Every
// comparator is ascending (min to the lower index), so comparators that touch an index >= NUSE -- registers that hold
// the +inf padding behind the members -- are no-ops and are pruned at compile time.
template <int NP, int NUSE>
__device__ __forceinline__ void sort_network(float (&x)[NP]) {
#pragma unroll
  for (int p = 1; p < NP; p <<= 1) {
#pragma unroll
    for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
      for (int j = k % p; j <= NP - 1 - k; j += 2 * k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          const int lo_i = i + j, hi_i = i + j + k;
          if (hi_i < NUSE && (lo_i / (2 * p)) == (hi_i / (2 * p))) {
            const float a = x[lo_i], b = x[hi_i];
            x[lo_i] = fminf(a, b);
            x[hi_i] = fmaxf(a, b);
          }
        }
      }
    }
  }
}

__device__ __forceinline__ float wave_total(float v) {  // fixed butterfly order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NP, int NUSE>  // NUSE = members rounded up to a multiple of 8 (<= NP = next power of two)
__global__ __launch_bounds__(TPB) void ensemble_scores_kernel(ScoreArgs a) {
  __shared__ float red[4][NQ + 7];
  const int c = blockIdx.y;
  const int HW = a.H * a.W;
  const int p = blockIdx.x * TPB + threadIdx.x;
  const bool in = p < HW;
  const int pp = in ? p : 0;
  const float* f = a.fc + static_cast<long long>(c) * a.fc_cs + pp;
  float x[NP];
  bool nan_m = false;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i < NUSE && i < a.M) {
      const float v = f[static_cast<long long>(i) * a.fc_ms];
      x[i] = v;
      sum += v;  // same left-to-right order as torch's mean over dim 0 for small M is not guaranteed; tolerance in tests
      nan_m = nan_m || (v != v);
    } else {
      x[i] = INFINITY;  // sorts behind every member
    }
  }
  const float t = a.truth[static_cast<long long>(c) * a.tr_cs + pp];
  const float w = a.lat_w[pp / a.W];
  const float Mf = static_cast<float>(a.M);
  float skill = 0.f;
#pragma unroll
  for (int i = 0; i < NUSE; ++i)
    if (i < a.M) skill += fabsf(t - x[i]);
  skill /= Mf;
  float spread = 0.f;
  if (a.M >= 2) {
    sort_network<NP, NUSE>(x);
    float ws = 0.f;
#pragma unroll
    for (int i = 0; i < NUSE; ++i)
      if (i < a.M) ws += x[i] * (2.0f * static_cast<float>(i + 1) - Mf - 1.0f);
    spread = 2.0f * ws / (Mf * (Mf - 1.0f));
  }
  const float nanv = __builtin_nanf("");
  if (nan_m) spread = nanv;  // the sort would have dropped the NaNs
  if (in) {
    if (a.skill_map) a.skill_map[static_cast<long long>(c) * HW + p] = skill;
    if (a.spread_map) a.spread_map[static_cast<long long>(c) * HW + p] = spread;
  }
  const float mean = sum / Mf;
  const float se = (mean - t) * (mean - t);
  const float crps = skill - 0.5f * spread;
  // validity per reduced quantity (NaN propagates through the reference's elementwise ops)
  const bool v_skill = in && skill == skill, v_spread = in && spread == spread, v_crps = in && crps == crps, v_se = in && se == se;
  float fa = 0.f, ta = 0.f;
  bool v_acc = false;
  if (a.clim) {
    const float cl = a.clim[static_cast<long long>(c) * a.cl_cs + pp];
    fa = mean - cl;
    ta = t - cl;
    // the reference takes three independent nanmeans; a point is dropped from each where that product is NaN
    v_acc = in;
  }
  const float q0 = v_skill ? skill * w : 0.f, q1 = v_spread ? spread * w : 0.f, q2 = v_crps ? crps * w : 0.f, q3 = v_se ? se * w : 0.f;
  const float fta = fa * ta * w, ffa = fa * fa * w, tta = ta * ta * w;
  const bool v4 = v_acc && fta == fta, v5 = v_acc && ffa == ffa, v6 = v_acc && tta == tta;
  // fixed-order workgroup reduction of the 15 partial sums: lanes by butterfly, then the 4 wave totals in order
  float v[NQ + 7];
  v[0] = q0; v[1] = q1; v[2] = q2; v[3] = q3;
  v[4] = v4 ? fta : 0.f; v[5] = v5 ? ffa : 0.f; v[6] = v6 ? tta : 0.f;
  // counts of valid points (exact in fp32 up to 2^24 points)
  v[7] = v_skill ? 1.f : 0.f; v[8] = v_spread ? 1.f : 0.f; v[9] = v_crps ? 1.f : 0.f; v[10] = v_se ? 1.f : 0.f;
  v[11] = v4 ? 1.f : 0.f; v[12] = v5 ? 1.f : 0.f; v[13] = v6 ? 1.f : 0.f; v[14] = in ? 1.f : 0.f;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NQ + 7; ++i) {
    const float t_ = wave_total(v[i]);
    if ((threadIdx.x & 63) == 0) red[wave][i] = t_;
  }
  __syncthreads();
  if (threadIdx.x < NQ + 7) {
    const int i = threadIdx.x;
    float* dst = a.part + (static_cast<long long>(c) * a.nblk + blockIdx.x) * (NQ + 7);
    dst[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// out: [5][C] = acc, mse, crps_spread, crps_skill, crps.  One wave per channel: lane j adds the partial records
// j, j + 64, ... in order, then a fixed butterfly (deterministic).
__global__ __launch_bounds__(64) void ensemble_scores_finish_kernel(const float* __restrict__ part, int nblk, int C, int nan_channel,
                                                                    int has_clim, float* __restrict__ out) {
  const int c = blockIdx.x;
  float s[NQ + 7];
#pragma unroll
  for (int i = 0; i < NQ + 7; ++i) s[i] = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    const float* src = part + (static_cast<long long>(c) * nblk + b) * (NQ + 7);
#pragma unroll
    for (int i = 0; i < NQ + 7; ++i) s[i] += src[i];
  }
#pragma unroll
  for (int i = 0; i < NQ + 7; ++i) s[i] = wave_total(s[i]);
  if (threadIdx.x != 0) return;
  const float nanv = __builtin_nanf("");
  const float total = s[14];
  // mean: any NaN point -> NaN; nanmean: average over the valid points (all invalid -> NaN, as torch.nanmean)
  auto avg = [&](float sum, float cnt) {
    if (c == nan_channel) return cnt > 0.f ? sum / cnt : nanv;
    return cnt == total ? sum / total : nanv;
  };
  const float skill = avg(s[0], s[7]), spread = avg(s[1], s[8]), crps = avg(s[2], s[9]), mse = avg(s[3], s[10]);
  float acc = nanv;
  if (has_clim) {
    const float n4 = s[11] > 0.f ? s[4] / s[11] : nanv, n5 = s[12] > 0.f ? s[5] / s[12] : nanv, n6 = s[13] > 0.f ? s[6] / s[13] : nanv;
    acc = n4 / sqrtf(n5 * n6);
  }
  out[0 * C + c] = acc;
  out[1 * C + c] = mse;
  out[2 * C + c] = spread;
  out[3 * C + c] = skill;
  out[4 * C + c] = crps;
}

}  // namespace

extern "C" long long ldc_ensemble_scores_workspace_bytes(int C, int H, int W) {
  if (C <= 0 || H <= 0 || W <= 0) return 0;
  return static_cast<long long>(C) * ldc_cdiv(static_cast<long long>(H) * W, TPB) * (NQ + 7) * static_cast<long long>(sizeof(float));
}

extern "C" int ldc_ensemble_scores(const float* forecast, long long member_stride, long long channel_stride, const float* truth,
                                   long long truth_channel_stride, const float* clim, long long clim_channel_stride,
                                   const float* lat_weight, int M, int C, int H, int W, int nan_channel, float* out,
                                   float* skill_map, float* spread_map, void* workspace, long long workspace_bytes,
                                   void* stream) {
  LDC_CHECK_PTR(forecast);
  LDC_CHECK_PTR(truth);
  LDC_CHECK_PTR(lat_weight);
  LDC_CHECK_PTR(out);
  LDC_CHECK_PTR(workspace);
  if (M <= 0 || C <= 0 || H <= 0 || W <= 0) return LDC_ERR_ARG;
  if (M > 64 || C > 65535) return LDC_ERR_UNSUPPORTED;
  if (workspace_bytes < ldc_ensemble_scores_workspace_bytes(C, H, W)) return LDC_ERR_ARG;
  ScoreArgs a{};
  a.fc = forecast;
  a.truth = truth;
  a.clim = clim;
  a.lat_w = lat_weight;
  a.fc_ms = member_stride;
  a.fc_cs = channel_stride;
  a.tr_cs = truth_channel_stride;
  a.cl_cs = clim_channel_stride;
  a.M = M; a.C = C; a.H = H; a.W = W;
  a.skill_map = skill_map;
  a.spread_map = spread_map;
  a.part = static_cast<float*>(workspace);
  a.nblk = static_cast<int>(ldc_cdiv(static_cast<long long>(H) * W, TPB));
  dim3 grid(a.nblk, C);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M <= 8) hipLaunchKernelGGL((ensemble_scores_kernel<8, 8>), grid, dim3(TPB), 0, s, a);
  else if (M <= 16) hipLaunchKernelGGL((ensemble_scores_kernel<16, 16>), grid, dim3(TPB), 0, s, a);
  else if (M <= 24) hipLaunchKernelGGL((ensemble_scores_kernel<32, 24>), grid, dim3(TPB), 0, s, a);
  else if (M <= 32) hipLaunchKernelGGL((ensemble_scores_kernel<32, 32>), grid, dim3(TPB), 0, s, a);
  else if (M <= 40) hipLaunchKernelGGL((ensemble_scores_kernel<64, 40>), grid, dim3(TPB), 0, s, a);
  else if (M <= 48) hipLaunchKernelGGL((ensemble_scores_kernel<64, 48>), grid, dim3(TPB), 0, s, a);
  else if (M <= 56) hipLaunchKernelGGL((ensemble_scores_kernel<64, 56>), grid, dim3(TPB), 0, s, a);
  else hipLaunchKernelGGL((ensemble_scores_kernel<64, 64>), grid, dim3(TPB), 0, s, a);
  int st = ldc_launch_status();
  if (st != LDC_OK) return st;
  hipLaunchKernelGGL(ensemble_scores_finish_kernel, dim3(C), dim3(64), 0, s, a.part, a.nblk, C, nan_channel,
                     clim != nullptr ? 1 : 0, out);
  return ldc_launch_status();
}
