// Sampler state updates.  Built with -ffp-contract=off so that every product and sum rounds
// exactly where torch's elementwise kernels round: given the same model output F these
// updates are bit-identical to pipelines/edm_sampler.py:60-113 (fp64 state) and to the
// fp32 arithmetic of diffusers.EDMDPMSolverMultistepScheduler.step.
#include "common.h"

namespace {

__global__ void edm_scale_kernel(const double* __restrict__ x, double c_in, float* __restrict__ out, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = static_cast<float>(x[i] * c_in);
}

__global__ void edm_init_kernel(const float* __restrict__ noise, double sigma0, double* __restrict__ x, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) x[i] = static_cast<double>(noise[i]) * sigma0;
}

// stochastic churn (pipelines/edm_sampler.py:67-76): x_hat = x_cur + coef * noise, fp64, coef = sqrt(t_hat^2 - t_cur^2) * S_noise
__global__ void edm_churn_kernel(const double* __restrict__ x_cur, const double* __restrict__ noise, double coef,
                                 double* __restrict__ x_hat, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) x_hat[i] = x_cur[i] + coef * noise[i];
}

__global__ void edm_euler_kernel(const double* __restrict__ x_hat, const float* __restrict__ F, double c_skip,
                                 double c_out, double t_hat, double dt, double* __restrict__ x_next,
                                 double* __restrict__ d_cur, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = x_hat[i];
  const double den = c_skip * x + c_out * static_cast<double>(F[i]);
  const double d = (x - den) / t_hat;
  d_cur[i] = d;
  x_next[i] = x + dt * d;
}

__global__ void edm_heun_kernel(const double* __restrict__ x_hat, double* __restrict__ x_next,
                                const float* __restrict__ F, const double* __restrict__ d_cur, double c_skip,
                                double c_out, double t_next, double dt, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double xn = x_next[i];
  const double den = c_skip * xn + c_out * static_cast<double>(F[i]);
  const double dp = (xn - den) / t_next;
  x_next[i] = x_hat[i] + dt * (0.5 * d_cur[i] + 0.5 * dp);
}

__global__ void f64_to_f32_kernel(const double* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = static_cast<float>(x[i]);
}

__global__ void dpm_step_kernel(const float* __restrict__ sample, const float* __restrict__ F,
                                const float* __restrict__ m1, float* __restrict__ x0, float* __restrict__ prev,
                                float c_skip, float c_out, float a, float b, float inv_r0, int order, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = sample[i];
  const float m0 = c_skip * s + c_out * F[i];
  float r = a * s - b * m0;
  if (order == 2) {
    const float d1 = inv_r0 * (m0 - m1[i]);
    r = r - (0.5f * b) * d1;
  }
  x0[i] = m0;
  prev[i] = r;
}

// diffusers DDIMScheduler.step / DDPMScheduler.step (the scheduler classes the reference pipeline names, pipelines/pipeline_AR.py:19-21,
// 100-102) as ONE fused fp32 kernel per solver step; every product, sum and quotient rounds where the schedulers' elementwise torch
// ops round (no contraction: this file is built with -ffp-contract=off), the scalar coefficients come from the host's fp32 0-dim
// tensor arithmetic.  pred: 0 epsilon, 1 sample, 2 v_prediction.
//   x0   = (s - sb F) / sa | F | sa s - sb F           [clamped to +-clip when clip > 0]
//   DDIM: eps = F | (s - sa x0) / sb | sa F + sb s     (recomputed from the clamped x0 when reclip);  prev = sap x0 + cdir eps [+ sd noise]
//   DDPM: prev = c0 x0 + c1 s [+ sd noise]
struct DdStepArgs {
  const float* sample;
  const float* F;
  const float* noise;  // nullptr: no noise term
  float* x0;
  float* prev;
  float sa, sb;        // alpha_prod_t ** 0.5, beta_prod_t ** 0.5
  float c0, c1;        // DDIM: alpha_prod_t_prev ** 0.5, direction coefficient; DDPM: pred_original_sample_coeff, current_sample_coeff
  float sd;            // noise scale
  float clip;          // <= 0: no clamp
  int pred, ddpm, reclip;
};

__global__ void dd_step_kernel(DdStepArgs p, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = p.sample[i], f = p.F[i];
  float x0, eps;
  if (p.pred == 0) {
    x0 = (s - p.sb * f) / p.sa;
    eps = f;
  } else if (p.pred == 1) {
    x0 = f;
    eps = (s - p.sa * x0) / p.sb;
  } else {
    x0 = p.sa * s - p.sb * f;
    eps = p.sa * f + p.sb * s;
  }
  if (p.clip > 0.f) x0 = fminf(fmaxf(x0, -p.clip), p.clip);
  float r;
  if (p.ddpm) {
    r = p.c0 * x0 + p.c1 * s;
  } else {
    if (p.reclip) eps = (s - p.sa * x0) / p.sb;
    r = p.c0 * x0 + p.c1 * eps;
  }
  if (p.noise) r = r + p.sd * p.noise[i];
  p.x0[i] = x0;
  p.prev[i] = r;
}

__global__ void scale_f32_kernel(const float* __restrict__ x, float s, float* __restrict__ y, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * s;
}

__global__ void axpby_f32_kernel(const float* __restrict__ x, float a, const float* __restrict__ y, float b,
                                 float* __restrict__ out, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * x[i] + b * y[i];
}

inline dim3 grid1d(long long n) { return dim3(ldc_cdiv(n, 256)); }

}  // namespace

#define LDC_LAUNCH_1D(kern, n, ...)                                                                        \
  do {                                                                                                     \
    if ((n) <= 0) return LDC_ERR_ARG;                                                                      \
    hipLaunchKernelGGL(kern, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), __VA_ARGS__, (n)); \
    return ldc_launch_status();                                                                            \
  } while (0)

extern "C" int ldc_edm_scale_f64_to_f32(const double* x, double c_in, float* out, long long n, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(out);
  LDC_LAUNCH_1D(edm_scale_kernel, n, x, c_in, out);
}
extern "C" int ldc_edm_init_state(const float* noise, double sigma0, double* x, long long n, void* stream) {
  LDC_CHECK_PTR(noise);
  LDC_CHECK_PTR(x);
  LDC_LAUNCH_1D(edm_init_kernel, n, noise, sigma0, x);
}
extern "C" int ldc_edm_churn(const double* x_cur, const double* noise, double coef, double* x_hat, long long n, void* stream) {
  LDC_CHECK_PTR(x_cur);
  LDC_CHECK_PTR(noise);
  LDC_CHECK_PTR(x_hat);
  LDC_LAUNCH_1D(edm_churn_kernel, n, x_cur, noise, coef, x_hat);
}
extern "C" int ldc_edm_euler(const double* x_hat, const float* F, double c_skip, double c_out, double t_hat, double dt,
                             double* x_next, double* d_cur, long long n, void* stream) {
  LDC_CHECK_PTR(x_hat);
  LDC_CHECK_PTR(F);
  LDC_CHECK_PTR(x_next);
  LDC_CHECK_PTR(d_cur);
  LDC_LAUNCH_1D(edm_euler_kernel, n, x_hat, F, c_skip, c_out, t_hat, dt, x_next, d_cur);
}
extern "C" int ldc_edm_heun(const double* x_hat, double* x_next, const float* F, const double* d_cur, double c_skip,
                            double c_out, double t_next, double dt, long long n, void* stream) {
  LDC_CHECK_PTR(x_hat);
  LDC_CHECK_PTR(x_next);
  LDC_CHECK_PTR(F);
  LDC_CHECK_PTR(d_cur);
  LDC_LAUNCH_1D(edm_heun_kernel, n, x_hat, x_next, F, d_cur, c_skip, c_out, t_next, dt);
}
extern "C" int ldc_f64_to_f32(const double* x, float* y, long long n, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  LDC_LAUNCH_1D(f64_to_f32_kernel, n, x, y);
}
extern "C" int ldc_dpm_step(const float* sample, const float* F, const float* m1, float* x0, float* prev, float c_skip,
                            float c_out, float a, float b, float inv_r0, int order, long long n, void* stream) {
  LDC_CHECK_PTR(sample);
  LDC_CHECK_PTR(F);
  LDC_CHECK_PTR(x0);
  LDC_CHECK_PTR(prev);
  if (order != 1 && order != 2) return LDC_ERR_UNSUPPORTED;
  if (order == 2 && m1 == nullptr) return LDC_ERR_ARG;
  LDC_LAUNCH_1D(dpm_step_kernel, n, sample, F, m1, x0, prev, c_skip, c_out, a, b, inv_r0, order);
}
static int dd_step(const float* sample, const float* F, const float* noise, float* x0, float* prev, float sa, float sb, float c0, float c1,
                   float sd, float clip, int pred, int ddpm, int reclip, long long n, void* stream) {
  LDC_CHECK_PTR(sample);
  LDC_CHECK_PTR(F);
  LDC_CHECK_PTR(x0);
  LDC_CHECK_PTR(prev);
  if (pred < 0 || pred > 2) return LDC_ERR_UNSUPPORTED;
  DdStepArgs a{sample, F, noise, x0, prev, sa, sb, c0, c1, sd, clip, pred, ddpm, reclip};
  LDC_LAUNCH_1D(dd_step_kernel, n, a);
}
extern "C" int ldc_ddim_step(const float* sample, const float* F, const float* noise, float* x0, float* prev, float sqrt_alpha_t,
                             float sqrt_beta_t, float sqrt_alpha_prev, float dir_coef, float std_dev, float clip_range,
                             int prediction_type, int use_clipped_model_output, long long n, void* stream) {
  return dd_step(sample, F, noise, x0, prev, sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, dir_coef, std_dev, clip_range, prediction_type, 0,
                 use_clipped_model_output, n, stream);
}
extern "C" int ldc_ddpm_step(const float* sample, const float* F, const float* noise, float* x0, float* prev, float sqrt_alpha_t,
                             float sqrt_beta_t, float x0_coef, float sample_coef, float std_dev, float clip_range, int prediction_type,
                             long long n, void* stream) {
  return dd_step(sample, F, noise, x0, prev, sqrt_alpha_t, sqrt_beta_t, x0_coef, sample_coef, std_dev, clip_range, prediction_type, 1, 0, n,
                 stream);
}
extern "C" int ldc_scale_f32(const float* x, float s, float* y, long long n, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  LDC_LAUNCH_1D(scale_f32_kernel, n, x, s, y);
}
extern "C" int ldc_axpby_f32(const float* x, float a, const float* y, float b, float* out, long long n, void* stream) {
  LDC_CHECK_PTR(x);
  LDC_CHECK_PTR(y);
  LDC_CHECK_PTR(out);
  LDC_LAUNCH_1D(axpby_f32_kernel, n, x, a, y, b, out);
}
