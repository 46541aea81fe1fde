"""Oracle restatement of ``diffusers.EDMDPMSolverMultistepScheduler`` (v0.32.1).

PARITY UNPINNED (third-party source absent from /root/reference).  The reference
instantiates this class with *default* kwargs (``evaluate/pred_rollout.py:49-52``
passes ``"param"`` but ``ladcast/utils.py:52`` reads ``"params"``) and uses it in
``pipelines/edm_sampler.py:56-58,81-91`` (sigmas + preconditioning only) and in
``pipelines/pipeline_AR.py:85-102`` (``scale_model_input`` / ``step``).

All schedule arithmetic is done with fp32 torch CPU tensors exactly as the
original does, so sigma tables and step indices are bit-comparable.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch


class EDMDPMSolverMultistepScheduler:
    order = 1

    def __init__(
        self,
        sigma_min: float = 0.002,
        sigma_max: float = 80.0,
        sigma_data: float = 0.5,
        sigma_schedule: str = "karras",
        num_train_timesteps: int = 1000,
        prediction_type: str = "epsilon",
        rho: float = 7.0,
        solver_order: int = 2,
        thresholding: bool = False,
        algorithm_type: str = "dpmsolver++",
        solver_type: str = "midpoint",
        lower_order_final: bool = True,
        euler_at_final: bool = False,
        final_sigmas_type: str = "zero",
    ):
        if sigma_schedule != "karras":
            raise NotImplementedError("only the karras schedule is on the reference path")
        if algorithm_type != "dpmsolver++" or solver_type != "midpoint":
            raise NotImplementedError("only dpmsolver++ / midpoint is on the reference path")
        if thresholding:
            raise NotImplementedError("thresholding is off on the reference path")
        self.config = SimpleNamespace(
            sigma_min=sigma_min,
            sigma_max=sigma_max,
            sigma_data=sigma_data,
            sigma_schedule=sigma_schedule,
            num_train_timesteps=num_train_timesteps,
            prediction_type=prediction_type,
            rho=rho,
            solver_order=solver_order,
            thresholding=thresholding,
            algorithm_type=algorithm_type,
            solver_type=solver_type,
            lower_order_final=lower_order_final,
            euler_at_final=euler_at_final,
            final_sigmas_type=final_sigmas_type,
        )
        ramp = torch.linspace(0, 1, num_train_timesteps)
        sigmas = self._karras(ramp)
        self.timesteps = self.precondition_noise(sigmas)
        self.sigmas = torch.cat([sigmas, torch.zeros(1)]).to("cpu")
        self.num_inference_steps = None
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None

    # -- schedule ---------------------------------------------------------
    def _karras(self, ramp: torch.Tensor) -> torch.Tensor:
        c = self.config
        lo = c.sigma_min ** (1 / c.rho)
        hi = c.sigma_max ** (1 / c.rho)
        return (hi + ramp * (lo - hi)) ** c.rho

    @property
    def init_noise_sigma(self):
        return (self.config.sigma_max**2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ramp = torch.linspace(0, 1, num_inference_steps)
        sigmas = self._karras(ramp).to(dtype=torch.float32, device=device)
        self.timesteps = self.precondition_noise(sigmas)
        if self.config.final_sigmas_type == "sigma_min":
            last = self.config.sigma_min
        elif self.config.final_sigmas_type == "zero":
            last = 0
        else:
            raise ValueError(self.config.final_sigmas_type)
        self.sigmas = torch.cat(
            [sigmas, torch.tensor([last], dtype=torch.float32, device=device)]
        ).to("cpu")
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None

    # -- EDM preconditioning ------------------------------------------------
    def precondition_inputs(self, sample, sigma):
        c_in = 1 / ((sigma**2 + self.config.sigma_data**2) ** 0.5)
        return sample * c_in

    def precondition_noise(self, sigma):
        if not isinstance(sigma, torch.Tensor):
            sigma = torch.tensor([sigma])
        return 0.25 * torch.log(sigma)

    def precondition_outputs(self, sample, model_output, sigma):
        sd = self.config.sigma_data
        c_skip = sd**2 / (sigma**2 + sd**2)
        if self.config.prediction_type == "epsilon":
            c_out = sigma * sd / (sigma**2 + sd**2) ** 0.5
        elif self.config.prediction_type == "v_prediction":
            c_out = -sigma * sd / (sigma**2 + sd**2) ** 0.5
        else:
            raise ValueError(self.config.prediction_type)
        return c_skip * sample + c_out * model_output

    # -- stepping -------------------------------------------------------------
    def index_for_timestep(self, timestep, schedule_timesteps=None):
        if schedule_timesteps is None:
            schedule_timesteps = self.timesteps
        cand = (schedule_timesteps == timestep).nonzero()
        if len(cand) == 0:
            return len(self.timesteps) - 1
        if len(cand) > 1:
            return cand[1].item()
        return cand[0].item()

    def _init_step_index(self, timestep):
        if self._begin_index is None:
            if isinstance(timestep, torch.Tensor):
                timestep = timestep.to(self.timesteps.device)
            self._step_index = self.index_for_timestep(timestep)
        else:
            self._step_index = self._begin_index

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        return self.precondition_inputs(sample, sigma)

    def _first_order(self, x0, sample):
        sigma_t, sigma_s = self.sigmas[self._step_index + 1], self.sigmas[self._step_index]
        # alpha == 1 (inputs are pre-scaled), lambda = -log(sigma)
        h = (-torch.log(sigma_t)) - (-torch.log(sigma_s))
        return (sigma_t / sigma_s) * sample - (torch.exp(-h) - 1.0) * x0

    def _second_order(self, outs, sample):
        i = self._step_index
        sigma_t, sigma_s0, sigma_s1 = self.sigmas[i + 1], self.sigmas[i], self.sigmas[i - 1]
        lam_t, lam_s0, lam_s1 = -torch.log(sigma_t), -torch.log(sigma_s0), -torch.log(sigma_s1)
        m0, m1 = outs[-1], outs[-2]
        h, h_0 = lam_t - lam_s0, lam_s0 - lam_s1
        r0 = h_0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        return (
            (sigma_t / sigma_s0) * sample
            - (torch.exp(-h) - 1.0) * D0
            - 0.5 * (torch.exp(-h) - 1.0) * D1
        )

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            self._init_step_index(timestep)
        c = self.config
        n = len(self.timesteps)
        lower_order_final = (self._step_index == n - 1) and (
            c.euler_at_final or (c.lower_order_final and n < 15) or c.final_sigmas_type == "zero"
        )
        lower_order_second = (self._step_index == n - 2) and c.lower_order_final and n < 15
        x0 = self.precondition_outputs(sample, model_output, self.sigmas[self._step_index])
        for i in range(c.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        if c.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:
            prev = self._first_order(x0, sample)
        elif c.solver_order == 2 or self.lower_order_nums < 2 or lower_order_second:
            prev = self._second_order(self.model_outputs, sample)
        else:
            raise NotImplementedError("solver_order 3 is not on the reference path")
        if self.lower_order_nums < c.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return SimpleNamespace(prev_sample=prev)

    def add_noise(self, original_samples, noise, timesteps):
        sigmas = self.sigmas.to(device=original_samples.device, dtype=original_samples.dtype)
        sched = self.timesteps.to(original_samples.device)
        idx = [self.index_for_timestep(t, sched) for t in timesteps.to(original_samples.device)]
        sigma = sigmas[idx].flatten()
        while sigma.dim() < original_samples.dim():
            sigma = sigma.unsqueeze(-1)
        return original_samples + noise * sigma


# ---------------------------------------------------------------------------------------------------------------------------
# DDIMScheduler / DDPMScheduler (diffusers v0.32.1): the scheduler classes the reference's pipeline loop is documented against
# (``pipelines/pipeline_AR.py:19-21``; the loop itself, :85-102, only calls set_timesteps / timesteps / scale_model_input / step).
# PARITY UNPINNED like the class above (third-party source absent); restated op by op - every whole-tensor expression below is the
# one diffusers evaluates, in its order, with fp32 0-dim CPU tensors as coefficients.  Independent derivations:
# ``tests/test_oracle_independent_leaves.py`` (DDIM on exact epsilon lands on the closed-form marginal; the DDPM posterior mean /
# variance vs Ho et al. eq. 6-7 in fp64).  Not restated: dynamic thresholding, learned variances, custom timestep lists.
# ---------------------------------------------------------------------------------------------------------------------------
import math  # noqa: E402

import numpy as np  # noqa: E402


def _oracle_randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers.utils.torch_utils.randn_tensor: drawn on the generator's device (CPU here), a list draws (1, ...) per generator"""
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        one = (1,) + tuple(shape[1:])
        return torch.cat([torch.randn(one, generator=g, dtype=dtype) for g in generator], dim=0)
    return torch.randn(tuple(shape), generator=generator, dtype=dtype)


class _DDBase:
    order = 1

    def _betas(self, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, rescale_betas_zero_snr):
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
            betas = torch.tensor([min(1 - bar((i + 1) / num_train_timesteps) / bar(i / num_train_timesteps), 0.999) for i in range(num_train_timesteps)],
                                 dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            raise NotImplementedError("rescale_betas_zero_snr is not restated")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def _spacing(self, n):
        c = self.config
        if c.timestep_spacing == "linspace":
            return np.linspace(0, c.num_train_timesteps - 1, n).round()[::-1].copy().astype(np.int64)
        if c.timestep_spacing == "leading":
            return (np.arange(0, n) * (c.num_train_timesteps // n)).round()[::-1].copy().astype(np.int64) + c.steps_offset
        if c.timestep_spacing == "trailing":
            return np.round(np.arange(c.num_train_timesteps, 0, -c.num_train_timesteps / n)).astype(np.int64) - 1
        raise ValueError(c.timestep_spacing)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _x0_eps(self, model_output, sample, alpha_prod_t, beta_prod_t):
        pt = self.config.prediction_type
        if pt == "epsilon":
            return (sample - beta_prod_t ** (0.5) * model_output) / alpha_prod_t ** (0.5), model_output
        if pt == "sample":
            return model_output, (sample - alpha_prod_t ** (0.5) * model_output) / beta_prod_t ** (0.5)
        if pt == "v_prediction":
            return ((alpha_prod_t**0.5) * sample - (beta_prod_t**0.5) * model_output,
                    (alpha_prod_t**0.5) * model_output + (beta_prod_t**0.5) * sample)
        raise ValueError(pt)

    @staticmethod
    def _t(timestep):
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.reshape(-1)[0].item()
        return int(timestep)


class DDIMScheduler(_DDBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        assert not thresholding
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        self._betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, rescale_betas_zero_snr)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(self._spacing(num_inference_steps))

    def _get_variance(self, timestep, prev_timestep):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - alpha_prod_t_prev) / (1 - alpha_prod_t)) * (1 - alpha_prod_t / alpha_prod_t_prev)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None, return_dict=True):
        timestep = self._t(timestep)
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        pred_original_sample, pred_epsilon = self._x0_eps(model_output, sample, alpha_prod_t, beta_prod_t)
        if self.config.clip_sample:
            pred_original_sample = pred_original_sample.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** (0.5)
        if use_clipped_model_output:
            pred_epsilon = (sample - alpha_prod_t ** (0.5) * pred_original_sample) / beta_prod_t ** (0.5)
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t**2) ** (0.5) * pred_epsilon
        prev_sample = alpha_prod_t_prev ** (0.5) * pred_original_sample + pred_sample_direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = _oracle_randn_tensor(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev_sample = prev_sample + std_dev_t * variance_noise
        if not return_dict:
            return (prev_sample, pred_original_sample)
        return SimpleNamespace(prev_sample=prev_sample, pred_original_sample=pred_original_sample)


class DDPMScheduler(_DDBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 variance_type="fixed_small", clip_sample=True, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading", steps_offset=0,
                 rescale_betas_zero_snr=False):
        assert not thresholding
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        self._betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, rescale_betas_zero_snr)
        self.one = torch.tensor(1.0)
        self.variance_type = variance_type

    def set_timesteps(self, num_inference_steps=None, device=None):
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(self._spacing(num_inference_steps))

    def previous_timestep(self, timestep):
        if self.num_inference_steps:
            index = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            return torch.tensor(-1) if index == self.timesteps.shape[0] - 1 else self.timesteps[index + 1]
        return timestep - 1

    def _get_variance(self, t):
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        current_beta_t = 1 - alpha_prod_t / alpha_prod_t_prev
        variance = torch.clamp((1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t, min=1e-20)
        if self.variance_type == "fixed_small_log":
            variance = torch.exp(0.5 * torch.log(variance))
        elif self.variance_type == "fixed_large":
            variance = current_beta_t
        elif self.variance_type == "fixed_large_log":
            variance = torch.log(current_beta_t)
        return variance

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        t = self._t(timestep)
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        pred_original_sample, _ = self._x0_eps(model_output, sample, alpha_prod_t, beta_prod_t)
        if self.config.clip_sample:
            pred_original_sample = pred_original_sample.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        pred_original_sample_coeff = (alpha_prod_t_prev ** (0.5) * current_beta_t) / beta_prod_t
        current_sample_coeff = current_alpha_t ** (0.5) * beta_prod_t_prev / beta_prod_t
        pred_prev_sample = pred_original_sample_coeff * pred_original_sample + current_sample_coeff * sample
        variance = 0
        if t > 0:
            variance_noise = _oracle_randn_tensor(model_output.shape, generator=generator, dtype=model_output.dtype)
            if self.variance_type == "fixed_small_log":
                variance = self._get_variance(t) * variance_noise
            else:
                variance = (self._get_variance(t) ** 0.5) * variance_noise
        pred_prev_sample = pred_prev_sample + variance
        if not return_dict:
            return (pred_prev_sample, pred_original_sample)
        return SimpleNamespace(prev_sample=pred_prev_sample, pred_original_sample=pred_original_sample)
