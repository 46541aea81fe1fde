"""``DDIMScheduler`` and ``DDPMScheduler`` with the diffusers (v0.32.1) surface.

The reference's pipeline loop is written against "one of ``DDPMScheduler`` or ``DDIMScheduler``"
(pipelines/pipeline_AR.py:19-21) and only touches ``set_timesteps / timesteps / scale_model_input / step``
(:85-102); BASELINE's metric is quoted on "20-step DDIM".  Both classes here keep diffusers' constructor kwargs,
``.config``, ``init_noise_sigma``, ``add_noise`` and ``step(...)`` return values (``prev_sample``,
``pred_original_sample``).

Split of work, as for the EDM class next door: the schedule (betas, cumulative alphas, timestep spacing, the per-step
scalar coefficients) is host arithmetic on fp32 torch CPU tensors with diffusers' own expressions - so indexing and
coefficients are bit-exact - and every whole-tensor update is ONE fused HIP kernel per solver step
(``ldc_ddim_step`` / ``ldc_ddpm_step``, csrc/sampler.hip, built without FMA contraction: the same roundings as the
schedulers' elementwise torch ops).  Noise (DDIM with eta > 0, DDPM's ancestral term) is drawn with the caller's
generator by diffusers' ``randn_tensor`` rule (CPU generator -> drawn on the host, moved to the device).
Not built (raise): dynamic thresholding, learned variances, custom timestep lists.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

from .. import hip
from ..pipelines.torch_utils import randn_tensor

_PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}


def betas_for_alpha_bar(num_diffusion_timesteps, max_beta=0.999, alpha_transform_type="cosine"):
    """diffusers `betas_for_alpha_bar` (squaredcos_cap_v2)"""
    if alpha_transform_type == "cosine":
        fn = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
    elif alpha_transform_type == "exp":
        fn = lambda t: math.exp(t * -12.0)  # noqa: E731
    else:
        raise ValueError(f"Unsupported alpha_transform_type: {alpha_transform_type}")
    betas = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - fn(t2) / fn(t1), max_beta))
    return torch.tensor(betas, dtype=torch.float32)


def rescale_zero_terminal_snr(betas):
    """diffusers `rescale_zero_terminal_snr` (https://arxiv.org/abs/2305.08891, algorithm 1)"""
    alphas = 1.0 - betas
    alphas_bar_sqrt = torch.cumprod(alphas, dim=0).sqrt()
    a0, aT = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt -= aT
    alphas_bar_sqrt *= a0 / (a0 - aT)
    alphas_bar = alphas_bar_sqrt**2
    alphas = alphas_bar[1:] / alphas_bar[:-1]
    alphas = torch.cat([alphas_bar[0:1], alphas])
    return 1 - alphas


def _f32_dev(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError("scheduler tensor ops run in HIP kernels: need a contiguous fp32 device tensor")
    return t


def _host_timestep(timestep) -> int:
    """the integer timestep of a `step` call.  The pipeline loop hands over `t.expand(batch).to(device)`; reading that back would
    stall the host on everything queued, so this package's loop attaches the host value (`tensor.host_value`); any other device
    tensor is read back (one synchronisation per step - correct, slow)."""
    if isinstance(timestep, torch.Tensor):
        hv = getattr(timestep, "host_value", None)
        if hv is not None:
            timestep = hv
        if isinstance(timestep, torch.Tensor):
            flat = timestep.reshape(-1)
            timestep = flat[0].item()
    t = int(timestep)
    if t != timestep:
        raise ValueError(f"DDIM / DDPM timesteps are integers, got {timestep!r}")
    return t


class _BetaSchedule:
    """what DDIMScheduler.__init__ and DDPMScheduler.__init__ share: betas -> alphas_cumprod (fp32, host)"""

    order = 1

    def _init_betas(self, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, rescale_betas_zero_snr, allow_sigmoid):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            self.betas = betas_for_alpha_bar(num_train_timesteps)
        elif beta_schedule == "sigmoid" and allow_sigmoid:
            betas = torch.linspace(-6, 6, num_train_timesteps)
            self.betas = torch.sigmoid(betas) * (beta_end - beta_start) + beta_start
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def _spaced_timesteps(self, num_inference_steps):
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`:"
                f" {c.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {c.num_train_timesteps} timesteps."
            )
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            step_ratio = c.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += c.steps_offset
        elif c.timestep_spacing == "trailing":
            step_ratio = c.num_train_timesteps / num_inference_steps
            ts = np.round(np.arange(c.num_train_timesteps, 0, -step_ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported. Please make sure to choose one of 'leading' or 'trailing'.")
        return ts

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(a_t) x0 + sqrt(1 - a_t) noise, one timestep per sample (training-side surface; one kernel per sample)"""
        out = torch.empty_like(original_samples)
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        for j, t in enumerate(timesteps.reshape(-1).tolist()):
            a, b = ac[int(t)] ** 0.5, (1 - ac[int(t)]) ** 0.5
            hip.axpby_f32(_f32_dev(original_samples[j]), float(a), _f32_dev(noise[j]), float(b), out[j])
        return out

    def __len__(self):
        return self.config.num_train_timesteps

    # what a captured sampling loop needs to know about this object (pipelines/pipeline_AR.py::_graph_loop)
    def graph_signature(self):
        return (type(self).__name__, tuple(sorted((k, str(v)) for k, v in vars(self.config).items())), tuple(int(v) for v in self.timesteps.tolist()))

    def graph_state(self):
        return None

    def set_graph_state(self, state):
        pass


class DDIMScheduler(_BetaSchedule):
    """diffusers.DDIMScheduler (v0.32.1).  `step(..., eta=0)` is deterministic: host scalars + one kernel launch, so a whole sampling
    loop can be captured into one hipGraph (`launch_only`); with `eta > 0` the loop is launched eagerly (fresh noise every step)."""

    launch_only = True

    def __init__(
        self,
        num_train_timesteps: int = 1000,
        beta_start: float = 0.0001,
        beta_end: float = 0.02,
        beta_schedule: str = "linear",
        trained_betas=None,
        clip_sample: bool = True,
        set_alpha_to_one: bool = True,
        steps_offset: int = 0,
        prediction_type: str = "epsilon",
        thresholding: bool = False,
        dynamic_thresholding_ratio: float = 0.995,
        clip_sample_range: float = 1.0,
        sample_max_value: float = 1.0,
        timestep_spacing: str = "leading",
        rescale_betas_zero_snr: bool = False,
    ):
        if thresholding:
            raise NotImplementedError("dynamic thresholding is not built (off by default; the reference never sets it)")
        if prediction_type not in _PRED:
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        self._init_betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, rescale_betas_zero_snr, allow_sigmoid=False)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def _get_variance(self, timestep, prev_timestep):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        return (beta_prod_t_prev / beta_prod_t) * (1 - alpha_prod_t / alpha_prod_t_prev)

    def set_timesteps(self, num_inference_steps: int, device=None):
        ts = self._spaced_timesteps(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(ts)  # lives on the host and drives the loop (diffusers moves it to `device`; the values are the same)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False, generator=None,
             variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        c = self.config
        t = _host_timestep(timestep)
        prev_t = t - c.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        variance = self._get_variance(t, prev_t)
        std_dev_t = eta * variance ** (0.5)
        dir_coef = (1 - alpha_prod_t_prev - std_dev_t**2) ** (0.5)
        sample, model_output = _f32_dev(sample), _f32_dev(model_output)
        noise = None
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            noise = _f32_dev(variance_noise.contiguous())
        x0, prev = torch.empty_like(sample), torch.empty_like(sample)
        hip.ddim_step(sample, model_output, noise, x0, prev, float(alpha_prod_t ** (0.5)), float(beta_prod_t ** (0.5)), float(alpha_prod_t_prev ** (0.5)),
                      float(dir_coef), float(std_dev_t), float(c.clip_sample_range) if c.clip_sample else 0.0, _PRED[c.prediction_type],
                      use_clipped_model_output)
        if not return_dict:
            return (prev, x0)
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DDPMScheduler(_BetaSchedule):
    """diffusers.DDPMScheduler (v0.32.1): ancestral sampling, the noise of every step drawn with the caller's generator - a loop over
    it is launched eagerly (`launch_only = False`)."""

    launch_only = False

    def __init__(
        self,
        num_train_timesteps: int = 1000,
        beta_start: float = 0.0001,
        beta_end: float = 0.02,
        beta_schedule: str = "linear",
        trained_betas=None,
        variance_type: str = "fixed_small",
        clip_sample: bool = True,
        prediction_type: str = "epsilon",
        thresholding: bool = False,
        dynamic_thresholding_ratio: float = 0.995,
        clip_sample_range: float = 1.0,
        sample_max_value: float = 1.0,
        timestep_spacing: str = "leading",
        steps_offset: int = 0,
        rescale_betas_zero_snr: bool = False,
    ):
        if thresholding:
            raise NotImplementedError("dynamic thresholding is not built (off by default; the reference never sets it)")
        if variance_type not in ("fixed_small", "fixed_small_log", "fixed_large", "fixed_large_log"):
            raise NotImplementedError("learned variances need a model with 2 C output channels; the LaDCast transformer has none")
        if prediction_type not in _PRED:
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample` or `v_prediction`  for the DDPMScheduler.")
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        self._init_betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, rescale_betas_zero_snr, allow_sigmoid=True)
        self.one = torch.tensor(1.0)
        self.custom_timesteps = False
        self.variance_type = variance_type

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        if timesteps is not None:
            raise NotImplementedError("custom timestep lists are not built")
        ts = self._spaced_timesteps(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        self.custom_timesteps = False
        self.timesteps = torch.from_numpy(ts)

    def previous_timestep(self, timestep):
        if self.custom_timesteps or self.num_inference_steps:
            index = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            if index == self.timesteps.shape[0] - 1:
                return torch.tensor(-1)
            return self.timesteps[index + 1]
        return timestep - 1

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        current_beta_t = 1 - alpha_prod_t / alpha_prod_t_prev
        variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
        variance = torch.clamp(variance, min=1e-20)
        if variance_type is None:
            variance_type = self.config.variance_type
        if variance_type == "fixed_small_log":
            variance = torch.exp(0.5 * torch.log(variance))
        elif variance_type == "fixed_large":
            variance = current_beta_t
        elif variance_type == "fixed_large_log":
            variance = torch.log(current_beta_t)
        return variance

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True):
        c = self.config
        t = _host_timestep(timestep)
        prev_t = int(self.previous_timestep(t))
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        x0_coef = (alpha_prod_t_prev ** (0.5) * current_beta_t) / beta_prod_t
        sample_coef = current_alpha_t ** (0.5) * beta_prod_t_prev / beta_prod_t
        sample, model_output = _f32_dev(sample), _f32_dev(model_output)
        noise, sd = None, 0.0
        if t > 0:
            noise = _f32_dev(randn_tensor(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype).contiguous())
            v = self._get_variance(t)
            sd = float(v if self.variance_type == "fixed_small_log" else v ** 0.5)
        x0, prev = torch.empty_like(sample), torch.empty_like(sample)
        hip.ddpm_step(sample, model_output, noise, x0, prev, float(alpha_prod_t ** (0.5)), float(beta_prod_t ** (0.5)), float(x0_coef), float(sample_coef),
                      sd, float(c.clip_sample_range) if c.clip_sample else 0.0, _PRED[c.prediction_type])
        if not return_dict:
            return (prev, x0)
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)
