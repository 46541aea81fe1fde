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
