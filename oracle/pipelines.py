"""Oracle restatement of the rollout driver and samplers
(pipelines/edm_sampler.py:10-120, pipelines/pipeline_AR.py:26-107,
pipelines/utils.py:51-80,250-742, dataloader/utils.py:223-269,
dataloader/ar_dataloder.py:11-18).

The xarray dataset access of ``roll_out_serial`` is factored behind plain tensors
(the tensor contract of SURVEY §8 A0); the arithmetic and the ordering of every
tensor operation follow the reference line by line.

PINNED (tests/test_oracle_reference_pins.py, fixtures from the reference's own code via
tests/golden/make_golden.py::sampler_fixtures): edm_AR_sampler, AutoRegressive2DPipeline.__call__,
ensemble_AR_sampler, the latent transforms.  roll_out_serial / decode_latent_ens (xarray-bound in the
reference) are restatements checked by structure only.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass
from datetime import datetime, timedelta
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from .layers import randn_tensor


# -- transforms (dataloader/utils.py:223-269) ---------------------------------
def normalize_transform_3D(sample, mean, std, target_std=1):
    if not isinstance(mean, torch.Tensor):
        mean = torch.tensor(mean, device=sample.device)
        std = torch.tensor(std, device=sample.device)
    return ((sample - mean[:, None, None, None]) / std[:, None, None, None]) * target_std


def inverse_normalize_transform_3D(sample, mean, std, target_std=1):
    if not isinstance(mean, torch.Tensor):
        mean = torch.tensor(mean, device=sample.device)
        std = torch.tensor(std, device=sample.device)
    return (sample / target_std) * std[:, None, None, None] + mean[:, None, None, None]


def get_transform_3D(transform, args):
    if transform == "normalize":
        if "target_std" in args:
            return lambda x: normalize_transform_3D(x, args["mean"], args["std"], args["target_std"])
        return lambda x: normalize_transform_3D(x, args["mean"], args["std"])
    if transform is None:
        return lambda x: x
    raise NotImplementedError(f"Transform: {transform} not implemented.")


def get_inv_transform_3D(transform, args):
    if transform == "normalize":
        if "target_std" in args:
            return lambda x: inverse_normalize_transform_3D(x, args["mean"], args["std"], args["target_std"])
        return lambda x: inverse_normalize_transform_3D(x, args["mean"], args["std"])
    if transform is None:
        return lambda x: x
    raise NotImplementedError(f"Transform: {transform} not implemented.")


def convert_datetime_to_int(dt: datetime) -> int:
    """dataloader/ar_dataloder.py:11-18"""
    return int(dt.strftime("%Y%m%d%H"))


@dataclass
class Fields2DPipelineOutput:
    fields: Union[torch.Tensor, np.ndarray]


# -- samplers -------------------------------------------------------------------
@torch.no_grad()
def edm_AR_sampler(
    net,
    noise_scheduler,
    batch_size=1,
    return_seq_len=1,
    randn_like=torch.randn_like,
    num_inference_steps=18,
    S_churn=0,
    S_min=0,
    S_max=float("inf"),
    S_noise=0,
    deterministic=True,
    known_latents=None,
    timestamps=None,
    generator=None,
    device="cpu",
):
    """EDM Heun sampler, fp64 state (pipelines/edm_sampler.py:10-120), incl. the stochastic-churn branch (:67-76)."""
    if isinstance(generator, list) and len(generator) != batch_size:
        raise ValueError(
            f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
            f" size of {batch_size}. Make sure the batch size matches the length of the generators."
        )
    assert known_latents is not None, "known_latents must be provided"
    device = torch.device(device) if isinstance(device, str) else device
    shape = (batch_size, net.config.out_channels, return_seq_len, *known_latents.shape[-2:])
    latents = randn_tensor(shape, generator=generator, device=device, dtype=net.dtype)
    noise_scheduler.set_timesteps(num_inference_steps, device=device)
    t_steps = noise_scheduler.sigmas.to(device)
    x_next = latents.to(torch.float64) * t_steps[0]
    for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
        x_cur = x_next
        if not deterministic:  # :67-76: push the state back up to t_hat = (1 + gamma) t_cur with fresh noise
            gamma = min(S_churn / num_inference_steps, np.sqrt(2) - 1) if S_min <= t_cur <= S_max else 0
            t_hat = torch.as_tensor(t_cur + gamma * t_cur)
            x_hat = x_cur + (t_hat**2 - t_cur**2).sqrt() * S_noise * randn_like(x_cur)
        else:
            x_hat, t_hat = x_cur, t_cur
        c_noise = noise_scheduler.precondition_noise(t_hat)
        x_in = noise_scheduler.precondition_inputs(x_hat.clone(), t_hat)
        den = net(x_in.to(torch.float32), c_noise.reshape(-1).to(torch.float32), known_latents, time_elapsed=timestamps).sample
        den = noise_scheduler.precondition_outputs(x_hat, den.to(torch.float64), t_hat)
        d_cur = (x_hat - den) / t_hat
        x_next = x_hat + (t_next - t_hat) * d_cur
        if i < num_inference_steps - 1:
            c_noise = noise_scheduler.precondition_noise(t_next)
            x_in = noise_scheduler.precondition_inputs(x_next.clone(), t_next)
            den = net(x_in.to(torch.float32), c_noise.reshape(-1).to(torch.float32), known_latents, time_elapsed=timestamps).sample
            den = noise_scheduler.precondition_outputs(x_next, den.to(torch.float64), t_next)
            d_prime = (x_next - den) / t_next
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next.float()


class AutoRegressive2DPipeline:
    """pipelines/pipeline_AR.py:9-107.  Keeps the reference quirk that the initial
    noise is NOT scaled by ``init_noise_sigma`` (:77-82)."""

    def __init__(self, ar_model, scheduler, scheduler_step_kwargs: Optional[dict] = None):
        self.ar_model = ar_model
        self.scheduler = scheduler
        self.scheduler_step_kwargs = scheduler_step_kwargs or {}

    @property
    def _execution_device(self):
        return self.ar_model.device

    @torch.no_grad()
    def __call__(
        self,
        batch_size: int = 1,
        return_seq_len: int = 1,
        known_latents: torch.Tensor = None,
        timestamps=None,
        generator=None,
        num_inference_steps: int = 50,
        return_dict: bool = True,
        do_edm_style: bool = True,
    ):
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators."
            )
        assert known_latents is not None, "known_latents must be provided"
        shape = (batch_size, self.ar_model.config.out_channels, return_seq_len, *known_latents.shape[-2:])
        image = randn_tensor(shape, generator=generator, device=self._execution_device, dtype=self.ar_model.dtype)
        self.scheduler.set_timesteps(num_inference_steps)
        for t in self.scheduler.timesteps:
            if not do_edm_style:
                raise NotImplementedError("Only EDM style is supported for now")
            x_in = self.scheduler.scale_model_input(image, t)
            t = t.expand(batch_size).to(self._execution_device)
            out = self.ar_model(x_in, t, known_latents, time_elapsed=timestamps, return_dict=False)[0]
            image = self.scheduler.step(out, t, image, **self.scheduler_step_kwargs, return_dict=False)[0]
        if not return_dict:
            return (image,)
        return Fields2DPipelineOutput(fields=image)


@torch.no_grad()
def ensemble_AR_sampler(
    pipeline,
    sample_size: int,
    return_seq_len: int,
    num_inference_steps: int,
    sampler_kwargs=None,
    known_latents: torch.Tensor = None,
    timestamps=None,
    batch_size: int = 64,
    sampler_type: Optional[str] = "edm",
    device="cpu",
    member_ids: Optional[Sequence[int]] = None,
):
    """pipelines/utils.py:664-742.  Member k is always seeded with k (:703-706).
    ``member_ids`` (not in the reference) lets a shard compute a subset of members
    with their global seeds; default = ``range(sample_size)`` = reference behaviour."""
    if member_ids is None:
        member_ids = list(range(sample_size))
    assert len(member_ids) == sample_size
    sizes = [batch_size] * int(sample_size / batch_size) + [sample_size % batch_size]
    samples = torch.empty(
        sample_size, pipeline.ar_model.config.out_channels, return_seq_len, *known_latents.shape[-2:], device=device, dtype=pipeline.ar_model.dtype
    )
    sampler_kwargs = sampler_kwargs or {}
    if sampler_type == "edm":
        model = pipeline.ar_model
        sched = copy.deepcopy(pipeline.scheduler)
    count = 0
    for n in sizes:
        if n == 0:  # reference appends a zero-size chunk when sample_size % batch_size == 0
            continue
        gens = [torch.Generator("cpu").manual_seed(int(member_ids[j]) % (1 << 32)) for j in range(count, count + n)]
        if known_latents.shape[0] == 1:
            kl = known_latents.expand(n, *known_latents.shape[1:])
        else:
            kl = known_latents[count : count + n] if known_latents.shape[0] != n else known_latents
        if sampler_type == "edm":
            out = edm_AR_sampler(
                model, sched, batch_size=n, return_seq_len=return_seq_len, num_inference_steps=num_inference_steps,
                generator=gens, device=device, known_latents=kl, timestamps=timestamps, **sampler_kwargs,
            )
        elif sampler_type == "pipeline":
            out = pipeline(
                batch_size=n, return_seq_len=return_seq_len, num_inference_steps=num_inference_steps, generator=gens,
                known_latents=kl, timestamps=timestamps, return_dict=False, do_edm_style=True, **sampler_kwargs,
            )[0]
        else:
            raise ValueError(sampler_type)
        samples[count : count + n] = out
        count += n
    return samples


@torch.no_grad()
def decode_latent_ens(encdec_model, latents, mean_tensor=None, std_tensor=None, extract_first=None):
    """pipelines/utils.py:51-80"""
    B, _, T, _, _ = latents.shape
    if extract_first is None:
        extract_first = T
    x = latents[:, :, :extract_first].to(encdec_model.device)
    x = x.permute(0, 2, 1, 3, 4).reshape(B * extract_first, *x.shape[1:2], *x.shape[3:])
    y = encdec_model.decode(x).sample
    y = y.reshape(B, extract_first, *y.shape[1:]).permute(0, 2, 1, 3, 4)
    if mean_tensor is not None:
        y = inverse_normalize_transform_3D(y, mean_tensor.to(y.device), std_tensor.to(y.device))
    return y


@torch.no_grad()
def roll_out_serial(
    input_fields: Callable[[datetime], torch.Tensor],
    pred_timestamp: Sequence[datetime],
    pipeline,
    mean_tensor: Optional[torch.Tensor] = None,
    std_tensor: Optional[torch.Tensor] = None,
    ensemble_size: int = 1,
    num_inference_steps: int = 20,
    return_seq_len: int = 8,
    encdec_model=None,
    encdec_model_type: str = "ae",
    static_tensor4encdec: Optional[torch.Tensor] = None,
    latent_transform: Optional[str] = "normalize",
    latent_transform_args: Optional[dict] = None,
    total_lead_time_hour: int = 240,
    step_size_hour: int = 6,
    sampler_type: Optional[str] = "pipeline",
    input_seq_len: int = 1,
    return_latent: bool = False,
    noise_level: float = 0,
    member_ids: Optional[Sequence[int]] = None,
    raw_input_fields: Optional[Callable[[datetime], torch.Tensor]] = None,
    **_ignored,
):
    """Tensor mode (``return_tensor=True``) of pipelines/utils.py:249-661.

    ``input_fields(t)`` returns the *normalised* ``(C, T_in, H, W)`` field tensor the
    reference gets from ``xarr_to_tensor(ds.sel(time=...), mean, std)`` (:457-461).
    Output: ``(n_init, ens, C, 1+steps, h, w)`` fp32 on CPU, NaN-initialised (:413-440);
    slot 0 = un-normalised IC latent (return_latent); in decoded mode slot 0 is the RAW IC field
    ``xarr_to_tensor(ds.sel(time=[t0]))[:, -1]`` (:462-468) = ``raw_input_fields(t0)`` ``(C, H, W)`` when that
    callable is given, NaN otherwise.  Unknown kwargs (``log_pred_interval_hour``,
    evaluate/pred_rollout.py:384) are accepted and ignored."""
    if total_lead_time_hour % step_size_hour != 0:
        raise ValueError("total_lead_time_hour must be divisible by step_size_hour.")
    total = int(total_lead_time_hour / step_size_hour)
    reps = math.ceil(total / return_seq_len)
    fwd = get_transform_3D(latent_transform, latent_transform_args)
    inv = get_inv_transform_3D(latent_transform, latent_transform_args)
    dev = pipeline._execution_device
    out = None
    for pi, t0 in enumerate(pred_timestamp):
        field = input_fields(t0)
        enc = encdec_model.encode(
            field.permute(1, 0, 2, 3).to(encdec_model.device),
            static_conditioning_tensor=static_tensor4encdec.unsqueeze(0).to(encdec_model.device),
        )
        if encdec_model_type != "ae":
            raise ValueError("Unknown encdec_model_type.")
        known = enc.latent.permute(1, 0, 2, 3)
        if out is None:
            if return_latent:
                shape = (len(pred_timestamp), ensemble_size, encdec_model.config.latent_channels, total + 1, *known.shape[-2:])
            else:
                shape = (
                    len(pred_timestamp), ensemble_size, encdec_model.config.out_channels - encdec_model.config.static_channels,
                    total + 1, *field.shape[-2:],
                )
            out = torch.full(shape, float("nan"), dtype=torch.float32, device="cpu")
        if return_latent:
            out[pi, :, :, 0] = known.clone()[:, -1].unsqueeze(0).expand(ensemble_size, -1, -1, -1)
        elif raw_input_fields is not None:
            out[pi, :, :, 0] = raw_input_fields(t0).unsqueeze(0).expand(ensemble_size, -1, -1, -1)
        known = fwd(known)
        if noise_level > 0:
            lstd = torch.tensor(latent_transform_args["std"], dtype=torch.float32).to(known.device)[:, None, None, None]
            known = known + torch.randn_like(known) * noise_level * lstd
        known = known.unsqueeze(0)
        for step in range(reps):
            cur = min(1 + (step + 1) * return_seq_len, total + 1)
            sel = cur - (1 + step * return_seq_len)
            ts = convert_datetime_to_int(t0 + timedelta(hours=step * step_size_hour * return_seq_len))
            ts = torch.tensor([ts], device=dev)
            smp = ensemble_AR_sampler(
                pipeline, sample_size=ensemble_size, return_seq_len=return_seq_len, num_inference_steps=num_inference_steps,
                known_latents=known, timestamps=ts, sampler_type=sampler_type, device=dev, member_ids=member_ids,
            )
            known = smp[:, :, -input_seq_len:].clone()
            B, C, T = smp.shape[:3]
            flat = inv(smp.permute(1, 0, 2, 3, 4).reshape(C, B * T, *smp.shape[3:]))
            smp = flat.reshape(C, B, T, *smp.shape[3:]).permute(1, 0, 2, 3, 4)
            if return_latent:
                out[pi, :, :, 1 + step * return_seq_len : cur] = smp[:, :, :sel].clone().to("cpu")
            else:
                dec = decode_latent_ens(encdec_model, smp[:, :, :sel], mean_tensor=mean_tensor, std_tensor=std_tensor)
                out[pi, :, :, 1 + step * return_seq_len : cur] = dec.clone().to("cpu")
    return out
