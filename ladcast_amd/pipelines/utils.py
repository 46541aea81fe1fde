"""Rollout driver (pipelines/utils.py:26-80,249-742) in tensor mode.

``roll_out_serial`` keeps the reference's argument names; the xarray dataset is replaced by
``input_fields``: a callable returning the *normalised* ``(C, T_in, H, W)`` field tensor the
reference obtains from ``xarr_to_tensor(ds.sel(time=...), mean, std)`` (:457-461).  xarray
packing (``return_tensor=False``) is out of scope.

Multi-GPU: ensemble members are independent given the IC latent and member k is always
seeded with k (:703-706), so ``member_ids`` lets each rank run a subset with the SAME noise it
would have had in the single-process run -- the result does not depend on the partition.
"""
from __future__ import annotations

import copy
import functools
import math
from dataclasses import dataclass
from datetime import datetime, timedelta
from typing import Callable, Dict, Optional, Sequence, Union

import numpy as np
import torch

from .. import hip
from .edm_sampler import edm_AR_sampler
from .torch_utils import few_host_threads


@dataclass
class Fields2DPipelineOutput:
    """Output class for field pipelines (pipelines/utils.py:26-35)."""

    fields: Union[torch.Tensor, np.ndarray]

    def __getitem__(self, i):
        return (self.fields,)[i]


def convert_datetime_to_int(dt) -> int:
    """dataloader/ar_dataloder.py:11-18"""
    if isinstance(dt, np.datetime64):
        dt = dt.astype("datetime64[h]").tolist()
    return int(dt.strftime("%Y%m%d%H"))


# -- per-channel latent / field transforms (dataloader/utils.py:223-269) on the device ----------
_host_to_device = hip.upload_nonblocking


# results of a rollout are collected on the GPU and copied to the host ONCE at the end while they fit this many bytes: a per-chunk
# `.to("cpu")` stalls the host on every chunk, and the GPU then idles through the host's preparation of the next one (~1 ms of 111)
DEVICE_OUTPUT_MAX_BYTES = 8 << 30


_vector_cache: Dict = {}


def _device_vector(v, dev):
    """per-channel statistics as an fp32 device vector.  Host values (the lists of the reference's JSON) are uploaded once per
    (device, values) and kept: a fresh pageable upload per call is a synchronous copy, i.e. a host stall on the previous chunk."""
    if isinstance(v, torch.Tensor) and v.device.type != "cpu":
        return v.to(dev, torch.float32)
    host = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
    key = (str(dev), host.numpy().tobytes())
    hit = _vector_cache.get(key)
    if hit is None:
        if len(_vector_cache) >= 64:
            _vector_cache.clear()
        hit = _vector_cache[key] = host.to(dev)
    return hit


def _chan_affine(x, mean, std, target_std, inverse):
    """x: (C, T, H, W) or (B, C, T, H, W) contiguous fp32 device tensor."""
    x = x.contiguous()
    if x.numel() == 0:  # a rank that owns no members: nothing to launch
        return torch.empty_like(x)
    dev = x.device
    mean, std = _device_vector(mean, dev), _device_vector(std, dev)
    C = mean.numel()
    cdim = 0 if x.dim() == 4 else 1
    assert x.shape[cdim] == C
    outer = 1 if cdim == 0 else x.shape[0]
    inner = x.numel() // (outer * C)
    y = torch.empty_like(x)
    hip.chan_affine(x, y, mean, std, float(target_std), outer=outer, C=C, inner=inner, inverse=inverse)
    return y


def normalize_transform_3D(sample, mean, std, target_std=1):
    return _chan_affine(sample, mean, std, target_std, inverse=False)


def inverse_normalize_transform_3D(normalized_sample, mean, std, target_std=1):
    return _chan_affine(normalized_sample, mean, std, target_std, inverse=True)


def get_transform_3D(transform, transform_args):
    if transform == "normalize":
        ts = transform_args.get("target_std", 1)
        return lambda x: normalize_transform_3D(x, transform_args["mean"], transform_args["std"], ts)
    if transform is None:
        return lambda x: x
    raise NotImplementedError(f"Transform: {transform} not implemented.")


def get_inv_transform_3D(transform, transform_args):
    if transform == "normalize":
        ts = transform_args.get("target_std", 1)
        return lambda x: inverse_normalize_transform_3D(x, transform_args["mean"], transform_args["std"], ts)
    if transform is None:
        return lambda x: x
    raise NotImplementedError(f"Transform: {transform} not implemented.")


def _host_glue(fn):
    """run the host side of an entry point on a few intra-op threads (torch_utils.few_host_threads: the tensors are tiny, the default
    128-thread pool only adds wake-up stalls)"""

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with few_host_threads():
            return fn(*args, **kwargs)

    return wrapper


@torch.no_grad()
def decode_latent_ens(encdec_model, latents, mean_tensor=None, std_tensor=None, extract_first=None):
    """latents (B, C, T, H, W) -> decoded fields (B, C', T', H', W') (pipelines/utils.py:51-80)."""
    B, _, T, _, _ = latents.shape
    if extract_first is None:
        extract_first = T
    x = latents[:, :, :extract_first].to(encdec_model.device)
    x = x.permute(0, 2, 1, 3, 4).reshape(B * extract_first, x.shape[1], *x.shape[3:]).contiguous()
    y = encdec_model.decode(x).sample
    y = y.reshape(B, extract_first, *y.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()
    if mean_tensor is not None:
        y = inverse_normalize_transform_3D(y, mean_tensor, std_tensor)
    return y


@_host_glue
@torch.no_grad()
def ensemble_AR_sampler(
    pipeline,
    sample_size: int,
    return_seq_len: int,
    num_inference_steps: int,
    sampler_kwargs=None,
    known_latents: torch.Tensor = None,
    timestamps: Optional[torch.LongTensor] = None,
    batch_size: int = 64,
    sampler_type: Optional[str] = "edm",
    device="cpu",
    member_ids: Optional[Sequence[int]] = None,
):
    """pipelines/utils.py:664-742.  ``member_ids`` (extension) = global member index of each of the
    ``sample_size`` local members; default ``range(sample_size)`` is the reference behaviour."""
    if member_ids is None:
        member_ids = list(range(sample_size))
    if len(member_ids) != sample_size:
        raise ValueError("member_ids must list one global member id per local sample")
    sizes = [batch_size] * int(sample_size / batch_size) + [sample_size % batch_size]
    samples = torch.empty(
        sample_size, pipeline.ar_model.config.out_channels, return_seq_len, *known_latents.shape[-2:],
        device=device, dtype=torch.float32,  # (pipelines/utils.py:690-696 asks for ar_model.dtype: this build's models compute and return fp32 - a model
    )                                        # cast to bf16 / fp16 is up-cast at its first use, AFTER this buffer exists; see pipeline_AR.py)
    sampler_kwargs = sampler_kwargs or {}
    if sampler_type == "edm":
        model = pipeline.ar_model
        noise_scheduler = copy.deepcopy(pipeline.scheduler)
    elif sampler_type != "pipeline":
        raise ValueError(f"unknown sampler_type {sampler_type!r}")
    count = 0
    for n in sizes:
        if n == 0:  # the reference appends an empty chunk when sample_size % batch_size == 0 (Q5)
            continue
        generator = [torch.Generator("cpu").manual_seed(int(member_ids[j]) % (1 << 32)) for j in range(count, count + n)]
        if known_latents.shape[0] == 1:
            kl = known_latents.expand(n, *known_latents.shape[1:])
        elif known_latents.shape[0] == n:
            kl = known_latents
        else:
            kl = known_latents[count : count + n]
        if sampler_type == "edm":
            out = edm_AR_sampler(
                model, noise_scheduler, batch_size=n, return_seq_len=return_seq_len, num_inference_steps=num_inference_steps,
                generator=generator, device=device, known_latents=kl, timestamps=timestamps, **sampler_kwargs,
            )
        else:
            out = pipeline(
                batch_size=n, return_seq_len=return_seq_len, num_inference_steps=num_inference_steps, generator=generator,
                known_latents=kl, timestamps=timestamps, return_dict=False, do_edm_style=True, **sampler_kwargs,
            )[0]
        samples[count : count + n] = out
        count += n
    return samples


@_host_glue
@torch.no_grad()
def roll_out_serial(
    input_fields: Callable[[datetime], torch.Tensor],
    pred_timestamp: Sequence[datetime],
    pipeline,
    normalization_param_dict: Optional[Dict] = None,
    ensemble_size: int = 1,
    num_inference_steps: int = 20,
    return_seq_len: int = 8,
    return_ensemble_mean: bool = False,
    encdec_model=None,
    encdec_model_type: str = "vae",
    static_tensor4encdec: Optional[torch.Tensor] = None,
    latent_transform: Optional[str] = "normalize",
    latent_transform_args: Optional[Dict] = None,
    total_lead_time_hour: int = 240,
    step_size_hour: int = 6,
    dataset_interval_hour: int = 1,
    sampler_type: Optional[str] = "pipeline",
    generator=None,
    input_seq_len: int = 1,
    return_tensor: bool = True,
    return_latent: bool = False,
    noise_level: Optional[float] = 0,
    member_ids: Optional[Sequence[int]] = None,
    known_latents_override: Optional[torch.Tensor] = None,
    raw_input_fields: Optional[Callable[[datetime], torch.Tensor]] = None,
    ic_noise_seed: Optional[int] = None,
    output_device=None,
    decode_batch_frames: Optional[int] = None,
    **_ignored,  # e.g. log_pred_interval_hour, which the reference CLI passes (evaluate/pred_rollout.py:384, Q2)
) -> torch.Tensor:
    """Tensor mode of pipelines/utils.py:249-661.

    ``normalization_param_dict``: ``{"mean": (C,), "std": (C,)}`` field statistics (the reference derives
    them from its JSON via ``precompute_mean_std``, :315-317); only used to de-normalise decoded fields.
    Returns ``(n_init, ens, C, 1+steps, h, w)`` fp32 on the host, NaN-initialised (:413-440); slot 0 is
    the un-normalised IC latent when ``return_latent`` (:486-492), otherwise the RAW (un-normalised) IC
    field at t0 as the reference writes it (:462-468): ``raw_input_fields(t0)`` -> ``(C, H, W)`` or
    ``(C, T, H, W)`` (last frame used) when given - the reference's un-normalised dataset read - else the
    normalised field de-normalised with ``normalization_param_dict`` (equal up to one fp32 rounding), else
    NaN.  ``input_fields(t0)`` must return all ``input_seq_len`` frames ``(C, input_seq_len, H, W)``, oldest
    first (:452-461).  ``known_latents_override`` (extension for latent-only benchmarks): a
    ``(C, T_in, h, w)`` un-normalised IC latent used instead of encoding.  ``ensemble_size == 0`` (a rank
    without members) returns the ``(n_init, 0, ...)``-shaped tensor without launching kernels when the
    shape is known without encoding (``known_latents_override``), and is otherwise handled by the caller
    (``pipelines/distributed.py``).  ``noise_level > 0`` perturbs the IC latent ONCE per initial time, shared by all members
    (:518-528).  ``ic_noise_seed=None`` draws that perturbation from torch's global RNG as the reference does; an integer draws it
    from a CPU generator seeded with ``ic_noise_seed + YYYYMMDDHH`` of the initial time, so that every call that touches this initial
    time - e.g. the pieces of one forecast cut over several ranks (``pipelines/distributed.py``) - perturbs it identically.
    ``output_device`` (extension; default: the host, as the reference): where the result tensor is returned.  A CUDA device returns it
    WITHOUT synchronising - the caller's next call then prepares its inputs while this one still runs on the GPU (``bench.py``, and the
    sharded driver, whose gather runs on the device anyway).
    ``decode_batch_frames`` (extension, decoded mode; default None = decode every chunk right after it, as :580-585): keep the chunks' latents in
    HBM (151 KB per member-chunk) and decode them AFTER the last chunk of an initial time in batches of up to that many frames - the decoder runs
    at its large-batch rate (2.4 instead of 3.3 ms per frame on one MI355X at 32 frames).  The same decoder on the same latents: values agree
    with the per-chunk order to fp32 rounding in the fp32 / bf16x3 modes (the conv kernels pick their schedule from the launch size), not bit for
    bit; in the single-term bf16 mode, where every activation is rounded to bf16, at that mode's own error level (tests/test_gpu_cfg5.py).
    """
    if not return_tensor:
        raise NotImplementedError("xarray output is out of scope; use return_tensor=True")
    if return_ensemble_mean and return_latent:
        raise ValueError("return_ensemble_mean must be False when return_latent is True.")
    if total_lead_time_hour % step_size_hour != 0:
        raise ValueError("total_lead_time_hour must be divisible by step_size_hour.")
    assert step_size_hour % dataset_interval_hour == 0, "step_size_hour must be divisible by dataset_interval_hour."
    total = int(total_lead_time_hour / step_size_hour)
    reps = math.ceil(total / return_seq_len)
    fwd = get_transform_3D(latent_transform, latent_transform_args)
    inv = get_inv_transform_3D(latent_transform, latent_transform_args)
    dev = pipeline._execution_device
    return_size = 1 if return_ensemble_mean else ensemble_size
    mean_tensor = std_tensor = None
    if normalization_param_dict is not None:
        mean_tensor, std_tensor = normalization_param_dict["mean"], normalization_param_dict["std"]
    out = None
    for pi, t0 in enumerate(pred_timestamp):
        if known_latents_override is not None:
            known = known_latents_override.to(dev, torch.float32)
            field_hw = None
        else:
            field = input_fields(t0)  # (C, T_in, H, W) normalised
            if field.dim() != 4 or field.shape[1] != input_seq_len:
                raise ValueError(f"input_fields must return (C, input_seq_len={input_seq_len}, H, W), got {tuple(field.shape)}")
            enc = encdec_model.encode(
                field.permute(1, 0, 2, 3).to(encdec_model.device),
                static_conditioning_tensor=static_tensor4encdec.unsqueeze(0).to(encdec_model.device),
            )
            if encdec_model_type != "ae":
                raise ValueError("Unknown encdec_model_type.")
            known = enc.latent.permute(1, 0, 2, 3)  # (C, T_in, h, w)
            field_hw = field.shape[-2:]
        if out is None:
            if return_latent:
                shape = (len(pred_timestamp), return_size, known.shape[0], total + 1, *known.shape[-2:])
            else:
                c_out = encdec_model.config.out_channels - encdec_model.config.static_channels
                hw = field_hw if field_hw is not None else tuple(s * encdec_model.spatial_compression_ratio for s in known.shape[-2:])
                shape = (len(pred_timestamp), return_size, c_out, total + 1, *hw)
            # collected where the sampler runs (one copy to the host at the end) unless that would hold too much device memory
            nbytes = 4 * math.prod(shape)
            acc = torch.device(dev) if (torch.device(dev).type == "cuda" and nbytes <= DEVICE_OUTPUT_MAX_BYTES) else torch.device("cpu")
            out = torch.full(shape, float("nan"), dtype=torch.float32, device=acc)
        if return_latent:
            out[pi, :, :, 0] = known[:, -1].to(acc).unsqueeze(0).expand(return_size, -1, -1, -1)
        elif field_hw is not None:  # raw IC field in slot 0 (:462-468)
            raw = None
            if raw_input_fields is not None:
                raw = torch.as_tensor(raw_input_fields(t0), dtype=torch.float32).to("cpu")
                raw = raw[:, -1] if raw.dim() == 4 else raw
            elif mean_tensor is not None:
                m_ = torch.as_tensor(mean_tensor, dtype=torch.float32).to("cpu")[:, None, None]
                s_ = torch.as_tensor(std_tensor, dtype=torch.float32).to("cpu")[:, None, None]
                raw = field[:, -1].to("cpu", torch.float32) * s_ + m_
            if raw is not None:
                out[pi, :, :, 0] = _host_to_device(raw, acc).unsqueeze(0).expand(return_size, -1, -1, -1)
        if ensemble_size == 0:
            continue
        known = fwd(known.contiguous())
        if noise_level and noise_level > 0:
            lstd = torch.tensor(latent_transform_args["std"], dtype=torch.float32, device=known.device)[:, None, None, None]
            if ic_noise_seed is None:
                eps = torch.randn_like(known)
            else:
                gen = torch.Generator("cpu").manual_seed((int(ic_noise_seed) + convert_datetime_to_int(t0)) % (1 << 62))
                eps = torch.randn(tuple(known.shape), generator=gen, dtype=torch.float32).to(known.device)
            known = known + eps * noise_level * lstd
        known = known.unsqueeze(0)
        pending = []
        for step in range(reps):
            cur = min(1 + (step + 1) * return_seq_len, total + 1)
            sel = cur - (1 + step * return_seq_len)
            ts_host = convert_datetime_to_int(t0 + timedelta(hours=step * step_size_hour * return_seq_len))
            ts = _host_to_device(torch.tensor([ts_host]), dev)
            ts.host_values = [ts_host]  # the model needs the value on the host: spares it a device read-back (= a stall on the previous chunk)
            smp = ensemble_AR_sampler(
                pipeline, sample_size=ensemble_size, return_seq_len=return_seq_len, num_inference_steps=num_inference_steps,
                known_latents=known, timestamps=ts, sampler_type=sampler_type, device=dev, member_ids=member_ids,
            )
            known = smp[:, :, -input_seq_len:].contiguous()
            smp = inv(smp)  # per-channel inverse on (B, C, T, h, w): same values as the reference's rearranged call (:566-574)
            if return_latent:
                out[pi, :, :, 1 + step * return_seq_len : cur] = smp[:, :, :sel].to(acc)
            elif decode_batch_frames:
                pending.append((1 + step * return_seq_len, smp[:, :, :sel]))  # decoded after the last chunk, in large batches
            else:
                dec = decode_latent_ens(encdec_model, smp[:, :, :sel], mean_tensor=mean_tensor, std_tensor=std_tensor)
                if return_ensemble_mean:
                    out[pi, 0, :, 1 + step * return_seq_len : cur] = dec.mean(dim=0).to(acc)
                else:
                    out[pi, :, :, 1 + step * return_seq_len : cur] = dec.to(acc)
        if pending:
            lat = torch.cat([p[1] for p in pending], dim=2)  # (members, C, total, h, w): every lead step of this initial time
            per = max(1, int(decode_batch_frames) // max(lat.shape[0], 1))  # lead steps per decoder call: all members of a lead step stay together
            for s0 in range(0, lat.shape[2], per):
                dec = decode_latent_ens(encdec_model, lat[:, :, s0 : s0 + per], mean_tensor=mean_tensor, std_tensor=std_tensor)
                if return_ensemble_mean:
                    out[pi, 0, :, 1 + s0 : 1 + s0 + dec.shape[2]] = dec.mean(dim=0).to(acc)
                else:
                    out[pi, :, :, 1 + s0 : 1 + s0 + dec.shape[2]] = dec.to(acc)
    if out is None:
        return out
    return out.to(torch.device(output_device) if output_device is not None else torch.device("cpu"))
