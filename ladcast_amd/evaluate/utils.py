"""Ensemble scoring on the device (reference: ladcast/evaluate/utils.py:9-149 and the per-lead-time block of
ladcast/evaluate/evaluate_ens_gpu.py:339-425).  Same function names and argument meaning; the forecast is read once by
one fused HIP kernel (`ldc_ensemble_scores`) instead of ~40 torch ops.  Tensors must live on a HIP device: there is no
CPU path."""
from typing import Dict, Optional

import torch

from .. import hip


def get_lat_weights_from_lat_tensor(lat: torch.Tensor) -> torch.Tensor:
    """evaluate/utils.py:9-37 (host-side table, L values)"""
    lat_rad = torch.deg2rad(lat)
    midpoints = (lat_rad[:, :-1] + lat_rad[:, 1:]) / 2
    B = lat_rad.shape[0]
    lower = torch.full((B, 1), -torch.pi / 2, dtype=lat_rad.dtype, device=lat_rad.device)
    upper = torch.full((B, 1), torch.pi / 2, dtype=lat_rad.dtype, device=lat_rad.device)
    bounds = torch.cat([lower, midpoints, upper], dim=1)
    cell_area = torch.sin(bounds[:, 1:]) - torch.sin(bounds[:, :-1])
    return cell_area / cell_area.mean(dim=1, keepdim=True)


def get_normalized_lat_weights_based_on_cos(lat: torch.Tensor) -> torch.Tensor:
    """evaluate/utils.py:40-48"""
    weights = torch.cos(torch.deg2rad(lat))
    return weights / weights.mean()


def _as_echw(forecast: torch.Tensor, ensemble_dim: int) -> torch.Tensor:
    if forecast.dim() != 4:
        raise ValueError("forecast must be (ens, C, H, W) up to the position of the ensemble axis")
    f = forecast.movedim(ensemble_dim, 0)
    if f.dtype != torch.float32:
        raise NotImplementedError("fp32 only")
    if f.stride(-1) != 1 or f.stride(-2) != f.shape[-1]:
        f = f.contiguous()
    return f


def _scores(forecast, truth, clim, lat_weight, nan_channel, want_maps):
    f = _as_echw(forecast, 0)
    M, C, H, W = f.shape
    dev = f.device
    t = truth.to(dev, torch.float32).expand(C, H, W)
    t = t if (t.stride(-1) == 1 and t.stride(-2) == W) else t.contiguous()
    c = None
    if clim is not None:
        c = clim.to(dev, torch.float32).expand(C, H, W)
        c = c if (c.stride(-1) == 1 and c.stride(-2) == W) else c.contiguous()
    w = (torch.ones(H, device=dev) if lat_weight is None else lat_weight.to(dev, torch.float32).reshape(-1)).contiguous()
    if w.numel() != H:
        raise ValueError("lat_weight must have one value per latitude row")
    out = torch.empty(5, C, device=dev, dtype=torch.float32)
    skill = torch.empty(C, H, W, device=dev) if want_maps else None
    spread = torch.empty(C, H, W, device=dev) if want_maps else None
    hip.ensemble_scores(f, t, c, w, out, M=M, C=C, H=H, W=W, member_stride=f.stride(0), channel_stride=f.stride(1),
                        truth_channel_stride=t.stride(0), clim_channel_stride=0 if c is None else c.stride(0), nan_channel=nan_channel,
                        skill_map=skill, spread_map=spread)
    return out, skill, spread


@torch.no_grad()
def pointwise_crps_skill(forecast: torch.Tensor, truth: torch.Tensor, ensemble_dim: int) -> torch.Tensor:
    """evaluate/utils.py:51-59 for a (ens, C, H, W) forecast (ensemble axis anywhere) -> (C, H, W)"""
    f = _as_echw(forecast, ensemble_dim)
    t = truth
    if t.dim() == 4:
        t = t.movedim(ensemble_dim, 0)[0]
    return _scores(f, t, None, None, -1, True)[1]


@torch.no_grad()
def pointwise_crps_spread(forecast: torch.Tensor, ensemble_dim: int) -> torch.Tensor:
    """evaluate/utils.py:62-103 -> (C, H, W)"""
    f = _as_echw(forecast, ensemble_dim)
    zeros = torch.zeros(f.shape[1:], device=f.device)
    return _scores(f, zeros, None, None, -1, True)[2]


@torch.no_grad()
def get_crps(forecast: torch.Tensor, truth: torch.Tensor, ensemble_dim: int = 0) -> torch.Tensor:
    """evaluate/utils.py:106-120 -> (C, H, W)"""
    f = _as_echw(forecast, ensemble_dim)
    t = truth
    if t.dim() == 4:
        t = t.movedim(ensemble_dim, 0)[0]
    _, skill, spread = _scores(f, t, None, None, -1, True)
    return skill - 0.5 * spread


@torch.no_grad()
def get_acc(forecast: torch.Tensor, truth: torch.Tensor, climate: torch.Tensor, lat_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """evaluate/utils.py:123-149 for (C, H, W) fields and lat_weight broadcastable as (1, H, 1) -> (C,)"""
    out, _, _ = _scores(forecast.unsqueeze(0), truth, climate, None if lat_weight is None else lat_weight.reshape(-1), -1, False)
    return out[0]


@torch.no_grad()
def ensemble_scores(dec_t: torch.Tensor, ref_t: torch.Tensor, clim_t: torch.Tensor, lat_weight: torch.Tensor, sst_channel: int) -> Dict[str, torch.Tensor]:
    """One lead time of evaluate/evaluate_ens_gpu.py:339-425: dec_t (ens, C, H, W) (any member / channel strides, e.g. a
    `[:, :, t]` view of the (ens, C, T, H, W) array), ref_t / clim_t (C, H, W), lat_weight (H,) -> (C,) device tensors
    ens_acc, ens_mse, crps_spread, crps_skill, crps; channel `sst_channel` is averaged with nanmean, the others with mean."""
    out, _, _ = _scores(dec_t, ref_t, clim_t, lat_weight, sst_channel, False)
    return dict(ens_acc=out[0], ens_mse=out[1], crps_spread=out[2], crps_skill=out[3], crps=out[4])
