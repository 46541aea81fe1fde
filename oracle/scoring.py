"""CPU oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py) for the ensemble scoring step, SURVEY.md section 8(f)
rank 1.  A torch restatement of ladcast/evaluate/utils.py:9-149 and of the per-lead-time block of
ladcast/evaluate/evaluate_ens_gpu.py:339-425.

PINNED: tests/golden/scoring_ref.npz holds inputs and outputs of the reference's own function bodies
(tests/golden/make_golden.py executes the definitions found in /root/reference/ladcast/evaluate/utils.py; the module
itself cannot be imported here because it imports xarray at the top, which none of these functions use);
tests/test_oracle_scoring.py checks this restatement against them bit for bit."""
from typing import Optional

import torch


def get_lat_weights_from_lat_tensor(lat: torch.Tensor) -> torch.Tensor:
    """evaluate/utils.py:9-37: WeatherBench2 cell-area weights for latitudes (B, L) in degrees"""
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


def pointwise_crps_skill(forecast: torch.Tensor, truth: torch.Tensor, ensemble_dim: int) -> torch.Tensor:
    """evaluate/utils.py:51-59"""
    return torch.abs(truth - forecast).mean(dim=ensemble_dim)


def pointwise_crps_spread(forecast: torch.Tensor, ensemble_dim: int) -> torch.Tensor:
    """evaluate/utils.py:62-103: 2 / (M (M-1)) * sum_i (2 i - M - 1) x_(i) over the sorted members"""
    n = forecast.shape[ensemble_dim]
    if n < 2:
        return torch.zeros_like(forecast.select(ensemble_dim, 0))
    sorted_forecast, _ = torch.sort(forecast, dim=ensemble_dim)
    weights = 2 * (torch.arange(1, n + 1, device=forecast.device, dtype=forecast.dtype)) - n - 1
    shape = [1] * forecast.ndim
    shape[ensemble_dim] = -1
    weighted_sum = (sorted_forecast * weights.view(*shape)).sum(dim=ensemble_dim)
    return 2 * weighted_sum / (n * (n - 1))


def get_crps(forecast: torch.Tensor, truth: torch.Tensor, ensemble_dim: int = 0) -> torch.Tensor:
    """evaluate/utils.py:106-120"""
    return pointwise_crps_skill(forecast, truth, ensemble_dim) - 0.5 * pointwise_crps_spread(forecast, ensemble_dim)


def get_acc(forecast: torch.Tensor, truth: torch.Tensor, climate: torch.Tensor, lat_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """evaluate/utils.py:123-149"""
    fa = forecast - climate
    ta = truth - climate
    if lat_weight is not None:
        return (fa * ta * lat_weight).nanmean(dim=(-2, -1)) / torch.sqrt(
            (fa**2 * lat_weight).nanmean(dim=(-2, -1)) * (ta**2 * lat_weight).nanmean(dim=(-2, -1)))
    return (fa * ta).nanmean(dim=(-2, -1)) / torch.sqrt((fa**2).nanmean(dim=(-2, -1)) * (ta**2).nanmean(dim=(-2, -1)))


def ensemble_scores(dec_t: torch.Tensor, ref_t: torch.Tensor, clim_t: torch.Tensor, lat_weight: torch.Tensor, sst_channel: int):
    """the per-lead-time block of evaluate/evaluate_ens_gpu.py:339-425: dec_t (ens, C, H, W), ref_t / clim_t (C, H, W),
    lat_weight (H,) -> dict of (C,) tensors.  Channel `sst_channel` is averaged with nanmean, the others with mean."""
    weights = lat_weight.view(1, -1, 1)
    mean_t = dec_t.mean(dim=0)
    acc = get_acc(mean_t, ref_t, clim_t, weights)
    se_t = (mean_t - ref_t) ** 2 * weights
    spread_t = pointwise_crps_spread(dec_t, ensemble_dim=0) * weights
    skill_t = pointwise_crps_skill(dec_t, ref_t.unsqueeze(0), 0) * weights
    crps_t = skill_t - 0.5 * spread_t

    def split_mean(x):
        out = torch.empty(x.shape[0], dtype=x.dtype)
        s = sst_channel
        out[:s] = x[:s].mean(dim=(1, 2))
        out[s : s + 1] = torch.nanmean(x[s : s + 1], dim=(1, 2))
        out[s + 1 :] = x[s + 1 :].mean(dim=(1, 2))
        return out

    return dict(ens_acc=acc, ens_mse=split_mean(se_t), crps_spread=split_mean(spread_t), crps_skill=split_mean(skill_t), crps=split_mean(crps_t))
