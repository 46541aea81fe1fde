"""Tensor-level counterpart of the reference's rollout driver ``evaluate/pred_rollout.py``: everything
that script does between "inputs are tensors" and "``latent_YYYYMMDDHH.npy`` is on disk" - static-field preparation
(:246-291), latent normalisation arguments (:221-226), the per-initial-time loop with ranks sharing the work
(:347-400) and the saved layout (:420-430).  The zarr / xarray reading of the reference stays with the caller
(``input_fields`` is a callable returning the normalised IC field), as for ``roll_out_serial``.

Work split: the reference takes ``batch_size`` (default: the number of ranks) initial times at a time and deals whole initial
times to ranks (``accelerator.split_between_processes``, :336-358), then gathers result tensors.  Here the same batches are cut
in two dimensions: the ``n_init x ensemble_size`` (initial time, member) items of a batch go to ranks in contiguous, balanced
blocks (``pipelines/distributed.py::shard_work``; results are partition-invariant because member k is always seeded with k) -
the same files come out, 8 GPUs cooperate on one forecast instead of idling when there are fewer initial times than ranks, and
the README's 20 members on 8 ranks balance exactly whenever a batch holds an even number of initial times.
"""
from __future__ import annotations

import json
from datetime import datetime
from typing import Callable, Dict, List, Optional, Sequence, Union

import torch

from ..pipelines.distributed import roll_out_sharded
from ..pipelines.io import save_latent_npy
from ..pipelines.utils import roll_out_serial

SST_CHANNEL = 82  # evaluate/evaluate_ens_gpu.py:50 (84-channel order of evaluate/pred_rollout.py:33-46)


def crop_south_pole(t: torch.Tensor) -> torch.Tensor:
    """drop the first latitude row (-90 deg): (..., 121, 240) -> (..., 120, 240) (pred_rollout.py:253,265,271)"""
    return t[..., 1:, :]


def build_static_conditioning(lsm: Optional[torch.Tensor] = None, orography: Optional[torch.Tensor] = None, crop_pole: bool = True) -> Optional[torch.Tensor]:
    """land-sea mask (lat, lon) first, then the 4 orography fields (4, lat, lon); each channel z-scored over the grid with the
    UNBIASED standard deviation (pred_rollout.py:273-291) -> (C, lat, lon) or None"""
    parts = []
    if lsm is not None:
        lsm = lsm.float()
        parts.append((crop_south_pole(lsm) if crop_pole else lsm).unsqueeze(0))
    if orography is not None:
        orography = orography.float()
        parts.append(crop_south_pole(orography) if crop_pole else orography)
    if not parts:
        return None
    static = torch.cat(parts, dim=0)
    mean = static.mean(dim=(1, 2), keepdim=True)
    std = static.std(dim=(1, 2), keepdim=True)
    return (static - mean) / std


def fill_sst_nan(field: torch.Tensor, sst_channel_idx: int = SST_CHANNEL, value: float = -2.0) -> torch.Tensor:
    """NaNs of the (already normalised) sea-surface-temperature channel - land points - become -2, in place
    (dataloader/utils.py:396-400); field: (C, T, lat, lon)"""
    ch = field[sst_channel_idx]
    ch[torch.isnan(ch)] = value
    return field


def load_latent_transform_args(latent_normal_json: Union[str, Dict], target_std: float = 0.5) -> Dict:
    """the ``{"mean": [...], "std": [...]}`` latent statistics plus ``target_std = 0.5`` (pred_rollout.py:221-226)"""
    if isinstance(latent_normal_json, dict):
        args = dict(latent_normal_json)
    else:
        with open(latent_normal_json) as f:
            args = json.load(f)
    args["target_std"] = target_std
    return args


@torch.no_grad()
def run_rollout(
    input_fields: Callable[[datetime], torch.Tensor],
    init_times: Sequence[datetime],
    pipeline,
    encdec_model,
    latent_transform_args: Dict,
    normalization_param_dict: Optional[Dict] = None,
    static_conditioning_tensor: Optional[torch.Tensor] = None,
    output: Optional[str] = None,
    ensemble_size: int = 1,
    num_inference_steps: int = 20,
    return_seq_len: int = 4,
    input_seq_len: int = 1,
    total_lead_time_hour: int = 240,
    step_size_hour: int = 6,
    sampler_type: str = "edm",
    save_as_latent: bool = True,
    noise_level: float = 0,
    device: Optional[torch.device] = None,
    batch_size: Optional[int] = None,
    ic_noise_seed: int = 0,
    latent_hw: Optional[Sequence[int]] = None,
    **roll_out_kwargs,
) -> List[torch.Tensor]:
    """One ``roll_out_serial`` call per initial time with the reference CLI's fixed arguments (``encdec_model_type="ae"``,
    ``latent_transform="normalize"``, ``return_tensor=True``; pred_rollout.py:367-390), the ensemble shared between the ranks of the
    default process group when there is one.  ``save_as_latent``: rank 0 writes ``<output>/latent_YYYYMMDDHH.npy`` holding
    ``(ens, 84, 1 + steps, 15, 30)`` (:420-430).  Returns the per-initial-time tensors ``(ens, C, 1 + steps, h, w)`` (on every rank).
    ``device``: where the per-rank result block lives for the gather - pass this rank's GPU with the nccl (RCCL) backend.
    ``batch_size``: initial times per sharded batch (default: the number of ranks, as pred_rollout.py:336-338).  Memory of the
    gather: the collective itself moves the per-rank blocks in pieces of at most 256 MiB (``distributed.GATHER_CHUNK_BYTES``); the
    assembled batch - ``batch_size x ens`` items - lives on ``device`` in latent mode (12 MB per member at 240 h) and on the HOST in
    decoded mode (``save_as_latent=False``: 0.8 GB per member at 240 h; 8 initial times x 20 members = 130 GB, never on a GPU).
    ``noise_level > 0`` perturbs each initial time's IC once, from a CPU generator seeded with ``ic_noise_seed + YYYYMMDDHH`` of the
    initial time, so the result does not depend on the number of ranks.  NOTE the default ``ic_noise_seed = 0``: unlike the
    reference (process RNG stream), two runs over the same initial time draw the SAME IC perturbation unless the caller varies
    the seed.  ``latent_hw``: the latent grid ``(h, w)``; default: the static field's grid over the autoencoder's compression
    ratio.  Known up front, it makes the latent-mode exchange the single all_gather of north_star (no shape agreement round)."""
    if total_lead_time_hour % step_size_hour:
        raise ValueError("total_lead_time_hour must be divisible by step_size_hour")  # pipelines/utils.py:305-306
    import torch.distributed as dist

    active = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if active else 0
    if batch_size is None:
        batch_size = dist.get_world_size() if active else 1
    init_times = list(init_times)
    if latent_hw is None and static_conditioning_tensor is not None and hasattr(encdec_model, "spatial_compression_ratio"):
        r_ = int(encdec_model.spatial_compression_ratio)
        latent_hw = tuple(int(v) // r_ for v in static_conditioning_tensor.shape[-2:])
    results = []
    for b0 in range(0, len(init_times), batch_size):
        batch = init_times[b0 : b0 + batch_size]
        full = roll_out_sharded(
            roll_out_serial, ensemble_size=ensemble_size, device=device, out_device=None if save_as_latent else "cpu", input_fields=input_fields, pred_timestamp=batch, pipeline=pipeline,
            normalization_param_dict=normalization_param_dict, num_inference_steps=num_inference_steps, return_seq_len=return_seq_len,
            encdec_model=encdec_model, encdec_model_type="ae", static_tensor4encdec=static_conditioning_tensor, latent_transform="normalize",
            latent_transform_args=latent_transform_args, total_lead_time_hour=total_lead_time_hour, step_size_hour=step_size_hour,
            sampler_type=sampler_type, input_seq_len=input_seq_len, return_tensor=True, return_latent=save_as_latent, noise_level=noise_level,
            ic_noise_seed=ic_noise_seed, latent_hw=latent_hw, **roll_out_kwargs,
        )  # (len(batch), ens, C, 1 + steps, h, w)
        if output is not None and save_as_latent and rank == 0:
            save_latent_npy(full, batch, output)
        results.extend(full[i] for i in range(len(batch)))
    return results
