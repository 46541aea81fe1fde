"""On-disk conventions of the rollout output, so that files written here are readable
by the reference's tooling and vice versa:

* writer: `ladcast/evaluate/pred_rollout.py:421-430` saves one file per initial time, `latent_YYYYMMDDHH.npy`, holding
  `result[i]` = `(ens, 84, 1 + steps, 15, 30)` fp32 with slot 0 = the un-normalised IC latent;
* readers: `ladcast/evaluate/evaluate_ens_gpu.py:208-283` globs `latent_*.npy`, sorts the paths, takes the timestamp
  from the file name, drops initial times later than `end_date - total_lead_time_hour`, optionally crops slot 0
  (`--crop_init`) and the ensemble (`--force_ens_size`); `ladcast/pipelines/utils.py:129-137` also accepts a 6-D
  `(1, ens, C, T, h, w)` array and uses its first entry.
Plain host-side numpy: nothing here is a hot path."""
import glob
import os
from datetime import datetime, timedelta
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ..models.embeddings import convert_int_to_datetime
from .utils import convert_datetime_to_int


def latent_file_name(timestamp) -> str:
    """`latent_{convert_datetime_to_int(t)}.npy` (pred_rollout.py:425-428); accepts datetime, np.datetime64 or YYYYMMDDHH int"""
    if isinstance(timestamp, (int, np.integer)):
        ts = int(timestamp)
    else:
        ts = convert_datetime_to_int(timestamp)
    if len(str(ts)) != 10:
        raise ValueError(f"timestamp {ts} is not YYYYMMDDHH")
    return f"latent_{ts}.npy"


def save_latent_npy(result: torch.Tensor, timestamps: Sequence, output_dir: str) -> List[str]:
    """result: (n_init, ens, C, 1 + steps, h, w) as returned by `roll_out_serial(..., return_latent=True)`; one file per
    initial time (pred_rollout.py:420-430).  Returns the paths written."""
    if result.dim() != 6:
        raise ValueError("result must be (n_init, ens, C, 1 + steps, h, w)")
    if result.shape[0] != len(timestamps):
        raise ValueError(f"{result.shape[0]} initial times in the tensor, {len(timestamps)} timestamps")
    os.makedirs(output_dir, exist_ok=True)
    paths = []
    for i, t in enumerate(timestamps):
        path = os.path.join(output_dir, latent_file_name(t))
        np.save(path, result[i].detach().cpu().numpy())
        paths.append(path)
    return paths


def list_latent_files(result_path: str, end_date: Optional[Union[str, datetime]] = None, total_lead_time_hour: int = 0) -> List[Tuple[str, str]]:
    """[(YYYYMMDDHH string, path)] sorted as the reference sorts them (evaluate_ens_gpu.py:208-220); initial times
    after `end_date - total_lead_time_hour` are dropped when `end_date` is given"""
    paths = sorted(glob.glob(os.path.join(result_path, "latent_*.npy")))
    out = []
    limit = None
    if end_date is not None:
        end = end_date if isinstance(end_date, datetime) else datetime.fromisoformat(str(end_date))
        limit = end - timedelta(hours=int(total_lead_time_hour))
    for path in paths:
        time_str = path.split("/")[-1].split("_")[-1].split(".")[0]
        if limit is None or convert_int_to_datetime(int(time_str)) <= limit:
            out.append((time_str, path))
    return out


def load_latent_npy(path: str, crop_init: bool = False, force_ens_size: Optional[int] = None, device=None) -> Tuple[torch.Tensor, int]:
    """-> ((ens, C, T, h, w) fp32 tensor, YYYYMMDDHH int).  A 6-D array is reduced to its first entry
    (pipelines/utils.py:129-137); `crop_init` drops slot 0, `force_ens_size` keeps the first members
    (evaluate_ens_gpu.py:273-283)."""
    arr = np.load(path)
    if arr.ndim == 6:
        arr = arr[0]
    if arr.ndim != 5:
        raise ValueError(f"{path}: expected (ens, C, T, h, w), got shape {arr.shape}")
    timestamp = int(path.split("/")[-1].split("_")[-1].split(".")[0])
    t = torch.from_numpy(arr)
    if crop_init:
        t = t[:, :, 1:, ...]
    if force_ens_size is not None:
        t = t[:force_ens_size, ...]
    if device is not None:
        t = t.to(device)
    return t, timestamp
