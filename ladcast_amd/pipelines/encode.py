"""Bulk encoder path: a stream of normalised frames -> the latent store `(N, 84, 15, 30)`.

Reference: `ladcast/preprocecss/encode_data.py:20-100` (`encode_latents_and_save_zarr_direct`) encodes ONE frame per
`vae.encode` call (`input (1, C, H, W)`, `static_conditioning_tensor (1, S, H, W)`) into a NaN-prefilled
`(total_samples, 84, 15, 30)` fp32 host tensor.  Same result here, but the frames go through the HIP DCAE encoder in
batches (the convs reach ~2x the single-frame rate at 32 frames, profiles/r01_h_dcae_encode_decode.log); the xarray /
zarr side of the reference (reading ERA5, writing the store) stays out of scope, so frames come from a tensor or a
callable and the latents are returned."""
from typing import Callable, Optional, Union

import torch


@torch.no_grad()
def encode_latents(vae, frames: Union[torch.Tensor, Callable[[int], torch.Tensor]], total_samples: Optional[int] = None,
                   static_conditioning_tensor: Optional[torch.Tensor] = None, batch_size: int = 32) -> torch.Tensor:
    """frames: `(N, C, H, W)` normalised fields, or `frames(idx) -> (C, H, W)` for idx in [0, total_samples)
    (what `xarr_to_tensor(ds.sel(time=[t]), ...)` yields in the reference); static_conditioning_tensor `(S, H, W)`.
    Returns the `(N, latent_channels, h, w)` fp32 host tensor of `encode_data.py:47-80`."""
    if callable(frames):
        if total_samples is None:
            raise ValueError("total_samples is required when frames is a callable")
        n = int(total_samples)
        get = frames
    else:
        if frames.dim() != 4:
            raise ValueError("frames must be (N, C, H, W)")
        n = frames.shape[0]
        get = lambda i: frames[i]  # noqa: E731
    if batch_size < 1:
        raise ValueError("batch_size must be positive")
    static = None
    if static_conditioning_tensor is not None:
        static = static_conditioning_tensor.to(vae.device).unsqueeze(0)  # (1, S, H, W), broadcast over the batch by encode()
    out = None
    for i0 in range(0, n, batch_size):
        idx = range(i0, min(n, i0 + batch_size))
        x = torch.stack([get(i) for i in idx], dim=0).to(vae.device)
        latent = vae.encode(x, static_conditioning_tensor=static).latent
        if out is None:
            out = torch.full((n,) + tuple(latent.shape[1:]), float("nan"), dtype=torch.float32, device="cpu")
        out[i0 : i0 + len(idx)] = latent.detach().cpu()
    if out is None:
        out = torch.empty(0)
    return out
