"""Host-side (scalar / table) embedding helpers of the AR transformer.

These are tiny, data-independent tables the reference rebuilds on every forward
(models/LaDCast_3D_model.py:885-938, models/embeddings.py:274-327,422-520); here they are
computed once on the host with the same fp32 formulas and cached on the device.
"""
from __future__ import annotations

import math
from datetime import datetime
from typing import Sequence

import torch


def convert_int_to_datetime(ts: int) -> datetime:
    """models/embeddings.py:428-439"""
    s = str(int(ts))
    return datetime(int(s[0:4]), int(s[4:6]), int(s[6:8]), int(s[8:10]))


def compute_year_progress(dt: datetime) -> float:
    """models/embeddings.py:442-447"""
    y0, y1 = datetime(dt.year, 1, 1), datetime(dt.year + 1, 1, 1)
    return (dt - y0).total_seconds() / (y1 - y0).total_seconds()


def get_year_sincos_embedding(timestamps: Sequence[int], embedding_dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    """models/embeddings.py:467-520 -> (B, embedding_dim) fp32 on the host: [sin | cos] halves,
    frequencies 1..half, magnitude exp(-ln(max_period) * (k-1) / half)."""
    prog = torch.tensor([compute_year_progress(convert_int_to_datetime(t)) for t in timestamps], dtype=torch.float32)
    half = embedding_dim // 2
    freqs = torch.arange(1, half + 1).float()
    mag = torch.exp(-math.log(max_period) * torch.arange(0, half).float() / half)
    arg = (2 * math.pi * prog.reshape(-1, 1)) * freqs.reshape(1, -1)
    emb = torch.zeros((len(timestamps), embedding_dim))
    emb[:, :half] = torch.sin(arg) * mag.reshape(1, -1)
    emb[:, half:] = torch.cos(arg) * mag.reshape(1, -1)
    return emb


def rotary_1d(dim: int, pos: torch.Tensor, theta: float):
    """diffusers get_1d_rotary_pos_embed(use_real=True): cos/sin(outer(pos, theta^(-2j/dim))),
    each repeat-interleaved by two -> (S, dim)."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos.to(torch.float32), freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def rope_tables_from_grid(axes_dim: Sequence[int], grids: Sequence[torch.Tensor], theta: float):
    """LaDCastRotaryPosEmbed_from_grid.forward (models/embeddings.py:274-327): token order is the
    ij-meshgrid flattening t*(H*W) + h*W + w, the same as the patch embed's flatten(2)."""
    mesh = torch.stack(torch.meshgrid(*[g.to(torch.float32) for g in grids], indexing="ij"), dim=0)
    cos, sin = zip(*[rotary_1d(d, mesh[i].reshape(-1), theta) for i, d in enumerate(axes_dim)])
    return torch.cat(cos, dim=1).contiguous(), torch.cat(sin, dim=1).contiguous()
