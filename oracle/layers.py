"""Oracle restatement of the diffusers==0.32.1 layer library used by the path.

PARITY UNPINNED (see ``oracle/__init__.py``).  Each class cites the reference
call site that fixes which options are in use; attribute names match diffusers
so that a reference checkpoint's state-dict keys load strictly.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSNorm(nn.Module):
    """``x * rsqrt(mean(x^2) + eps) * weight (+ bias)`` with fp32 statistics.

    Used as q/k norm ``RMSNorm(128, eps=1e-7)`` (models/LaDCast_3D_model.py:258-268,
    408-419,488-500) and over channels in the DCAE (models/DCAE.py:145,300-302,349,701).
    """

    def __init__(self, dim: int, eps: float, elementwise_affine: bool = True, bias: bool = False):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None
        self.bias = nn.Parameter(torch.zeros(dim)) if (elementwise_affine and bias) else None

    def forward(self, x):
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.eps)
        if self.weight is not None:
            x = x * self.weight
            if self.bias is not None:
                x = x + self.bias
        return x


def get_timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0):
    """``Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0)``:
    ``[cos(t f_k) | sin(t f_k)]``, ``f_k = exp(-ln(1e4) k / 128)``
    (models/LaDCast_3D_model.py:362-364,673)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device)
    freqs = torch.exp(exponent / half)
    arg = t[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class TimestepEmbedding(nn.Module):
    """``linear_2(SiLU(linear_1(x)))`` (models/LaDCast_3D_model.py:676-678)."""

    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features: int, hidden_size: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepTextProjEmbeddings(nn.Module):
    """``timestep_embedder(sinusoid(t)) + text_embedder(pooled)``
    (models/LaDCast_3D_model.py:362-364,384,673,954-956).  A ``(1,)`` timestep
    broadcasts against a ``(B, D)`` pooled projection."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim)

    def forward(self, timestep, pooled):
        t_emb = self.timestep_embedder(get_timestep_embedding(timestep, 256).to(pooled.dtype))
        return t_emb + self.text_embedder(pooled)


class AdaLayerNormZero(nn.Module):
    """6-chunk AdaLN-Zero, LN eps 1e-6, no affine (models/LaDCast_3D_model.py:485-486,524-529)."""

    def __init__(self, dim: int):
        super().__init__()
        self.linear = nn.Linear(dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        emb = self.linear(F.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    """3-chunk variant (shift, scale, gate), LN eps 1e-6 (models/LaDCast_3D_model.py:421,441)."""

    def __init__(self, dim: int):
        super().__init__()
        self.linear = nn.Linear(dim, 3 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        emb = self.linear(F.silu(emb))
        shift, scale, gate = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale[:, None]) + shift[:, None]
        return x, gate


class AdaLayerNormContinuous(nn.Module):
    """``LN(x)(1+scale)+shift`` with chunk order (scale, shift); eps 1e-7, no affine
    (models/LaDCast_3D_model.py:754-756,1044)."""

    def __init__(self, dim: int, cond_dim: int, eps: float = 1e-7):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)

    def forward(self, x, cond):
        emb = self.linear(F.silu(cond).to(x.dtype))
        scale, shift = emb.chunk(2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class _ActProj(nn.Module):
    def __init__(self, dim_in: int, dim_out: int, kind: str):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.kind = kind

    def forward(self, x):
        x = self.proj(x)
        if self.kind == "gelu-approximate":
            return F.gelu(x, approximate="tanh")
        if self.kind == "linear-silu":
            return F.silu(x)
        raise ValueError(self.kind)


class FeedForward(nn.Module):
    """``net = [act_proj, Dropout, Linear]`` (models/LaDCast_3D_model.py:271-276,503-512)."""

    def __init__(self, dim: int, mult: float = 4, activation_fn: str = "gelu-approximate"):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.ModuleList([_ActProj(dim, inner, activation_fn), nn.Dropout(0.0), nn.Linear(inner, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class Attention(nn.Module):
    """Parameter container with diffusers' ``Attention`` ctor semantics for the
    options the reference passes; the arithmetic lives in the
    processor (``oracle.ar_model.LaDCastAttnProcessor``)."""

    def __init__(
        self,
        query_dim: int,
        heads: int,
        dim_head: int,
        bias: bool = True,
        eps: float = 1e-7,
        added_kv_proj_dim: Optional[int] = None,
        pre_only: bool = False,
        processor=None,
    ):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.norm_q = RMSNorm(dim_head, eps=eps)
        self.norm_k = RMSNorm(dim_head, eps=eps)
        if added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_q_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.norm_added_q = RMSNorm(dim_head, eps=eps)
            self.norm_added_k = RMSNorm(dim_head, eps=eps)
            self.to_add_out = nn.Linear(inner, query_dim, bias=True)
        else:
            self.add_q_proj = self.add_k_proj = self.add_v_proj = None
            self.norm_added_q = self.norm_added_k = None
            self.to_add_out = None
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])
        else:
            self.to_out = None
        self.processor = processor

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(
            self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw
        )


def get_1d_rotary_pos_embed(dim: int, pos: torch.Tensor, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``use_real=True`` form: cos/sin of ``outer(pos, theta^(-2j/dim))`` each
    repeat-interleaved by 2 -> ``(S, dim)`` fp32 (models/embeddings.py:315-320)."""
    assert dim % 2 == 0
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=pos.device)[: dim // 2] / dim))
    ang = torch.outer(pos, freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def apply_rotary_emb(x: torch.Tensor, freqs_cis: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """Adjacent-pair rotation, ``x*cos + stack(-x_odd, x_even)*sin`` in fp32
    (models/LaDCast_3D_model.py:109-169; diffusers ``use_real_unbind_dim=-1``)."""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers ``randn_tensor``: with a list of generators draw ``(1,)+shape[1:]``
    per generator on the generator's device and concatenate
    (pipelines/edm_sampler.py:53-55, pipelines/pipeline_AR.py:77-82)."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    batch = shape[0]
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        one = (1,) + tuple(shape[1:])
        parts = [
            torch.randn(one, generator=generator[i], device=generator[i].device, dtype=dtype) for i in range(batch)
        ]
        return torch.cat(parts, dim=0).to(device)
    rand_device = generator.device if generator is not None else device
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)
