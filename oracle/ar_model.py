"""Oracle restatement of ``LaDCastTransformer3DModel``
(models/LaDCast_3D_model.py:64-1071, models/embeddings.py:38-59,252-327,422-520).

Module / parameter names equal the reference's so state dicts interchange.
The diffusers layer pieces come from ``oracle.layers`` (PARITY UNPINNED); every forward in THIS file is PINNED to the reference's
own forward code (tests/test_oracle_reference_pins.py::test_transformer_forward_equals_the_reference_forward_code).
"""
from __future__ import annotations

import math
from datetime import datetime
from types import SimpleNamespace
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .autocast import fp32_island
from .layers import (
    AdaLayerNormContinuous,
    AdaLayerNormZero,
    AdaLayerNormZeroSingle,
    Attention,
    CombinedTimestepTextProjEmbeddings,
    FeedForward,
    TimestepEmbedding,
    apply_rotary_emb,
    get_1d_rotary_pos_embed,
)


# ---------------------------------------------------------------------------
# year-progress embedding (models/embeddings.py:428-520)
# ---------------------------------------------------------------------------
def convert_int_to_datetime(ts: int) -> datetime:
    s = str(int(ts))
    return datetime(int(s[0:4]), int(s[4:6]), int(s[6:8]), int(s[8:10]))


def compute_year_progress(dt: datetime) -> float:
    start = datetime(dt.year, 1, 1)
    end = datetime(dt.year + 1, 1, 1)
    return (dt - start).total_seconds() / (end - start).total_seconds()


def get_year_sincos_embedding(timestamps: torch.Tensor, embedding_dim: int = 256, max_period: int = 10000):
    """(B,) int YYYYMMDDHH -> (B, embedding_dim): ``[sin(2 pi p k) m_k | cos(2 pi p k) m_k]``,
    k = 1..half, ``m_k = exp(-ln(max_period) (k-1)/half)`` (models/embeddings.py:467-520)."""
    prog = torch.tensor(
        [compute_year_progress(convert_int_to_datetime(int(t))) for t in timestamps.tolist()], dtype=torch.float32
    )
    half = embedding_dim // 2
    freqs = torch.arange(1, half + 1).float()
    mag = torch.exp(-math.log(max_period) * torch.arange(0, half).float() / half)
    arg = (2 * math.pi * prog.reshape(-1, 1)) * freqs.reshape(1, -1)
    emb = torch.zeros((timestamps.shape[0], embedding_dim))
    emb[:, :half] = torch.sin(arg) * mag.reshape(1, -1)
    emb[:, half:] = torch.cos(arg) * mag.reshape(1, -1)
    return emb


# ---------------------------------------------------------------------------
# embeddings
# ---------------------------------------------------------------------------
class HunyuanVideoPatchEmbed(nn.Module):
    """Conv3d patch embed then ``flatten(2).transpose(1,2)`` (models/embeddings.py:38-59)."""

    def __init__(self, patch_size: Tuple[int, int, int], in_chans: int, embed_dim: int):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def rope_from_grid(rope_dim_list: Sequence[int], grids: Sequence[torch.Tensor], theta: float):
    """``LaDCastRotaryPosEmbed_from_grid.forward`` (models/embeddings.py:274-327):
    ij-meshgrid of the axis coordinates, per-axis 1-D rotary tables, concatenated on dim 1."""
    mesh = torch.stack(torch.meshgrid(*[g.to(torch.float32) for g in grids], indexing="ij"), dim=0)
    cos, sin = [], []
    for i, d in enumerate(rope_dim_list):
        c, s = get_1d_rotary_pos_embed(d, mesh[i].reshape(-1), theta)
        cos.append(c)
        sin.append(s)
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


# ---------------------------------------------------------------------------
# attention processor (models/LaDCast_3D_model.py:64-221)
# ---------------------------------------------------------------------------
class LaDCastAttnProcessor:
    def __call__(
        self,
        attn: Attention,
        hidden_states,
        encoder_hidden_states=None,
        attention_mask=None,
        image_rotary_emb=None,
        cond_image_rotary_emb=None,
    ):
        single = attn.add_q_proj is None and encoder_hidden_states is not None
        if single:
            hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)

        def heads(t):
            return t.unflatten(2, (attn.heads, -1)).transpose(1, 2)

        q, k, v = heads(attn.to_q(hidden_states)), heads(attn.to_k(hidden_states)), heads(attn.to_v(hidden_states))
        q, k = attn.norm_q(q), attn.norm_k(k)

        if image_rotary_emb is not None:
            if single:
                nc = encoder_hidden_states.shape[1]
                assert cond_image_rotary_emb is not None
                q = torch.cat(
                    [apply_rotary_emb(q[:, :, :-nc], image_rotary_emb), apply_rotary_emb(q[:, :, -nc:], cond_image_rotary_emb)],
                    dim=2,
                )
                k = torch.cat(
                    [apply_rotary_emb(k[:, :, :-nc], image_rotary_emb), apply_rotary_emb(k[:, :, -nc:], cond_image_rotary_emb)],
                    dim=2,
                )
            else:
                q = apply_rotary_emb(q, image_rotary_emb)
                k = apply_rotary_emb(k, image_rotary_emb)

        if attn.add_q_proj is not None and encoder_hidden_states is not None:
            eq = attn.norm_added_q(heads(attn.add_q_proj(encoder_hidden_states)))
            ek = attn.norm_added_k(heads(attn.add_k_proj(encoder_hidden_states)))
            ev = heads(attn.add_v_proj(encoder_hidden_states))
            q, k, v = torch.cat([q, eq], dim=2), torch.cat([k, ek], dim=2), torch.cat([v, ev], dim=2)

        out = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        out = out.transpose(1, 2).flatten(2, 3).to(q.dtype)

        if encoder_hidden_states is not None:
            nc = encoder_hidden_states.shape[1]
            out, enc = out[:, :-nc], out[:, -nc:]
            if attn.to_out is not None:
                out = attn.to_out[1](attn.to_out[0](out))
            if attn.to_add_out is not None:
                enc = attn.to_add_out(enc)
            return out, enc
        return out, None


# ---------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------
class HunyuanVideoAdaNorm(nn.Module):
    """models/LaDCast_3D_model.py:224-238"""

    def __init__(self, in_features: int, out_features: Optional[int] = None):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features or 2 * in_features)

    def forward(self, temb):
        g1, g2 = self.linear(F.silu(temb)).chunk(2, dim=1)
        return g1.unsqueeze(1), g2.unsqueeze(1)


class RefinerBlock(nn.Module):
    """models/LaDCast_3D_model.py:241-302"""

    def __init__(self, heads: int, head_dim: int, mlp_width_ratio: float = 4.0):
        super().__init__()
        d = heads * head_dim
        self.norm1 = nn.LayerNorm(d, elementwise_affine=True, eps=1e-7)
        self.attn = Attention(d, heads, head_dim, bias=True, eps=1e-7, pre_only=True, processor=LaDCastAttnProcessor())
        self.norm2 = nn.LayerNorm(d, elementwise_affine=True, eps=1e-7)
        self.ff = FeedForward(d, mult=mlp_width_ratio, activation_fn="linear-silu")
        self.norm_out = HunyuanVideoAdaNorm(d, 2 * d)

    def forward(self, x, temb, attention_mask=None, image_rotary_emb=None):
        a, _ = self.attn(self.norm1(x), encoder_hidden_states=None, attention_mask=attention_mask, image_rotary_emb=image_rotary_emb)
        gate_msa, gate_mlp = self.norm_out(temb)
        x = x + a * gate_msa
        return x + self.ff(self.norm2(x)) * gate_mlp


class IndividualTokenRefiner(nn.Module):
    def __init__(self, heads, head_dim, num_layers):
        super().__init__()
        self.refiner_blocks = nn.ModuleList([RefinerBlock(heads, head_dim) for _ in range(num_layers)])

    def forward(self, x, temb, attention_mask=None, image_rotary_emb=None):
        for b in self.refiner_blocks:
            x = b(x, temb, attention_mask, image_rotary_emb)
        return x


class TokenRefiner(nn.Module):
    """models/LaDCast_3D_model.py:345-390"""

    def __init__(self, in_channels: int, heads: int, head_dim: int, num_layers: int):
        super().__init__()
        d = heads * head_dim
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(d, in_channels)
        self.proj_in = nn.Linear(in_channels, d, bias=True)
        self.token_refiner = IndividualTokenRefiner(heads, head_dim, num_layers)

    def forward(self, x, timestep, attention_mask=None, image_rotary_emb=None):
        temb = self.time_text_embed(timestep, x.mean(dim=1))
        return self.token_refiner(self.proj_in(x), temb, attention_mask, image_rotary_emb)


class SingleBlock(nn.Module):
    """models/LaDCast_3D_model.py:394-468"""

    def __init__(self, heads: int, head_dim: int, mlp_ratio: float = 4.0):
        super().__init__()
        d = heads * head_dim
        mlp = int(d * mlp_ratio)
        self.attn = Attention(d, heads, head_dim, bias=True, eps=1e-7, pre_only=True, processor=LaDCastAttnProcessor())
        self.norm = AdaLayerNormZeroSingle(d)
        self.proj_mlp = nn.Linear(d, mlp)
        self.proj_out = nn.Linear(d + mlp, d)

    def forward(self, x, ctx, temb, attention_mask=None, image_rotary_emb=None, cond_image_rotary_emb=None):
        nc = ctx.shape[1]
        h = torch.cat([x, ctx], dim=1)
        residual = h
        nh, gate = self.norm(h, emb=temb)
        mlp = F.gelu(self.proj_mlp(nh), approximate="tanh")
        a, ca = self.attn(
            nh[:, :-nc],
            encoder_hidden_states=nh[:, -nc:],
            attention_mask=attention_mask,
            image_rotary_emb=image_rotary_emb,
            cond_image_rotary_emb=cond_image_rotary_emb,
        )
        h = torch.cat([torch.cat([a, ca], dim=1), mlp], dim=2)
        h = gate.unsqueeze(1) * self.proj_out(h) + residual
        return h[:, :-nc], h[:, -nc:]


class DualBlock(nn.Module):
    """models/LaDCast_3D_model.py:472-566"""

    def __init__(self, heads: int, head_dim: int, mlp_ratio: float):
        super().__init__()
        d = heads * head_dim
        self.norm1 = AdaLayerNormZero(d)
        self.norm1_context = AdaLayerNormZero(d)
        self.attn = Attention(d, heads, head_dim, bias=True, eps=1e-7, added_kv_proj_dim=d, processor=LaDCastAttnProcessor())
        self.norm2 = nn.LayerNorm(d, elementwise_affine=False, eps=1e-7)
        self.ff = FeedForward(d, mult=mlp_ratio, activation_fn="gelu-approximate")
        self.norm2_context = nn.LayerNorm(d, elementwise_affine=False, eps=1e-7)
        self.ff_context = FeedForward(d, mult=mlp_ratio, activation_fn="gelu-approximate")

    def forward(self, x, ctx, temb, attention_mask=None, freqs_cis=None, cond_freqs_cis=None):
        nx, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(x, emb=temb)
        nc, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(ctx, emb=temb)
        a, ca = self.attn(
            nx,
            encoder_hidden_states=nc,
            attention_mask=attention_mask,
            image_rotary_emb=freqs_cis,
            cond_image_rotary_emb=cond_freqs_cis,
        )
        x = x + a * gate_msa.unsqueeze(1)
        ctx = ctx + ca * c_gate_msa.unsqueeze(1)
        nx = self.norm2(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        nc = self.norm2_context(ctx) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        x = x + gate_mlp.unsqueeze(1) * self.ff(nx)
        ctx = ctx + c_gate_mlp.unsqueeze(1) * self.ff_context(nc)
        return x, ctx


# ---------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------
CONFIG_375M = dict(
    in_channels=84,
    out_channels=84,
    num_attention_heads=12,
    attention_head_dim=128,
    num_layers=2,
    num_single_layers=4,
    num_refiner_layers=1,
    mlp_ratio=4,
    patch_size=1,
    patch_size_t=1,
    qk_norm="rms_norm",
    rope_theta=256.0,
    rope_axes_dim=(16, 56, 56),
    rope_spatial_grid_start_pos=(-499.5, 5.25),
    rope_spatial_grid_end_pos=(508.5, 353.25),
    spatial_deg2rad=True,
    conditioning_tensor_in_channels=84,
    conditioning_tensor_rope_axes_dim=(16, 56, 56),
    incl_time_elapsed=True,
)  # configs/ladcast_375M.yaml:1-30
CONFIG_1_6B = dict(CONFIG_375M, num_attention_heads=16, num_layers=5, num_single_layers=10, num_refiner_layers=3)
# configs/ladcast_1.6B.yaml:5-9


class LaDCastTransformer3DModel(nn.Module):
    """models/LaDCast_3D_model.py:569-1071 (``nope`` - temporal-only rotary embedding - is restated since round 5;
    ``scale_attn_by_lat`` - also off in both shipped configs - is restated: a (1, 1, 1, keys) float mask of normalised
    cos-latitude weights ADDED to the attention scores of every block, :682-693,873-882,950)."""

    def __init__(
        self,
        in_channels: int = 16,
        out_channels: int = 16,
        num_attention_heads: int = 24,
        attention_head_dim: int = 128,
        num_layers: int = 20,
        num_single_layers: int = 40,
        num_refiner_layers: int = 2,
        mlp_ratio: float = 4.0,
        patch_size: int = 1,
        patch_size_t: int = 1,
        qk_norm: str = "rms_norm",
        rope_theta: float = 256.0,
        rope_axes_dim: Tuple[int, ...] = (16, 56, 56),
        rope_spatial_grid_start_pos=0,
        rope_spatial_grid_end_pos=None,
        spatial_deg2rad: bool = False,
        conditioning_tensor_in_channels: int = None,
        conditioning_tensor_intermediate_proj_dim: Optional[int] = None,
        conditioning_tensor_rope_axes_dim: Tuple[int, ...] = (16, 56, 56),
        incl_time_elapsed: bool = False,
        nope: bool = False,
        scale_attn_by_lat: bool = False,
    ):
        super().__init__()
        # nope=True (models/LaDCast_3D_model.py:710-712,897-918): no spatial rotary embedding - the whole head dimension rotates with the
        # temporal coordinate only (see rope_tables)
        if scale_attn_by_lat and patch_size != 1:
            raise NotImplementedError("scale_attn_by_lat hard-wires the 15 x 30 token grid (models/LaDCast_3D_model.py:684-693): patch size 1 only")
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        d = num_attention_heads * attention_head_dim
        out_channels = out_channels or in_channels
        ps = (patch_size_t, patch_size, patch_size)
        self.x_embedder = HunyuanVideoPatchEmbed(ps, in_channels, d)
        if conditioning_tensor_intermediate_proj_dim is None:
            conditioning_tensor_intermediate_proj_dim = d
        self.context_embedder = HunyuanVideoPatchEmbed(ps, conditioning_tensor_in_channels, d)
        self.context_refiner = TokenRefiner(
            conditioning_tensor_intermediate_proj_dim, num_attention_heads, attention_head_dim, num_refiner_layers
        )
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(d, d)
        self.time_elapsed_embed = TimestepEmbedding(256, 2 * d) if incl_time_elapsed else None
        if spatial_deg2rad:
            rope_spatial_grid_start_pos = [float(np.deg2rad(v)) for v in rope_spatial_grid_start_pos]
            rope_spatial_grid_end_pos = [float(np.deg2rad(v)) for v in rope_spatial_grid_end_pos]
        self.rope_spatial_grid_start_pos = rope_spatial_grid_start_pos
        self.rope_spatial_grid_end_pos = rope_spatial_grid_end_pos
        assert sum(rope_axes_dim) == attention_head_dim
        assert sum(conditioning_tensor_rope_axes_dim) == attention_head_dim
        self.transformer_blocks = nn.ModuleList(
            [DualBlock(num_attention_heads, attention_head_dim, mlp_ratio) for _ in range(num_layers)]
        )
        self.single_transformer_blocks = nn.ModuleList(
            [SingleBlock(num_attention_heads, attention_head_dim, mlp_ratio) for _ in range(num_single_layers)]
        )
        self.norm_out = AdaLayerNormContinuous(d, d, eps=1e-7)
        self.proj_out = nn.Linear(d, patch_size_t * patch_size * patch_size * out_channels)
        self.scale_attn_by_lat = scale_attn_by_lat
        if scale_attn_by_lat:  # :682-693 (the 15 x 30 latent grid is hard-wired in the reference)
            w = np.cos(np.deg2rad(np.linspace(-83.25, 84.75, 15)))
            w = w / w.mean()  # evaluate/utils.py:40-48
            w = torch.from_numpy(w / w.sum()).float()
            self.attn_lat_weights = w.repeat_interleave(30).view(1, 1, 1, -1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_config(cls, cfg: dict):
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        return cls(**cfg)

    def rope_tables(self, return_seq_len: int, input_seq_len: int, height: int, width: int, device=None):
        """models/LaDCast_3D_model.py:885-938"""
        c = self.config
        cond_t = torch.arange(-input_seq_len + 1, 1, device=device, dtype=torch.float32)
        pred_t = torch.arange(1, return_seq_len + 1, device=device, dtype=torch.float32)
        lat = torch.linspace(
            self.rope_spatial_grid_start_pos[0], self.rope_spatial_grid_end_pos[0], steps=height, device=device, dtype=torch.float32
        )
        lon = torch.linspace(
            self.rope_spatial_grid_start_pos[1], self.rope_spatial_grid_end_pos[1], steps=width, device=device, dtype=torch.float32
        )
        if getattr(c, "nope", False):  # :897-918: get_1d_rotary_pos_embed(head_dim, temporal coordinate), every frame's row repeated over its h * w tokens
            from .layers import get_1d_rotary_pos_embed

            pc, ps = get_1d_rotary_pos_embed(c.attention_head_dim, pred_t, c.rope_theta)
            cc, cs = get_1d_rotary_pos_embed(c.attention_head_dim, cond_t, c.rope_theta)
            n = height * width
            return (pc.repeat_interleave(n, dim=0), ps.repeat_interleave(n, dim=0)), (cc.repeat_interleave(n, dim=0), cs.repeat_interleave(n, dim=0))
        pred = rope_from_grid(c.rope_axes_dim, [pred_t, lat, lon], c.rope_theta)
        cond = rope_from_grid(c.conditioning_tensor_rope_axes_dim, [cond_t, lat, lon], c.rope_theta)
        return pred, cond

    def forward(
        self,
        hidden_states,
        timestep,
        conditioning_tensors,
        time_elapsed=None,
        attention_kwargs=None,
        return_dict: bool = True,
        coords=None,
    ):
        b, _, r, h, w = hidden_states.shape
        t_in = conditioning_tensors.shape[2]
        p_, pt_ = self.config.patch_size, self.config.patch_size_t  # :866-871: everything below counts PATCHES
        r, t_in, h, w = r // pt_, t_in // pt_, h // p_, w // p_
        with fp32_island():  # models/embeddings.py:282 (the rotary tables are built inside an fp32 island)
            image_rope, cond_rope = self.rope_tables(r, t_in, h, w, device=hidden_states.device)

        x = self.x_embedder(hidden_states)
        ctx = self.context_embedder(conditioning_tensors)
        pred_mask = cond_mask = None
        if self.scale_attn_by_lat:  # :873-880
            pred_mask = self.attn_lat_weights.repeat(1, 1, 1, t_in + r).to(hidden_states.device)
            cond_mask = self.attn_lat_weights.repeat(1, 1, 1, t_in).to(hidden_states.device)
        ctx = self.context_refiner(ctx, timestep, image_rotary_emb=cond_rope, attention_mask=cond_mask)

        # :953-969 - the one fp32 island of the transformer under mixed precision (oracle/autocast.py); a no-op in the fp32 oracle
        with fp32_island():
            temb = self.time_text_embed(timestep, ctx.mean(dim=1).float())
            if time_elapsed is not None and self.time_elapsed_embed is not None:
                te = get_year_sincos_embedding(time_elapsed, embedding_dim=256)
                te = self.time_elapsed_embed(te.to(hidden_states.device))
                scale, shift = te.chunk(2, dim=-1)
                temb = temb * (1 + scale) + shift

        for blk in self.transformer_blocks:
            x, ctx = blk(x, ctx, temb, pred_mask, image_rope, cond_rope)
        for blk in self.single_transformer_blocks:
            x, ctx = blk(x, ctx, temb, pred_mask, image_rope, cond_rope)

        x = self.proj_out(self.norm_out(x, temb))
        x = x.reshape(b, r, h, w, -1, pt_, p_, p_).permute(0, 4, 1, 5, 2, 6, 3, 7)  # :1047-1062: (B, C, T', p_t, H', p, W', p)
        x = x.flatten(6, 7).flatten(4, 5).flatten(2, 3)
        if not return_dict:
            return (x,)
        return SimpleNamespace(sample=x)
