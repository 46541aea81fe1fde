"""Oracle restatement of ``AutoencoderDC`` and its blocks (models/DCAE.py:67-1087)
for the shipped configuration (configs/DC_AE_84_pretrain.yaml:1-48: ``rms_norm``, pixel-(un)shuffle sampling, no timestep
conditioning) and, since round 5, the constructor variants ``upsample_block_type="interpolate"`` and ``temb_channels`` (the
timestep-conditioned ResBlock / linear attention, models/DCAE.py:36-64,110-153,193-198,334-365,786-850,982-984).

Parameter names equal the reference's.  RMSNorm comes from
``oracle.layers`` (PARITY UNPINNED); SphereConv2d from ``oracle.sphere_conv`` (PINNED); every forward in THIS file is PINNED to the
reference's own forward code (tests/test_oracle_reference_pins.py::test_dcae_forward_equals_the_reference_forward_code).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .autocast import fp32_island
from .layers import RMSNorm, TimestepEmbedding, get_timestep_embedding
from .sphere_conv import SphereConv2d


def _act(name):
    return {"silu": nn.SiLU(), "relu": nn.ReLU(), "relu6": nn.ReLU6(), "gelu": nn.GELU(), "mish": nn.Mish()}[name]


def _chan_rmsnorm(norm: RMSNorm, x):
    return norm(x.movedim(1, -1)).movedim(-1, 1)


class SanaMultiscaleAttentionProjection(nn.Module):
    """models/DCAE.py:67-93: depthwise sphere conv k then grouped 1x1 (groups = 3*heads)."""

    def __init__(self, in_channels: int, num_attention_heads: int, kernel_size: int):
        super().__init__()
        ch = 3 * in_channels
        self.proj_in = SphereConv2d(ch, ch, kernel_size, padding=kernel_size // 2, groups=ch, bias=False)
        self.proj_out = nn.Conv2d(ch, ch, 1, 1, 0, groups=3 * num_attention_heads, bias=False)

    def forward(self, x):
        return self.proj_out(self.proj_in(x))


class AdaLayerNormZeroSingle4Sana(nn.Module):
    """models/DCAE.py:36-64 (the reference's own class): emb -> Linear(SiLU(emb)) -> (shift, scale, gate); LayerNorm over the channels of every
    pixel without affine, eps 1e-15 (diffusers' FP32LayerNorm: computed in fp32, cast back), x_hat (1 + scale) + shift."""

    def __init__(self, embedding_dim: int, bias: bool = True):
        super().__init__()
        self.linear = nn.Linear(embedding_dim, 3 * embedding_dim, bias=bias)
        self.embedding_dim = embedding_dim

    def forward(self, x, emb):
        emb = self.linear(F.silu(emb))
        shift, scale, gate = emb.chunk(3, dim=1)
        xl = x.movedim(1, -1)
        xn = F.layer_norm(xl.float(), (self.embedding_dim,), None, None, 1e-15).to(xl.dtype)
        return (xn * (1 + scale[:, None, None]) + shift[:, None, None]).movedim(-1, 1), gate[:, :, None, None]


class SanaMultiscaleLinearAttention(nn.Module):
    """models/DCAE.py:96-267 (processor folded in).  Heads = ``int(C // 32)`` so the
    inner width is 480 / 992 for C = 504 / 1008; the multi-scale concat is regrouped
    as consecutive 96-channel groups split (q, k, v) = (32, 32, 32) -- reference quirk,
    reproduced literally (models/DCAE.py:226-243)."""

    def __init__(self, in_channels, out_channels, attention_head_dim=32, mult=1.0, kernel_sizes=(5,), eps=1e-15, residual_connection=True,
                 temb_channels=None):
        super().__init__()
        if temb_channels is not None:  # models/DCAE.py:147-153
            self.time_emb_porj = nn.Linear(temb_channels, out_channels)
            self.norm_in = AdaLayerNormZeroSingle4Sana(out_channels)
        else:
            self.time_emb_porj, self.norm_in = None, None
        self.eps = eps
        self.attention_head_dim = attention_head_dim
        self.residual_connection = residual_connection
        heads = int(in_channels // attention_head_dim * mult)
        inner = heads * attention_head_dim
        self.to_q = nn.Linear(in_channels, inner, bias=False)
        self.to_k = nn.Linear(in_channels, inner, bias=False)
        self.to_v = nn.Linear(in_channels, inner, bias=False)
        self.to_qkv_multiscale = nn.ModuleList(
            [SanaMultiscaleAttentionProjection(inner, heads, ks) for ks in kernel_sizes]
        )
        self.to_out = nn.Linear(inner * (1 + len(kernel_sizes)), out_channels, bias=False)
        self.norm_out = RMSNorm(out_channels, eps=1e-5, elementwise_affine=True, bias=True)  # get_normalization default eps

    def forward(self, x, temb=None):
        gate = None
        if self.norm_in is not None:  # models/DCAE.py:193-198: the block's own projection of relu(temb), then the AdaLN; the residual below
            x, gate = self.norm_in(x, self.time_emb_porj(F.relu(temb)))  # is the NORMALISED tensor (the processor takes it from its input, :217)
        b, _, h, w = x.shape
        residual = x
        xl = x.movedim(1, -1)
        qkv = torch.cat([self.to_q(xl), self.to_k(xl), self.to_v(xl)], dim=3).movedim(-1, 1)
        multi = [qkv] + [blk(qkv) for blk in self.to_qkv_multiscale]
        hs = torch.cat(multi, dim=1)
        if h * w <= self.attention_head_dim:
            raise NotImplementedError("quadratic branch is never taken at 15x30 / 30x60")
        original_dtype = hs.dtype  # bf16 under the reference's mixed precision (oracle/autocast.py); fp32 otherwise
        hs = hs.to(torch.float32).reshape(b, -1, 3 * self.attention_head_dim, h * w)
        q, k, v = hs.chunk(3, dim=2)
        q, k = F.relu(q), F.relu(k)
        v = F.pad(v, (0, 0, 0, 1), mode="constant", value=1)
        with fp32_island():  # models/DCAE.py:162-175
            scores = torch.matmul(v.to(torch.float32), k.transpose(-1, -2).to(torch.float32))
            out = torch.matmul(scores.to(torch.float32), q.to(torch.float32))
            out = out[:, :, :-1] / (out[:, :, -1:] + self.eps)
        out = out.to(original_dtype)  # models/DCAE.py:249
        out = out.reshape(b, -1, h, w)
        out = self.to_out(out.movedim(1, -1)).movedim(-1, 1)
        if gate is not None:  # models/DCAE.py:256-257
            out = out * gate
        out = _chan_rmsnorm(self.norm_out, out)
        if self.residual_connection:
            out = out + residual
        return out


class GLUMBConv(nn.Module):
    """models/DCAE.py:270-324"""

    def __init__(self, in_channels, out_channels, expand_ratio=4):
        super().__init__()
        hid = int(expand_ratio * in_channels)
        self.conv_inverted = nn.Conv2d(in_channels, hid * 2, 1, 1, 0)
        self.conv_depth = SphereConv2d(hid * 2, hid * 2, 3, 1, 1, groups=hid * 2)
        self.conv_point = nn.Conv2d(hid, out_channels, 1, 1, 0, bias=False)
        self.norm = RMSNorm(out_channels, eps=1e-7, elementwise_affine=True, bias=True)

    def forward(self, x):
        residual = x
        x = F.silu(self.conv_inverted(x))
        x = self.conv_depth(x)
        x, gate = torch.chunk(x, 2, dim=1)
        x = x * F.silu(gate)
        x = self.conv_point(x)
        x = _chan_rmsnorm(self.norm, x)
        return x + residual


class ResBlock(nn.Module):
    """models/DCAE.py:327-377"""

    def __init__(self, in_channels, out_channels, act_fn="silu", temb_channels=None):
        super().__init__()
        self.nonlinearity = _act(act_fn)
        self.conv1 = SphereConv2d(in_channels, in_channels, 3, 1, 1)
        self.conv2 = SphereConv2d(in_channels, out_channels, 3, 1, 1, bias=False)
        self.norm = RMSNorm(out_channels, eps=1e-5, elementwise_affine=True, bias=True)
        self.time_emb_porj = nn.Linear(temb_channels, 2 * out_channels) if temb_channels is not None else None  # models/DCAE.py:351-354

    def forward(self, x, temb=None):
        residual = x
        x = self.nonlinearity(self.conv1(x))
        if self.time_emb_porj is not None:  # models/DCAE.py:361-365: scale and shift (not 1 + scale) from the block's activation of temb
            t = self.time_emb_porj(self.nonlinearity(temb))[:, :, None, None]
            scale, shift = torch.chunk(t, 2, dim=1)
            x = x * scale + shift
        x = self.conv2(x)
        x = _chan_rmsnorm(self.norm, x)
        return x + residual


class EfficientViTBlock(nn.Module):
    """models/DCAE.py:380-414"""

    def __init__(self, in_channels, attention_head_dim=32, qkv_multiscales=(5,), temb_channels=None):
        super().__init__()
        self.attn = SanaMultiscaleLinearAttention(
            in_channels, in_channels, attention_head_dim=attention_head_dim, kernel_sizes=qkv_multiscales, temb_channels=temb_channels
        )
        self.conv_out = GLUMBConv(in_channels, in_channels)

    def forward(self, x, temb=None):
        return self.conv_out(self.attn(x, temb))


def get_block(block_type, channels, attention_head_dim, act_fn, qkv_multiscales, temb_channels=None):
    if block_type == "ResBlock":
        return ResBlock(channels, channels, act_fn, temb_channels=temb_channels)
    if block_type == "EfficientViTBlock":
        return EfficientViTBlock(channels, attention_head_dim, tuple(qkv_multiscales), temb_channels=temb_channels)
    raise ValueError(f"Block with {block_type=} is not supported.")


class DCDownBlock2d(nn.Module):
    """models/DCAE.py:447-490 (pixel_unshuffle form)"""

    def __init__(self, in_channels, out_channels, shortcut=True):
        super().__init__()
        self.factor = 2
        self.group_size = in_channels * self.factor**2 // out_channels
        self.shortcut = shortcut
        assert out_channels % self.factor**2 == 0
        self.conv = SphereConv2d(in_channels, out_channels // self.factor**2, 3, 1, 1)

    def forward(self, x, temb=None):
        y = F.pixel_unshuffle(self.conv(x), self.factor)
        if self.shortcut:
            s = F.pixel_unshuffle(x, self.factor)
            s = s.unflatten(1, (-1, self.group_size)).mean(dim=2)
            y = y + s
        return y


class DCUpBlock2d(nn.Module):
    """models/DCAE.py:493-536: the pixel_shuffle form, or (interpolate=True, `upsample_block_type="interpolate"`, :498-525) nearest-neighbour
    x2 up-sampling followed by a conv at the output width"""

    def __init__(self, in_channels, out_channels, shortcut=True, interpolate=False, interpolation_mode="nearest"):
        super().__init__()
        self.factor = 2
        self.repeats = out_channels * self.factor**2 // in_channels
        self.shortcut = shortcut
        self.interpolate, self.interpolation_mode = interpolate, interpolation_mode
        self.conv = SphereConv2d(in_channels, out_channels if interpolate else out_channels * self.factor**2, 3, 1, 1)

    def forward(self, x, temb=None):
        if self.interpolate:
            y = self.conv(F.interpolate(x, scale_factor=self.factor, mode=self.interpolation_mode))
        else:
            y = F.pixel_shuffle(self.conv(x), self.factor)
        if self.shortcut:
            s = F.pixel_shuffle(x.repeat_interleave(self.repeats, dim=1), self.factor)
            y = y + s
        return y


class Encoder(nn.Module):
    """models/DCAE.py:539-631"""

    def __init__(self, in_channels, latent_channels, attention_head_dim, block_type, block_out_channels, layers_per_block, qkv_multiscales,
                 temb_channels=None):
        super().__init__()
        n = len(block_out_channels)
        if layers_per_block[0] > 0:
            self.conv_in = SphereConv2d(in_channels, block_out_channels[0], 3, 1, 1)
        else:  # models/DCAE.py:571-579: no stage at full resolution - conv_in is a down block WITHOUT shortcut straight to the second width
            self.conv_in = DCDownBlock2d(in_channels, block_out_channels[1], shortcut=False)
        self.down_blocks = nn.ModuleList()
        for i, (ch, nl) in enumerate(zip(block_out_channels, layers_per_block)):
            for _ in range(nl):
                self.down_blocks.append(get_block(block_type[i], ch, attention_head_dim, "silu", qkv_multiscales[i], temb_channels))
            if i < n - 1 and nl > 0:
                self.down_blocks.append(DCDownBlock2d(ch, block_out_channels[i + 1], shortcut=True))
        self.conv_out = SphereConv2d(block_out_channels[-1], latent_channels, 3, 1, 1)
        self.out_shortcut_average_group_size = block_out_channels[-1] // latent_channels

    def forward(self, x, temb=None):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            x = blk(x, temb)
        s = x.unflatten(1, (-1, self.out_shortcut_average_group_size)).mean(dim=2)
        return self.conv_out(x) + s


class Decoder(nn.Module):
    """models/DCAE.py:634-732"""

    def __init__(self, out_channels, latent_channels, attention_head_dim, block_type, block_out_channels, layers_per_block, qkv_multiscales, act_fn="silu",
                 upsample_block_type="pixel_shuffle", temb_channels=None):
        super().__init__()
        n = len(block_out_channels)
        self.conv_in = SphereConv2d(latent_channels, block_out_channels[-1], 3, 1, 1)
        self.in_shortcut_repeats = block_out_channels[-1] // latent_channels
        self.up_blocks = nn.ModuleList()
        for i, (ch, nl) in reversed(list(enumerate(zip(block_out_channels, layers_per_block)))):
            if i < n - 1 and nl > 0:
                self.up_blocks.append(DCUpBlock2d(block_out_channels[i + 1], ch, shortcut=True, interpolate=upsample_block_type == "interpolate"))  # models/DCAE.py:677-682
            for _ in range(nl):
                self.up_blocks.append(get_block(block_type[i], ch, attention_head_dim, act_fn if isinstance(act_fn, str) else act_fn[i], qkv_multiscales[i], temb_channels))
        ch0 = block_out_channels[0] if layers_per_block[0] > 0 else block_out_channels[1]  # models/DCAE.py:696-712
        self.norm_out = RMSNorm(ch0, 1e-7, elementwise_affine=True, bias=True)
        if layers_per_block[0] > 0:
            self.conv_out = SphereConv2d(ch0, out_channels, 3, 1, 1)
        else:  # the mirror of the encoder's conv_in: an up block WITHOUT shortcut from the second width to the fields
            self.conv_out = DCUpBlock2d(ch0, out_channels, shortcut=False, interpolate=upsample_block_type == "interpolate")

    def forward(self, z, temb=None):
        x = self.conv_in(z) + z.repeat_interleave(self.in_shortcut_repeats, dim=1)
        for blk in self.up_blocks:
            x = blk(x, temb)
        x = F.relu(_chan_rmsnorm(self.norm_out, x))
        return self.conv_out(x)


CONFIG_DCAE_84 = dict(
    in_channels=89,
    out_channels=89,
    latent_channels=84,
    attention_head_dim=32,
    encoder_block_types=("ResBlock", "ResBlock", "EfficientViTBlock", "EfficientViTBlock"),
    decoder_block_types=("ResBlock", "ResBlock", "EfficientViTBlock", "EfficientViTBlock"),
    encoder_block_out_channels=(252, 504, 504, 1008),
    decoder_block_out_channels=(252, 504, 504, 1008),
    encoder_layers_per_block=(4, 4, 4, 4),
    decoder_layers_per_block=(4, 4, 4, 4),
    encoder_qkv_multiscales=((), (), (5,), (5,)),
    decoder_qkv_multiscales=((), (), (5,), (5,)),
    upsample_block_type="pixel_shuffle",
    downsample_block_type="pixel_unshuffle",
    static_channels=5,
)  # configs/DC_AE_84_pretrain.yaml:1-48


class AutoencoderDC(nn.Module):
    """models/DCAE.py:735-1087"""

    def __init__(
        self,
        in_channels: int = 3,
        out_channels: Optional[int] = None,
        temb_channels: Optional[int] = None,
        latent_channels: int = 32,
        attention_head_dim: int = 32,
        encoder_block_types="ResBlock",
        decoder_block_types="ResBlock",
        encoder_block_out_channels: Tuple[int, ...] = (128, 256, 512, 512, 1024, 1024),
        decoder_block_out_channels: Tuple[int, ...] = (128, 256, 512, 512, 1024, 1024),
        encoder_layers_per_block: Tuple[int, ...] = (2, 2, 2, 3, 3, 3),
        decoder_layers_per_block: Tuple[int, ...] = (3, 3, 3, 3, 3, 3),
        encoder_qkv_multiscales=((), (), (), (5,), (5,), (5,)),
        decoder_qkv_multiscales=((), (), (), (5,), (5,), (5,)),
        upsample_block_type: str = "pixel_shuffle",
        downsample_block_type: str = "pixel_unshuffle",
        decoder_norm_types="rms_norm",
        decoder_act_fns="silu",
        scaling_factor: float = 1.0,
        static_channels: int = 0,
    ):
        super().__init__()
        if upsample_block_type not in ("pixel_shuffle", "interpolate") or downsample_block_type != "pixel_unshuffle":
            # (downsample_block_type "conv" builds a stride-2 SphereConv2d, which the reference's SphereConv2d refuses: sphere_conv.py asserts stride 1)
            raise NotImplementedError("sampling: pixel_unshuffle down, pixel_shuffle | interpolate up")
        acts = (decoder_act_fns,) * len(decoder_block_out_channels) if isinstance(decoder_act_fns, str) else tuple(decoder_act_fns)  # models/DCAE.py:663-664
        if decoder_norm_types != "rms_norm" or any(a not in ("silu", "relu") for a in acts):
            raise NotImplementedError
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        n = len(encoder_block_out_channels)
        ebt = (encoder_block_types,) * n if isinstance(encoder_block_types, str) else tuple(encoder_block_types)
        dbt = (decoder_block_types,) * n if isinstance(decoder_block_types, str) else tuple(decoder_block_types)
        self.encoder = Encoder(
            in_channels, latent_channels, attention_head_dim, ebt, encoder_block_out_channels, encoder_layers_per_block, encoder_qkv_multiscales,
            temb_channels=temb_channels,
        )
        self.decoder = Decoder(
            out_channels if out_channels is not None else in_channels,
            latent_channels,
            attention_head_dim,
            dbt,
            decoder_block_out_channels,
            decoder_layers_per_block,
            decoder_qkv_multiscales,
            act_fn=acts,
            upsample_block_type=upsample_block_type,
            temb_channels=temb_channels,
        )
        # models/DCAE.py:845-850: Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0) (no parameters) + TimestepEmbedding(256, temb_channels)
        self.timestep_embedder = TimestepEmbedding(256, temb_channels) if temb_channels is not None else None
        self.spatial_compression_ratio = 2 ** (n - 1)
        self.static_channels = static_channels
        self.use_slicing = False
        self.use_tiling = False

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @classmethod
    def from_config(cls, cfg: dict):
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def _embed_t(self, temb, embedded_t):
        """models/DCAE.py:982-984,1036-1038: a raw timestep goes through time_proj (sinusoid, 256) + timestep_embedder unless already embedded"""
        if temb is not None and not embedded_t:
            temb = self.timestep_embedder(get_timestep_embedding(temb, 256))
        return temb

    def encode(self, x, return_dict=True, temb=None, embedded_t=False, static_conditioning_tensor=None):
        temb = self._embed_t(temb, embedded_t)
        if static_conditioning_tensor is not None:
            x = torch.cat((x, static_conditioning_tensor), dim=1)
        if self.use_slicing and x.shape[0] > 1:
            raise NotImplementedError("Slicing is not supported for encoding.")
        z = self.encoder(x, temb)
        if not return_dict:
            return (z,)
        return SimpleNamespace(latent=z)

    def decode(self, z, return_dict=True, temb=None, embedded_t=False, return_static=False):
        temb = self._embed_t(temb, embedded_t)
        if self.use_slicing and z.size(0) > 1:
            raise NotImplementedError("Slicing is not supported for decoding.")
        y = self.decoder(z, temb)
        if not return_static and self.static_channels is not None:
            y = y[:, : -self.static_channels, :, :]
        if not return_dict:
            return (y,)
        return SimpleNamespace(sample=y)

    def forward(self, sample, return_dict=True, time_elapsed=None, static_conditioning_tensor=None, return_static=False):
        temb = self._embed_t(time_elapsed, False)  # models/DCAE.py:1067-1071
        z = self.encode(sample, return_dict=False, temb=temb, embedded_t=True, static_conditioning_tensor=static_conditioning_tensor)[0]
        y = self.decode(z, return_dict=False, temb=temb, embedded_t=True, return_static=return_static)[0]
        if not return_dict:
            return (y,)
        return SimpleNamespace(sample=y)
