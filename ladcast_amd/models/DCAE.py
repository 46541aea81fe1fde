"""``AutoencoderDC`` (models/DCAE.py:735-1087) on MI355X HIP kernels, shipped-config path
(configs/DC_AE_84_pretrain.yaml: ResBlock / EfficientViTBlock stages, rms_norm, pixel-(un)shuffle,
static conditioning channels, no timestep conditioning).

Same constructor kwargs / ``config`` / parameter names as the reference (a reference safetensors
file loads strictly) and the same ``encode`` / ``decode`` / ``forward`` signatures and outputs.
The module tree is a parameter container; all arithmetic is HIP.

MI355X-first layout: the reference is NCHW and pays a ``movedim`` round trip around every channel
RMSNorm, Linear and attention reshape (models/DCAE.py:222-260,317-319,371-373,729).  Here the
whole network runs NHWC (``[B*H*W, C]`` rows): 1x1 convs and Linears are plain GEMMs on the fp32
matrix cores, 3x3 sphere convs are implicit GEMMs with the pole/wrap gather done in the loader,
channel RMSNorm is a contiguous-row wave reduction, and NCHW appears only at the API boundary.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import hip
from .modeling_utils import ModelMixin
from .sphere_conv import (SphereConv2d, ceil4, pack_dense_weight, pack_dense_weight_bf16, pack_dense_weight_bf16x3, pack_dense_weight_f32ring,
                          pack_depthwise_weight)


class EncoderOutput(SimpleNamespace):
    def __getitem__(self, i):
        return (self.latent,)[i]


class DecoderOutput(SimpleNamespace):
    def __getitem__(self, i):
        return (self.sample,)[i]


# ---------------------------------------------------------------------------
# parameter containers (attribute names = reference)
# ---------------------------------------------------------------------------
class _RMSNormP(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        """torch arithmetic, reached ONLY from a user-supplied attention processor (`attn.processor = ...`): diffusers RMSNorm with bias over the
        last axis - fp32 statistics, x * rsqrt(mean(x^2) + eps) * weight + bias.  The built-in path runs ldc_rmsnorm_rows."""
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(var + self.eps) * self.weight + self.bias


class SanaMultiscaleAttentionProjection(nn.Module):
    def __init__(self, in_channels, num_attention_heads, kernel_size):
        super().__init__()
        ch = 3 * in_channels
        self.proj_in = SphereConv2d(ch, ch, kernel_size, padding=kernel_size // 2, groups=ch, bias=False)
        self.proj_out = nn.Conv2d(ch, ch, 1, 1, 0, groups=3 * num_attention_heads, bias=False)

    @torch.no_grad()
    def forward(self, x):
        """NCHW in, NCHW out (models/DCAE.py:87-93), reached ONLY from a user-supplied attention processor: the depthwise sphere conv through
        SphereConv2d.forward (HIP), the grouped 1 x 1 conv as one batched matrix product over the groups.  The built-in path runs
        ldc_sphere_dwconv_nhwc + ldc_grouped_conv1x1_nhwc on NHWC rows."""
        h = self.proj_in(x)
        B, C, H, W = h.shape
        G = self.proj_out.groups
        w = self.proj_out.weight.reshape(G, C // G, C // G)  # [group][out][in]
        return torch.einsum("goi,bgip->bgop", w, h.reshape(B, G, C // G, H * W)).reshape(B, C, H, W)


class SanaMultiscaleAttnProcessor2_0:
    """marker of the built-in (fused) linear-attention path, models/DCAE.py:200-267; never called"""


class TimestepEmbedding(nn.Module):
    """parameter container of diffusers' TimestepEmbedding(256, temb_channels): linear_2(SiLU(linear_1(sinusoid(t))))"""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class AdaLayerNormZeroSingle4Sana(nn.Module):
    """parameter container of models/DCAE.py:36-64: Linear(C -> 3C) producing (shift, scale, gate); the LayerNorm has no parameters"""

    def __init__(self, embedding_dim, bias=True):
        super().__init__()
        self.linear = nn.Linear(embedding_dim, 3 * embedding_dim, bias=bias)


class SanaMultiscaleLinearAttention(nn.Module):
    def __init__(self, in_channels, out_channels, attention_head_dim=32, mult=1.0, kernel_sizes=(5,), eps=1e-15, temb_channels=None):
        super().__init__()
        if temb_channels is not None:  # models/DCAE.py:147-153: the block's projection of relu(temb) + AdaLayerNormZeroSingle4Sana (:36-64)
            self.time_emb_porj = nn.Linear(temb_channels, out_channels)
            self.norm_in = AdaLayerNormZeroSingle4Sana(out_channels)
        else:
            self.time_emb_porj, self.norm_in = None, None
        if attention_head_dim != 32:
            raise NotImplementedError("the linear-attention kernel is built for attention_head_dim 32")
        self.eps = eps
        self.attention_head_dim = attention_head_dim
        self.heads = int(in_channels // attention_head_dim * mult)
        inner = self.heads * attention_head_dim
        self.inner = inner
        self.to_q = nn.Linear(in_channels, inner, bias=False)
        self.to_k = nn.Linear(in_channels, inner, bias=False)
        self.to_v = nn.Linear(in_channels, inner, bias=False)
        self.to_qkv_multiscale = nn.ModuleList([SanaMultiscaleAttentionProjection(inner, self.heads, ks) for ks in kernel_sizes])
        self.to_out = nn.Linear(inner * (1 + len(kernel_sizes)), out_channels, bias=False)
        self.norm_out = _RMSNormP(out_channels, 1e-5)  # diffusers get_normalization("rms_norm") default eps
        # what a processor written against the reference's module reads from it (models/DCAE.py:119-120,143,205-267)
        self.nonlinearity = nn.ReLU()
        self.norm_type = "rms_norm"
        self.residual_connection = True
        self.processor = SanaMultiscaleAttnProcessor2_0()

    # models/DCAE.py:156,205-210: the reference keeps its processor in a plain attribute and calls `self.processor(self, hidden_states, gate=gate_msa)`.
    # An instance of the built-in class (the default) is a MARKER: the block runs fused (conv GEMMs + ldc_relu_linear_attn_nhwc_fmt) and the
    # processor is never called.  Any other object is a FOREIGN processor: AutoencoderDC._evit CALLS it with the reference's protocol on torch
    # tensors (NCHW, fp32 mode, eager launches only) - it is never ignored.
    @property
    def foreign_processor(self):
        p = self.processor
        return None if isinstance(p, SanaMultiscaleAttnProcessor2_0) else p

    def apply_linear_attention(self, query, key, value):
        """torch arithmetic for user-supplied processors (models/DCAE.py:158-175): ReLU linear attention in fp32, the all-ones row appended to
        `value` carries the normaliser"""
        value = torch.nn.functional.pad(value, (0, 0, 0, 1), mode="constant", value=1)
        scores = torch.matmul(value.to(torch.float32), key.transpose(-1, -2).to(torch.float32))
        hs = torch.matmul(scores, query.to(torch.float32))
        return hs[:, :, :-1] / (hs[:, :, -1:] + self.eps)

    def apply_quadratic_attention(self, query, key, value):
        """torch arithmetic for user-supplied processors (models/DCAE.py:177-186)"""
        scores = torch.matmul(key.transpose(-1, -2), query).to(torch.float32)
        scores = scores / (torch.sum(scores, dim=2, keepdim=True) + self.eps)
        return torch.matmul(value, scores)


class GLUMBConv(nn.Module):
    def __init__(self, in_channels, out_channels, expand_ratio=4):
        super().__init__()
        hid = int(expand_ratio * in_channels)
        self.conv_inverted = nn.Conv2d(in_channels, hid * 2, 1, 1, 0)
        self.conv_depth = SphereConv2d(hid * 2, hid * 2, 3, 1, 1, groups=hid * 2)
        self.conv_point = nn.Conv2d(hid, out_channels, 1, 1, 0, bias=False)
        self.norm = _RMSNormP(out_channels, 1e-7)


class ResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, act_fn="silu", temb_channels=None):
        super().__init__()
        self.time_emb_porj = nn.Linear(temb_channels, 2 * out_channels) if temb_channels is not None else None  # models/DCAE.py:351-354
        self.act = {"silu": hip.ACT_SILU, "relu": hip.ACT_RELU}[act_fn]
        self.conv1 = SphereConv2d(in_channels, in_channels, 3, 1, 1)
        self.conv2 = SphereConv2d(in_channels, out_channels, 3, 1, 1, bias=False)
        self.norm = _RMSNormP(out_channels, 1e-5)


class EfficientViTBlock(nn.Module):
    def __init__(self, in_channels, attention_head_dim=32, qkv_multiscales=(5,), temb_channels=None):
        super().__init__()
        self.attn = SanaMultiscaleLinearAttention(in_channels, in_channels, attention_head_dim=attention_head_dim, kernel_sizes=qkv_multiscales,
                                                  temb_channels=temb_channels)
        self.conv_out = GLUMBConv(in_channels, in_channels)


class DCDownBlock2d(nn.Module):
    """models/DCAE.py:447-490 (pixel_unshuffle form); shortcut=False: the encoder's conv_in when layers_per_block[0] == 0 (:571-579)"""

    def __init__(self, in_channels, out_channels, shortcut=True):
        super().__init__()
        assert out_channels % 4 == 0
        self.in_channels, self.out_channels, self.shortcut = in_channels, out_channels, shortcut
        self.conv = SphereConv2d(in_channels, out_channels // 4, 3, 1, 1)


class DCUpBlock2d(nn.Module):
    """models/DCAE.py:493-536: conv at 4x the width + pixel_shuffle, or (interpolate, `upsample_block_type="interpolate"`) nearest x2 + conv;
    shortcut=False: the decoder's conv_out when layers_per_block[0] == 0 (:706-712)"""

    def __init__(self, in_channels, out_channels, interpolate=False, shortcut=True):
        super().__init__()
        self.in_channels, self.out_channels, self.interpolate, self.shortcut = in_channels, out_channels, interpolate, shortcut
        self.conv = SphereConv2d(in_channels, out_channels if interpolate else out_channels * 4, 3, 1, 1)


def _get_block(block_type, ch, head_dim, act_fn, multiscales, temb_channels=None):
    if block_type == "ResBlock":
        return ResBlock(ch, ch, act_fn, temb_channels=temb_channels)
    if block_type == "EfficientViTBlock":
        return EfficientViTBlock(ch, head_dim, tuple(multiscales), temb_channels=temb_channels)
    raise ValueError(f"Block with {block_type=} is not supported.")


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, head_dim, block_type, block_out_channels, layers_per_block, qkv_multiscales, temb_channels=None):
        super().__init__()
        n = len(block_out_channels)
        if layers_per_block[0] > 0:
            self.conv_in = SphereConv2d(in_channels, block_out_channels[0], 3, 1, 1)
        else:  # models/DCAE.py:571-579 (the DC-AE family's f64 / f128 form): no stage at full resolution, conv_in = a down block without shortcut
            self.conv_in = DCDownBlock2d(in_channels, block_out_channels[1], shortcut=False)
        self.down_blocks = nn.ModuleList()
        for i, (ch, nl) in enumerate(zip(block_out_channels, layers_per_block)):
            for _ in range(nl):
                self.down_blocks.append(_get_block(block_type[i], ch, head_dim, "silu", qkv_multiscales[i], temb_channels))
            if i < n - 1 and nl > 0:
                self.down_blocks.append(DCDownBlock2d(ch, block_out_channels[i + 1]))
        self.conv_out = SphereConv2d(block_out_channels[-1], latent_channels, 3, 1, 1)


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, head_dim, block_type, block_out_channels, layers_per_block, qkv_multiscales, act_fn="silu",
                 upsample_block_type="pixel_shuffle", temb_channels=None):
        super().__init__()
        n = len(block_out_channels)
        self.conv_in = SphereConv2d(latent_channels, block_out_channels[-1], 3, 1, 1)
        self.up_blocks = nn.ModuleList()
        for i, (ch, nl) in reversed(list(enumerate(zip(block_out_channels, layers_per_block)))):
            if i < n - 1 and nl > 0:
                self.up_blocks.append(DCUpBlock2d(block_out_channels[i + 1], ch, interpolate=upsample_block_type == "interpolate"))
            for _ in range(nl):
                self.up_blocks.append(_get_block(block_type[i], ch, head_dim, act_fn if isinstance(act_fn, str) else act_fn[i], qkv_multiscales[i], temb_channels))
        ch0 = block_out_channels[0] if layers_per_block[0] > 0 else block_out_channels[1]  # models/DCAE.py:696-712
        self.norm_out = _RMSNormP(ch0, 1e-7)
        if layers_per_block[0] > 0:
            self.conv_out = SphereConv2d(ch0, out_channels, 3, 1, 1)
        else:
            self.conv_out = DCUpBlock2d(ch0, out_channels, interpolate=upsample_block_type == "interpolate", shortcut=False)


# ---------------------------------------------------------------------------
class AutoencoderDC(ModelMixin):
    _supports_gradient_checkpointing = False
    GRAPH_MAX_FRAMES = 8  # graph mode covers the launch-bound batch sizes; a graph keeps its activations (0.3 GB per 120x240 frame) alive

    def __init__(
        self,
        in_channels: int = 3,
        out_channels: Optional[int] = None,
        temb_channels: Optional[int] = None,
        latent_channels: int = 32,
        attention_head_dim: int = 32,
        encoder_block_types: Union[str, Tuple[str]] = "ResBlock",
        decoder_block_types: Union[str, Tuple[str]] = "ResBlock",
        encoder_block_out_channels: Tuple[int, ...] = (128, 256, 512, 512, 1024, 1024),
        decoder_block_out_channels: Tuple[int, ...] = (128, 256, 512, 512, 1024, 1024),
        encoder_layers_per_block: Tuple[int] = (2, 2, 2, 3, 3, 3),
        decoder_layers_per_block: Tuple[int] = (3, 3, 3, 3, 3, 3),
        encoder_qkv_multiscales=((), (), (), (5,), (5,), (5,)),
        decoder_qkv_multiscales=((), (), (), (5,), (5,), (5,)),
        upsample_block_type: str = "pixel_shuffle",
        downsample_block_type: str = "pixel_unshuffle",
        decoder_norm_types: Union[str, Tuple[str]] = "rms_norm",
        decoder_act_fns: Union[str, Tuple[str]] = "silu",
        scaling_factor: float = 1.0,
        static_channels: int = 0,
    ) -> None:
        super().__init__()
        self.register_to_config(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        if upsample_block_type not in ("pixel_shuffle", "interpolate") or downsample_block_type != "pixel_unshuffle":
            # ("conv" down-sampling builds a stride-2 SphereConv2d, which the reference's SphereConv2d itself refuses: it asserts stride 1)
            raise NotImplementedError("sampling: pixel_unshuffle down (configs/DC_AE_84_pretrain.yaml:45-46), pixel_shuffle | interpolate up")
        acts = (decoder_act_fns,) * len(decoder_block_out_channels) if isinstance(decoder_act_fns, str) else tuple(decoder_act_fns)  # models/DCAE.py:663-664
        if decoder_norm_types != "rms_norm" or any(a not in ("silu", "relu") for a in acts):
            raise NotImplementedError("decoder ResBlocks: rms_norm with silu or relu (per stage); relu6 / gelu / mish and batch_norm are not built")
        n = len(encoder_block_out_channels)
        ebt = (encoder_block_types,) * n if isinstance(encoder_block_types, str) else tuple(encoder_block_types)
        dbt = (decoder_block_types,) * n if isinstance(decoder_block_types, str) else tuple(decoder_block_types)
        self.encoder = Encoder(in_channels, latent_channels, attention_head_dim, ebt, encoder_block_out_channels, encoder_layers_per_block, encoder_qkv_multiscales,
                               temb_channels=temb_channels)
        # models/DCAE.py:845-850: Timesteps(256) (a sinusoid, no parameters: ldc_timestep_embedding) + TimestepEmbedding(256, temb_channels)
        self.timestep_embedder = TimestepEmbedding(256, temb_channels) if temb_channels is not None else None
        self.decoder = Decoder(out_channels if out_channels is not None else in_channels, latent_channels, attention_head_dim, dbt,
                               decoder_block_out_channels, decoder_layers_per_block, decoder_qkv_multiscales, act_fn=acts, upsample_block_type=upsample_block_type,
                               temb_channels=temb_channels)
        self.spatial_compression_ratio = 2 ** (n - 1)
        self.temporal_compression_ratio = 1
        self.use_slicing = False
        self.use_tiling = False
        self.static_channels = static_channels
        self.requires_grad_(False)
        self._plan = None
        self._plan_gen = 0  # bumped on every plan rebuild: key of the graphs that captured pointers into it
        self.gemm_precision = "fp32"
        self.use_hip_graph = False
        self._graphs = {}
        self._capture_stream = None

    def enable_hip_graph(self, flag: bool = True):
        """Replay one captured hipGraph per (encode | decode, input shape) instead of ~400 launches from Python.  Same kernels and
        arguments, bit-identical results.  Measured on one MI355X it buys nothing on an idle host (one frame: 5.5 ms either way - the
        ~400 small kernels, not their launches, are the time); it takes the host out of the loop when 8 ranks share one box."""
        if flag and any(m.foreign_processor is not None for m in self.modules() if isinstance(m, SanaMultiscaleLinearAttention)):
            raise NotImplementedError("a user-supplied DCAE attention processor runs eagerly (torch tensors): it cannot be captured into a hipGraph")
        self.use_hip_graph = bool(flag)
        if not flag:
            self._graphs = {}
        return self

    def _graphed(self, key, fn, inputs):
        """fn(*static inputs) -> output tensor, kernel launches only on the current stream"""
        key = key + (self._plan_gen, self.gemm_precision)  # generation counter, not id(): a rebuilt dict may reuse a freed id
        ent = self._graphs.get(key)
        if ent is None:
            dev = self.device
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream(device=dev)
            side = self._capture_stream
            st = [torch.empty_like(t) for t in inputs]
            for a, b in zip(st, inputs):
                a.copy_(b)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up on the capture stream: per-stream workspaces are created here
                fn(*st)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                out = fn(*st)
            ent = (graph, st, out)
            self._graphs[key] = ent
        graph, st, out = ent
        for a, b in zip(st, inputs):
            a.copy_(b)
        graph.replay()
        return out.clone()

    def set_gemm_precision(self, precision: str):
        """'fp32' (default): every conv on the exact-fp32 matrix cores; 'bf16x3': the dense 3x3 SphereConv2d layers, the 1x1 convs and
        the Linears (99 % of the FLOPs) as split-bf16 implicit GEMMs (hi*hi + hi*lo + lo*hi, fp32 accumulate; ~4e-6 per layer) on
        activations their producers write pre-split; depthwise / grouped convs, norms, residual stream and the linear attention
        stay fp32.  'bf16': the mixed-precision mode of BASELINE configs[4] - the same convs with ONE bf16 MFMA per product on plain
        bf16 operand rows (what torch.autocast(bfloat16) does to a conv's operands; outputs, accumulation and the reference's fp32
        islands, models/DCAE.py:162,180, stay fp32); stated tolerances (ladcast_amd/precision.py: measured x 2) 2e-2 per encode, 1.2e-2 per decode against the fp32 oracle; closer to it than the oracle run under the reference's own autocast recipe (tests/test_gpu_dcae.py)."""
        if precision not in ("fp32", "bf16x3", "bf16"):
            raise ValueError("gemm precision must be 'fp32', 'bf16x3' or 'bf16'")
        if precision != self.gemm_precision:
            self.gemm_precision = precision
            self._plan = None
            self._graphs = {}  # captured graphs hold raw pointers into the old packed weights
        return self

    def enable_tiling(self, *a, **k):
        self.use_tiling = True

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def _apply(self, fn, *a, **k):
        self._plan = None
        self._graphs = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plan = None
        self._graphs = {}
        return super().load_state_dict(*a, **k)

    # -- plan: repacked weights (NHWC / tap-major) ------------------------------------------------
    def _build_plan(self):
        self._upcast_to_fp32()  # a bf16 / fp16 model loads: parameters are up-cast once, with a warning
        if not next(self.parameters()).is_cuda:
            raise RuntimeError("AutoencoderDC must live on a HIP device (no CPU fallback)")
        plan = {}
        split = self.gemm_precision != "fp32"
        # operand format of the mode; round 4: the exact-fp32 mode runs its dense convs / 1x1 convs / Linears on the same LDS-DMA ring kernel
        # (fp32 operand rows, fp32 matrix instruction) - every weight is the tap-major matrix with the taps padded to 32 * 2^j channels
        pack = {"bf16": pack_dense_weight_bf16, "bf16x3": pack_dense_weight_bf16x3, "fp32": pack_dense_weight_f32ring}[self.gemm_precision]
        for mod in self.modules():
            if isinstance(mod, SphereConv2d):
                if mod.groups == 1:
                    k3 = mod.kernel_size[0] == 3
                    plan[id(mod)] = pack(mod.weight) if k3 else pack_dense_weight(mod.weight)
                else:
                    plan[id(mod)] = pack_depthwise_weight(mod.weight)
            elif isinstance(mod, nn.Conv2d):  # 1x1 convs
                w = mod.weight.reshape(mod.weight.shape[0], -1).contiguous()
                plan[id(mod)] = pack(w[:, :, None, None]) if mod.groups == 1 else w
            elif isinstance(mod, SanaMultiscaleLinearAttention):
                w = torch.cat([mod.to_q.weight, mod.to_k.weight, mod.to_v.weight], dim=0).contiguous()
                plan[id(mod)] = pack(w[:, :, None, None])
                plan[id(mod.to_out)] = pack(mod.to_out.weight[:, :, None, None])
        self._plan = plan
        self._plan_gen += 1

    # -- NHWC building blocks ---------------------------------------------------------------------
    # A stream tensor is a pair (x32, xs): fp32 rows [B*H*W, C] (residual adds, depthwise convs, norms) and - in the `bf16x3` mode -
    # the same values as SPLIT rows [B*H*W, C rounded up to 8] (hip.FMT_SPLIT: what the dense convs / 1x1 convs / Linears read, written
    # once by whoever produces the tensor: conv / norm / pixel-shuffle / attention / GLU epilogues).  fp32 mode: xs is None.
    @property
    def _split(self):
        return self.gemm_precision != "fp32"

    @property
    def _fmt(self):  # format of the operand rows the dense convs read
        return hip.FMT_BF16 if self.gemm_precision == "bf16" else hip.FMT_SPLIT

    @staticmethod
    def _c8(c):
        return -(-c // 8) * 8

    def _srows(self, M, C, dev):
        return torch.empty(M, self._c8(C), device=dev, dtype=torch.float32) if self._split else None

    def _conv(self, x, B, H, W, conv, act=hip.ACT_NONE, R=None, out_split=False):
        """dense 3x3 SphereConv2d of the stream tensor x = (x32, xs); returns fp32 rows, or (out_split, bf16x3 mode) split rows"""
        x32, xs = x
        cin_p = ceil4(conv.in_channels)
        cout = conv.out_channels
        if self._split:
            ldy = self._c8(cout) if out_split else cout
            y = torch.empty(B * H * W, ldy, device=xs.device, dtype=torch.float32)
            hip.sphere_conv_nhwc_split(xs, self._plan[id(conv)], y, B=B, H=H, W=W, cin=cin_p, ldx=xs.shape[1], cout=cout, ldy=ldy,
                                       bias=conv.bias, R=R, ldr=cout if R is not None else 0, ksize=3, act=act, in_fmt=self._fmt,
                                       out_fmt=self._fmt if out_split else hip.FMT_F32)
            return y
        y = torch.empty(B * H * W, cout, device=x32.device, dtype=torch.float32)
        hip.sphere_conv_nhwc_split(x32, self._plan[id(conv)], y, B=B, H=H, W=W, cin=cin_p, ldx=x32.shape[1], cout=cout, ldy=cout, bias=conv.bias, R=R,
                                   ldr=cout if R is not None else 0, ksize=3, act=act, in_fmt=hip.FMT_F32, out_fmt=hip.FMT_F32)
        return y

    def _mm(self, x, key, w_fp32, y, B, H, W, N, K, ldc=None, bias=None, act=hip.ACT_NONE):
        """pointwise conv / Linear over the B*H*W pixel rows of x = (x32, xs) into fp32 rows y: exact-fp32 GEMM, or (bf16x3 mode) the
        pre-split conv kernel with ksize 1"""
        x32, xs = x
        if self._split:
            hip.sphere_conv_nhwc_split(xs, self._plan[key], y, B=B, H=H, W=W, cin=K, ldx=xs.shape[1], cout=N, ldy=ldc, bias=bias, ksize=1,
                                       act=act, in_fmt=self._fmt)
        else:  # exact fp32: the same conv entry with ksize 1 on fp32 rows (K need not be a multiple of 32: the taps are padded with zeros)
            hip.sphere_conv_nhwc_split(x32, self._plan[key], y, B=B, H=H, W=W, cin=K, ldx=x32.shape[1], cout=N, ldy=ldc if ldc is not None else N,
                                       bias=bias, ksize=1, act=act, in_fmt=hip.FMT_F32, out_fmt=hip.FMT_F32)

    def _norm(self, u, norm, resid, M, C, act=hip.ACT_NONE, want32=True):
        """RMSNorm rows (+ residual, activation) -> stream tensor"""
        y32 = torch.empty(M, C, device=u.device, dtype=torch.float32) if (want32 or not self._split) else None
        ys = self._srows(M, C, u.device)
        hip.rmsnorm_rows(u, norm.weight, y32, rows=M, C=C, eps=norm.eps, b=norm.bias, resid=resid, act=act, ys=ys, fmt=self._fmt)
        return y32, ys

    def _resblock(self, blk, x, B, H, W, temb=None):
        if blk.time_emb_porj is not None:
            # timestep-conditioned ResBlock (models/DCAE.py:361-365): act(conv1(x)) * scale + shift, (scale | shift) = Linear(act(temb)) per frame
            if temb is None:
                raise ValueError("this autoencoder was built with temb_channels: encode / decode need `temb`")
            M, C = B * H * W, blk.conv1.out_channels
            t32 = self._conv(x, B, H, W, blk.conv1, act=blk.act)
            mod = torch.empty(B, 2 * C, device=t32.device, dtype=torch.float32)
            hip.linear_small(temb, blk.time_emb_porj.weight, mod, rows=B, N=2 * C, K=temb.shape[1], bias=blk.time_emb_porj.bias, act_in=blk.act)
            for b in range(B):  # per frame: x * scale_c + shift_c over its H W pixel rows, in place (two roundings, as the reference's mul + add)
                rows = t32[b * H * W : (b + 1) * H * W]
                hip.chan_affine(rows, rows, mod[b, C:], mod[b, :C], 1.0, outer=H * W, C=C, inner=1, inverse=True)
            u = self._conv(self._stream(t32, M, C), B, H, W, blk.conv2)
            return self._norm(u, blk.norm, x[0], M, blk.conv2.out_channels)
        t = self._conv(x, B, H, W, blk.conv1, act=blk.act, out_split=True)  # only conv2 reads it
        t = (None, t) if self._split else (t, None)
        u = self._conv(t, B, H, W, blk.conv2)
        return self._norm(u, blk.norm, x[0], B * H * W, blk.conv2.out_channels)

    def _evit(self, blk, x, B, H, W, temb=None):
        M = B * H * W
        at = blk.attn
        C, inner, heads = at.to_q.in_features, at.inner, at.heads
        dev = x[0].device
        split = self._split
        gate = None
        if at.norm_in is not None:
            # timestep-conditioned attention (models/DCAE.py:36-64,193-198): emb = Linear(relu(temb)); (shift, scale, gate) = Linear(SiLU(emb));
            # x <- LayerNorm_channels(x, eps 1e-15) (1 + scale) + shift - which is also the block's residual (:217) - and gate scales to_out's output
            if temb is None:
                raise ValueError("this autoencoder was built with temb_channels: encode / decode need `temb`")
            e1 = torch.empty(B, C, device=dev, dtype=torch.float32)
            hip.linear_small(temb, at.time_emb_porj.weight, e1, rows=B, N=C, K=temb.shape[1], bias=at.time_emb_porj.bias, act_in=hip.ACT_RELU)
            mod = torch.empty(B, 3 * C, device=dev, dtype=torch.float32)
            hip.linear_small(e1, at.norm_in.linear.weight, mod, rows=B, N=3 * C, K=C, bias=at.norm_in.linear.bias, act_in=hip.ACT_SILU)
            xn = torch.empty(M, C, device=dev, dtype=torch.float32)
            ldx = x[0].shape[1]
            hip.layernorm_mod(x[0], xn, B=B, rows=H * W, D=C, ldx=ldx, x_bs=H * W * ldx, ldy=C, y_bs=H * W * C, scale=mod[:, C:], shift=mod, mod_bs=3 * C, mode=0,
                              eps=1e-15)
            x = self._stream(xn, M, C)
            gate = mod[:, 2 * C :]
        if at.foreign_processor is not None:
            # a user-supplied processor (models/DCAE.py:156,205-210): CALLED with the reference's protocol `(attn, hidden_states NCHW, gate=gate_msa)`;
            # it returns the block's attention output (norm_out and the residual applied, as the reference's processor does).  torch tensors,
            # exact-fp32 mode, eager launches only.
            if split:
                raise NotImplementedError("a user-supplied DCAE attention processor runs in the exact-fp32 mode only (set_gemm_precision('fp32')): "
                                          "the split modes keep activations as operand rows the processor cannot read")
            if torch.cuda.is_current_stream_capturing():
                raise NotImplementedError("a user-supplied DCAE attention processor is not graph-capturable: enable_hip_graph(False)")
            xin = x[0][:, :C].reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
            g4 = None if gate is None else gate.reshape(B, C, 1, 1)  # AdaLayerNormZeroSingle4Sana returns gate_msa[:, :, None, None] (:64)
            out = at.foreign_processor(at, xin, gate=g4)
            if tuple(out.shape) != (B, C, H, W):
                raise ValueError(f"the attention processor returned {tuple(out.shape)}, expected {(B, C, H, W)}")
            y = self._stream(out.to(torch.float32).permute(0, 2, 3, 1).reshape(M, C).contiguous(), M, C)
            return self._glumb(blk, y, B, H, W)
        n_ms = len(at.to_qkv_multiscale)
        wide = 3 * inner * (1 + n_ms)
        qkv = torch.empty(M, wide, device=dev, dtype=torch.float32)
        self._mm(x, id(at), self._plan[id(at)], qkv, B, H, W, N=3 * inner, K=C, ldc=wide)
        for s, ms in enumerate(at.to_qkv_multiscale):
            dw = torch.empty(M, 3 * inner, device=dev, dtype=torch.float32)
            hip.sphere_dwconv_nhwc(qkv, self._plan[id(ms.proj_in)], dw, B=B, H=H, W=W, C=3 * inner, ldx=wide, ksize=ms.proj_in.kernel_size[0])
            hip.grouped_conv1x1_nhwc(dw, self._plan[id(ms.proj_out)], qkv[:, 3 * inner * (1 + s) :], M=M, groups=3 * heads, ldx=3 * inner, ldy=wide)
        groups = wide // 96  # consecutive 96-channel groups of the concat, split (q, k, v) -- models/DCAE.py:239-243 (Q8)
        att = torch.empty(M, groups * 32, device=dev, dtype=torch.float32)  # only to_out reads it: split rows in the bf16x3 mode
        hip.relu_linear_attn_nhwc(qkv, att, B=B, P=H * W, groups=groups, ldq=wide, ldy=groups * 32, eps=at.eps,
                                  out_fmt=self._fmt if split else hip.FMT_F32)
        o = torch.empty(M, C, device=dev, dtype=torch.float32)
        self._mm((None, att) if split else (att, None), id(at.to_out), at.to_out.weight, o, B, H, W, N=C, K=groups * 32)
        if gate is not None:  # models/DCAE.py:256-257: hidden_states * gate, before norm_out
            og = torch.empty_like(o)
            hip.gate_residual(torch.zeros_like(o), o, gate, og, B=B, rows=H * W, D=C, ld_res=C, res_bs=H * W * C, ld_y=C, y_bs=H * W * C, gate_bs=3 * C)
            o = og
        y = self._norm(o, at.norm_out, x[0], M, C)
        return self._glumb(blk, y, B, H, W)

    def _glumb(self, blk, y, B, H, W):
        """GLUMBConv, models/DCAE.py:304-324"""
        M, C, dev, split = B * H * W, blk.attn.to_q.in_features, y[0].device, self._split
        g = blk.conv_out
        hid2 = g.conv_inverted.out_channels
        h1 = torch.empty(M, hid2, device=dev, dtype=torch.float32)
        self._mm(y, id(g.conv_inverted), self._plan[id(g.conv_inverted)], h1, B, H, W, N=hid2, K=C, bias=g.conv_inverted.bias, act=hip.ACT_SILU)
        h2 = torch.empty(M, hid2 // 2, device=dev, dtype=torch.float32)  # only conv_point reads it
        hip.sphere_dwconv_nhwc(h1, self._plan[id(g.conv_depth)], h2, B=B, H=H, W=W, C=hid2, bias=g.conv_depth.bias, ksize=3, glu=True,
                               out_fmt=self._fmt if split else hip.FMT_F32)
        h3 = torch.empty(M, C, device=dev, dtype=torch.float32)
        self._mm((None, h2) if split else (h2, None), id(g.conv_point), self._plan[id(g.conv_point)], h3, B, H, W, N=C, K=hid2 // 2)
        return self._norm(h3, g.norm, y[0], M, C)

    def _run_blocks(self, blocks, x, B, H, W, temb=None):
        for blk in blocks:
            if isinstance(blk, ResBlock):
                x = self._resblock(blk, x, B, H, W, temb)
            elif isinstance(blk, EfficientViTBlock):
                x = self._evit(blk, x, B, H, W, temb)
            elif isinstance(blk, DCDownBlock2d):
                cv = self._conv(x, B, H, W, blk.conv)
                M2 = B * (H // 2) * (W // 2)
                y = torch.empty(M2, blk.out_channels, device=cv.device, dtype=torch.float32)
                ys = self._srows(M2, blk.out_channels, cv.device)
                hip.pixel_unshuffle_shortcut(cv, x[0] if blk.shortcut else None, y, B=B, H2=H // 2, W2=W // 2, cout=blk.out_channels, cin=blk.in_channels, ys=ys,
                                             fmt=self._fmt)
                x, H, W = (y, ys), H // 2, W // 2
            elif isinstance(blk, DCUpBlock2d) and blk.interpolate:
                # nearest x2 (operand rows for the conv) -> conv at the output width with the shortcut as its residual operand (models/DCAE.py:519-532)
                M4, ci, co = B * 4 * H * W, blk.in_channels, blk.out_channels
                up32 = torch.empty(M4, ci, device=x[0].device, dtype=torch.float32) if not self._split else None
                ups = self._srows(M4, ci, x[0].device)
                hip.upsample_nearest2x_rows(x[0], up32, B=B, H=H, W=W, C=ci, ldx=x[0].shape[1], ys=ups, fmt=self._fmt)
                sc = torch.empty(M4, co, device=x[0].device, dtype=torch.float32)
                hip.pixel_shuffle_shortcut(None, x[0], sc, B=B, H=H, W=W, cout=co, cin=ci)
                y = self._conv((up32, ups), B, 2 * H, 2 * W, blk.conv, R=sc)
                x, H, W = self._stream(y, M4, co), 2 * H, 2 * W
            elif isinstance(blk, DCUpBlock2d):
                cv = self._conv(x, B, H, W, blk.conv)
                y = torch.empty(B * 4 * H * W, blk.out_channels, device=cv.device, dtype=torch.float32)
                ys = self._srows(B * 4 * H * W, blk.out_channels, cv.device)
                hip.pixel_shuffle_shortcut(cv, x[0], y, B=B, H=H, W=W, cout=blk.out_channels, cin=blk.in_channels, ys=ys, fmt=self._fmt)
                x, H, W = (y, ys), 2 * H, 2 * W
            else:
                raise TypeError(type(blk))
        return x, H, W

    def _stream(self, x32, M, C):
        """fp32 rows -> stream tensor (bf16x3 mode: + the split copy)"""
        xs = self._srows(M, C, x32.device)
        if xs is not None:
            hip.split_rows(x32, xs, rows=M, C=C, ldx=x32.shape[1], lds=xs.shape[1], fmt=self._fmt)
        return x32, xs

    def _embed_t_launch(self, t, B):
        """raw timesteps (B,) or (1,) -> temb [B, temb_channels]: time_proj (the 256-wide sinusoid, fused as the GEMV's input transform) +
        timestep_embedder (models/DCAE.py:982-984,845-850); launches only"""
        te = self.timestep_embedder
        T = te.linear_2.out_features
        n = t.numel()
        mid, out = torch.empty(n, T, device=t.device, dtype=torch.float32), torch.empty(n, T, device=t.device, dtype=torch.float32)
        hip.linear_small(t.reshape(-1), te.linear_1.weight, mid, rows=n, N=T, K=256, bias=te.linear_1.bias, act_in=hip.ACT_IN_TIMESTEP_SINCOS, act_out=hip.ACT_SILU)
        hip.linear_small(mid, te.linear_2.weight, out, rows=n, N=T, K=T, bias=te.linear_2.bias)
        return out if n == B else out.expand(B, T).contiguous()

    def _temb_arg(self, temb, embedded_t, B, dev):
        """the `temb` argument of encode / decode as a device tensor: (B, temb_channels) when already embedded, else raw timesteps"""
        if temb is None:
            return None
        if self.timestep_embedder is None:
            raise ValueError("temb was given, but this autoencoder was built without temb_channels")
        t = temb.to(device=dev, dtype=torch.float32).contiguous()
        if embedded_t and t.shape[0] != B:
            t = t.expand(B, -1).contiguous()
        return t

    def _encode_launch(self, x, st=None, temb=None, embedded_t=True):
        """kernel launches only (capturable): NCHW fp32 device tensors -> latent (B, lc, H/8, W/8)"""
        enc = self.encoder
        dev = x.device
        B, C, H, W = x.shape
        cs = st.shape[1] if st is not None else 0
        cp = ceil4(C + cs)
        tok = torch.empty(B * H * W, cp, device=dev, dtype=torch.float32)
        # torch.cat((x, static), dim=1) + NCHW->NHWC in one pass each (models/DCAE.py:988-989)
        hip.chan_to_token(x, tok, B=B, C=C, N=H * W, ldo=cp, fill_cols=C if cs else cp)
        if cs:
            hip.chan_to_token(st, tok[:, C:], B=B, C=cs, N=H * W, ldo=cp, fill_cols=cp - C)
        if temb is not None and not embedded_t:
            temb = self._embed_t_launch(temb, B)
        if isinstance(enc.conv_in, DCDownBlock2d):  # layers_per_block[0] == 0: conv + pixel_unshuffle, no shortcut (models/DCAE.py:571-579)
            h, H, W = self._run_blocks([enc.conv_in], self._stream(tok, B * H * W, cp), B, H, W)
        else:
            h = self._conv(self._stream(tok, B * H * W, cp), B, H, W, enc.conv_in)
            h = self._stream(h, B * H * W, enc.conv_in.out_channels)
        h, H, W = self._run_blocks(enc.down_blocks, h, B, H, W, temb)
        lc = enc.conv_out.out_channels
        sc = torch.empty(B * H * W, lc, device=dev, dtype=torch.float32)
        hip.chan_regroup(h[0], sc, M=B * H * W, cin=enc.conv_out.in_channels, cout=lc)  # out shortcut, :624-627
        z = self._conv(h, B, H, W, enc.conv_out, R=sc)
        out = torch.empty(B, lc, H, W, device=dev, dtype=torch.float32)
        hip.token_to_chan(z, out, B=B, C=lc, N=H * W, ldi=lc)
        return out

    def _decode_launch(self, z, return_static=False, temb=None, embedded_t=True):
        """kernel launches only (capturable): latent (B, C, h, w) -> fields (B, keep, 8h, 8w)"""
        dec = self.decoder
        dev = z.device
        B, C, H, W = z.shape
        tok = torch.empty(B * H * W, C, device=dev, dtype=torch.float32)
        hip.chan_to_token(z, tok, B=B, C=C, N=H * W, ldo=C)
        c0 = dec.conv_in.out_channels
        rep = torch.empty(B * H * W, c0, device=dev, dtype=torch.float32)
        hip.chan_regroup(tok, rep, M=B * H * W, cin=C, cout=c0)
        if temb is not None and not embedded_t:
            temb = self._embed_t_launch(temb, B)
        h = self._conv(self._stream(tok, B * H * W, C), B, H, W, dec.conv_in, R=rep)  # in shortcut = repeat_interleave, :720-722
        h, H, W = self._run_blocks(dec.up_blocks, self._stream(h, B * H * W, c0), B, H, W, temb)
        up_out = isinstance(dec.conv_out, DCUpBlock2d)  # layers_per_block[0] == 0: conv_out = an up block without shortcut (models/DCAE.py:706-712)
        n = self._norm(h[0], dec.norm_out, None, B * H * W, dec.norm_out.weight.numel(), act=hip.ACT_RELU,
                       want32=up_out and dec.conv_out.interpolate)  # (only conv_out reads it; the interpolate form up-samples the fp32 rows)
        co = dec.conv_out.out_channels
        keep = co
        if not return_static and self.static_channels is not None:
            keep = co - self.static_channels if self.static_channels else 0  # reference: decoded[:, :-static_channels] (:1050-1052)
        if up_out and not dec.conv_out.interpolate:
            cv = self._conv(n, B, H, W, dec.conv_out.conv)  # [B H W][4 co] -> pixel_shuffle straight into the NCHW result
            out = torch.empty(B, keep, 2 * H, 2 * W, device=dev, dtype=torch.float32)
            if keep:
                hip.pixel_shuffle_to_chan(cv, out, B=B, H=H, W=W, cout=co, keep=keep)
            return out
        if up_out:  # nearest x2, then the conv at the output width
            ci = dec.conv_out.in_channels
            up32 = torch.empty(B * 4 * H * W, ci, device=dev, dtype=torch.float32) if not self._split else None
            ups = self._srows(B * 4 * H * W, ci, dev)
            hip.upsample_nearest2x_rows(n[0], up32, B=B, H=H, W=W, C=ci, ldx=n[0].shape[1], ys=ups, fmt=self._fmt)
            n, H, W = (up32, ups), 2 * H, 2 * W
            y = self._conv(n, B, H, W, dec.conv_out.conv)
        else:
            y = self._conv(n, B, H, W, dec.conv_out)
        out = torch.empty(B, keep, H, W, device=dev, dtype=torch.float32)
        if keep:
            hip.token_to_chan(y, out, B=B, C=keep, N=H * W, ldi=co)
        return out

    # -- public API ---------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x, return_dict: bool = True, temb=None, embedded_t: bool = False, static_conditioning_tensor=None):
        if self.use_slicing and x.shape[0] > 1:
            raise NotImplementedError("Slicing is not supported for encoding.")
        if self._plan is None:
            self._build_plan()
        dev = self.device
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        B, C, H, W = x.shape
        if self.use_tiling and (W > 512 or H > 512):
            raise NotImplementedError("Tiling is not supported for encoding.")
        enc = self.encoder
        cs = 0
        if static_conditioning_tensor is not None:
            st = static_conditioning_tensor.to(device=dev, dtype=torch.float32)
            if st.shape[0] != B:
                st = st.expand(B, -1, -1, -1)
            st = st.contiguous()
            cs = st.shape[1]
        assert C + cs == enc.conv_in.in_channels, "channel count does not match conv_in"
        t = self._temb_arg(temb, embedded_t, B, dev)
        if t is not None:  # timestep-conditioned variant (models/DCAE.py:982-984): temb is one more static input of the captured launch sequence
            emb = bool(embedded_t)
            if self.use_hip_graph and B <= self.GRAPH_MAX_FRAMES:
                out = self._graphed(("enc_t", B, C, H, W, cs, emb, tuple(t.shape)), lambda xx, *r: self._encode_launch(xx, r[0] if cs else None, r[-1], emb), [x] + ([st] if cs else []) + [t])
            else:
                out = self._encode_launch(x, st if cs else None, t, emb)
        elif self.use_hip_graph and B <= self.GRAPH_MAX_FRAMES:
            out = self._graphed(("enc", B, C, H, W, cs), self._encode_launch, [x] + ([st] if cs else []))
        else:
            out = self._encode_launch(x, st if cs else None)
        if not return_dict:
            return (out,)
        return EncoderOutput(latent=out)

    @torch.no_grad()
    def decode(self, z, return_dict: bool = True, temb=None, embedded_t: bool = False, return_static=False):
        if self.use_slicing and z.size(0) > 1:
            raise NotImplementedError("Slicing is not supported for decoding.")
        if self._plan is None:
            self._build_plan()
        dev = self.device
        z = z.to(device=dev, dtype=torch.float32).contiguous()
        B, C, H, W = z.shape
        if self.use_tiling and (W > 512 // self.spatial_compression_ratio or H > 512 // self.spatial_compression_ratio):
            raise NotImplementedError("Tiling is not supported for decoding.")
        dec = self.decoder
        if C % 4:
            raise NotImplementedError("latent_channels must be a multiple of 4")
        t = self._temb_arg(temb, embedded_t, B, dev)
        if t is not None:
            emb = bool(embedded_t)
            if self.use_hip_graph and B <= self.GRAPH_MAX_FRAMES:
                out = self._graphed(("dec_t", B, C, H, W, bool(return_static), emb, tuple(t.shape)), lambda zz, tt: self._decode_launch(zz, return_static, tt, emb), [z, t])
            else:
                out = self._decode_launch(z, return_static, t, emb)
        elif self.use_hip_graph and B <= self.GRAPH_MAX_FRAMES:
            out = self._graphed(("dec", B, C, H, W, bool(return_static)), lambda zz: self._decode_launch(zz, return_static), [z])
        else:
            out = self._decode_launch(z, return_static)
        if not return_dict:
            return (out,)
        return DecoderOutput(sample=out)

    def forward(self, sample, return_dict: bool = True, time_elapsed=None, static_conditioning_tensor=None, return_static: bool = False):
        # models/DCAE.py:1067-1085: `time_elapsed` is embedded ONCE (time_proj + timestep_embedder) and handed to both halves with embedded_t=True
        temb, emb = time_elapsed, False
        if time_elapsed is not None and self.timestep_embedder is not None:
            if self._plan is None:
                self._build_plan()
            t = torch.as_tensor(time_elapsed).to(device=self.device, dtype=torch.float32).reshape(-1)
            temb, emb = self._embed_t_launch(t, int(sample.shape[0])), True
        z = self.encode(sample, return_dict=False, temb=temb, embedded_t=emb, static_conditioning_tensor=static_conditioning_tensor)[0]
        y = self.decode(z, return_dict=False, temb=temb, embedded_t=emb, return_static=return_static)[0]
        if not return_dict:
            return (y,)
        return DecoderOutput(sample=y)
