"""``LaDCastTransformer3DModel`` on MI355X HIP kernels.

Drop-in for ``ladcast.models.LaDCast_3D_model.LaDCastTransformer3DModel``
(models/LaDCast_3D_model.py:569-1071): same constructor kwargs / ``config``, same parameter
names (a reference ``diffusion_pytorch_model.safetensors`` loads strictly), same
``forward(hidden_states, timestep, conditioning_tensors, time_elapsed=None, ...)`` returning
an object with ``.sample`` (or a tuple).  The ``nn.Module`` tree below is a *parameter
container only*: no torch arithmetic runs in ``forward`` -- every tensor op is a call into
``libladcast_hip.so`` (``ladcast_amd.hip``).

MI355X-first execution plan (vs the reference's op-by-op torch graph):
* both token streams live in ONE joint buffer ``h[B, Nx+Nc, D]`` (pred tokens first), so the
  concatenations at models/LaDCast_3D_model.py:89,188-190,436,457,460 never move data;
* q/k/v weights are fused into one ``[3D, D]`` matrix per stream at load time, the QKV GEMM
  writes token-major ``[B, S, 3D]`` which the attention kernel reads in place (no head
  transposes), and attention writes straight into the ``[attn | mlp]`` concat buffer of the
  single-stream block;
* bias, activation, AdaLN gate and the residual add are the GEMM epilogue;
* RoPE tables, the year embedding MLP and the workspace are cached across the 39 forwards
  of a sampler chunk (the reference rebuilds them every call, Q11).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from .embeddings import get_year_sincos_embedding, rope_tables_from_grid, rotary_1d
from .modeling_utils import ModelMixin


# ---------------------------------------------------------------------------
# parameter containers (names = reference / diffusers attribute names)
# ---------------------------------------------------------------------------
class _RMSNormP(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        """torch arithmetic, reached ONLY from a user-supplied attention processor (`set_attn_processor`): diffusers RMSNorm - fp32
        statistics, x * rsqrt(mean(x^2) + eps) * weight.  The built-in path runs this inside ldc_qk_rmsnorm_rope / the QKV epilogue."""
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(var + self.eps) * self.weight


class LaDCastAttnProcessor2_0:
    """The built-in attention processor (models/LaDCast_3D_model.py:64-221).  Here it is a MARKER: an attention layer whose
    processor is an instance of this class (the default) runs the fused HIP path - QKV GEMM with the norm / rotary / split epilogue,
    `ldc_attn_fwd*`, output projections as GEMM epilogues - and `__call__` is never invoked.  Any other object installed through
    `set_attn_processor` / `Attention.set_processor` is CALLED with the reference's protocol, on torch tensors (slow path, fp32 mode
    only; see `LaDCastTransformer3DModel.set_attn_processor`)."""

    def __call__(self, *a, **k):
        raise RuntimeError("the built-in processor is fused into the HIP forward and is not callable on its own; "
                           "use the C ABI (ldc_gemm_grouped_bf16x3_qkv + ldc_attn_fwd_split, or ldc_qk_rmsnorm_rope + ldc_attn_fwd)")


class _TimestepEmbeddingP(nn.Module):
    def __init__(self, in_channels, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, dim)
        self.linear_2 = nn.Linear(dim, dim)


class _CombinedTimestepTextProjP(nn.Module):
    def __init__(self, dim, pooled_dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbeddingP(256, dim)
        self.text_embedder = _TimestepEmbeddingP(pooled_dim, dim)  # PixArtAlphaTextProjection: same two Linears


class _AdaLinearP(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.linear = nn.Linear(dim_in, dim_out)


class _ActProjP(nn.Module):
    def __init__(self, a, b):
        super().__init__()
        self.proj = nn.Linear(a, b)


class _FeedForwardP(nn.Module):
    def __init__(self, dim, mult):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.ModuleList([_ActProjP(dim, inner), nn.Identity(), nn.Linear(inner, dim)])


class _AttentionP(nn.Module):
    def __init__(self, dim, heads, head_dim, eps=1e-7, added=False, pre_only=False):
        super().__init__()
        inner = heads * head_dim
        self.heads = heads
        self.to_q = nn.Linear(dim, inner)
        self.to_k = nn.Linear(dim, inner)
        self.to_v = nn.Linear(dim, inner)
        self.norm_q = _RMSNormP(head_dim, eps)
        self.norm_k = _RMSNormP(head_dim, eps)
        if added:
            self.add_k_proj = nn.Linear(dim, inner)
            self.add_v_proj = nn.Linear(dim, inner)
            self.add_q_proj = nn.Linear(dim, inner)
            self.norm_added_q = _RMSNormP(head_dim, eps)
            self.norm_added_k = _RMSNormP(head_dim, eps)
            self.to_add_out = nn.Linear(inner, dim)
        else:  # diffusers Attention: absent projections are None attributes (the reference's processor tests `attn.add_q_proj is None`)
            self.add_q_proj = self.add_k_proj = self.add_v_proj = None
            self.norm_added_q = self.norm_added_k = None
            self.to_add_out = None
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Identity()])
        else:
            self.to_out = None
        self.processor = LaDCastAttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self):
        return self.processor

    @property
    def foreign_processor(self):
        """the user-supplied processor of this layer, or None while the built-in (fused) one is installed"""
        p = self.processor
        return None if p is None or isinstance(p, LaDCastAttnProcessor2_0) else p


class _PatchEmbedP(nn.Module):
    """parameter container of HunyuanVideoPatchEmbed (models/embeddings.py:38-59): a Conv3d with kernel = stride = the patch; run as a GEMM over
    the patch rows the model forms (`_patches`)"""

    def __init__(self, in_chans, dim, patch=(1, 1, 1)):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, dim, kernel_size=tuple(patch), stride=tuple(patch))


class _RefinerBlockP(nn.Module):
    def __init__(self, heads, head_dim, mlp_width_ratio=4.0):
        super().__init__()
        d = heads * head_dim
        self.norm1 = nn.LayerNorm(d, elementwise_affine=True, eps=1e-7)
        self.attn = _AttentionP(d, heads, head_dim, pre_only=True)
        self.norm2 = nn.LayerNorm(d, elementwise_affine=True, eps=1e-7)
        self.ff = _FeedForwardP(d, mlp_width_ratio)
        self.norm_out = _AdaLinearP(d, 2 * d)


class _IndividualTokenRefinerP(nn.Module):
    def __init__(self, heads, head_dim, n):
        super().__init__()
        self.refiner_blocks = nn.ModuleList([_RefinerBlockP(heads, head_dim) for _ in range(n)])


class _TokenRefinerP(nn.Module):
    def __init__(self, in_channels, heads, head_dim, n):
        super().__init__()
        d = heads * head_dim
        self.time_text_embed = _CombinedTimestepTextProjP(d, in_channels)
        self.proj_in = nn.Linear(in_channels, d)
        self.token_refiner = _IndividualTokenRefinerP(heads, head_dim, n)


class _SingleBlockP(nn.Module):
    def __init__(self, heads, head_dim, mlp_ratio):
        super().__init__()
        d = heads * head_dim
        mlp = int(d * mlp_ratio)
        self.attn = _AttentionP(d, heads, head_dim, pre_only=True)
        self.norm = _AdaLinearP(d, 3 * d)
        self.proj_mlp = nn.Linear(d, mlp)
        self.proj_out = nn.Linear(d + mlp, d)


class _DualBlockP(nn.Module):
    def __init__(self, heads, head_dim, mlp_ratio):
        super().__init__()
        d = heads * head_dim
        self.norm1 = _AdaLinearP(d, 6 * d)
        self.norm1_context = _AdaLinearP(d, 6 * d)
        self.attn = _AttentionP(d, heads, head_dim, added=True)
        self.ff = _FeedForwardP(d, mlp_ratio)
        self.ff_context = _FeedForwardP(d, mlp_ratio)


# ---------------------------------------------------------------------------
class _Workspace:
    """Device scratch for one (B, Nx, Nc) problem; allocated once, reused by every forward."""

    def __init__(self, dev, B, Bt, Nx, Nc, D, C_in, C_out, n_mod):
        f = dict(device=dev, dtype=torch.float32)
        self.mods = torch.empty(B, n_mod, **f)  # every temb-driven AdaLN modulation vector of one forward
        S = Nx + Nc
        self.h = torch.empty(B, S, D, **f)
        self.nh = torch.empty(B, S, D, **f)
        self.qkv = torch.empty(B, S, 3 * D, **f)
        self.att = torch.empty(B, S, D, **f)
        self.cat = torch.empty(B, S, 5 * D, **f)
        self.xtok = torch.empty(B, Nx, C_in, **f)
        self.ctok = torch.empty(B, Nc, C_in, **f)
        self.ctx0 = torch.empty(B, Nc, D, **f)
        self.otok = torch.empty(B, Nx, C_out, **f)
        self.tsin = torch.empty(Bt, 256, **f)
        self.t1 = torch.empty(Bt, D, **f)
        self.t2 = torch.empty(Bt, D, **f)
        self.t1m = torch.empty(Bt, D, **f)  # main time_text_embed's timestep MLP (computed with the refiner's, one launch per layer)
        self.t2m = torch.empty(Bt, D, **f)
        self.pooled = torch.empty(B, D, **f)
        self.p1 = torch.empty(B, D, **f)
        self.temb_r = torch.empty(B, D, **f)
        self.temb = torch.empty(B, D, **f)
        self.mod_a = torch.empty(B, 6 * D, **f)
        self.mod_b = torch.empty(B, 6 * D, **f)


class LaDCastTransformer3DModel(ModelMixin):
    _supports_gradient_checkpointing = False

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
    ) -> None:
        super().__init__()
        self.register_to_config(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        # nope=True (models/LaDCast_3D_model.py:710-712,897-918): the rotary tables are temporal-only over the whole head dimension (_rope_tables)
        # patch sizes != 1 (no shipped YAML; models/LaDCast_3D_model.py:657-663,758,866-871,1044-1062): since round 6 - the model re-orders a patch's
        # p_t p p values into the channel axis (`_patches`), after which every kernel sees the patch grid as a patch-size-1 problem
        if scale_attn_by_lat and patch_size != 1:
            raise NotImplementedError("scale_attn_by_lat hard-wires the 15 x 30 token grid (models/LaDCast_3D_model.py:684-693): patch size 1 only")
        if attention_head_dim != 128 or qk_norm != "rms_norm":
            raise NotImplementedError("the attention kernel is built for head_dim 128 + rms_norm q/k")
        if in_channels % 4 or (conditioning_tensor_in_channels or in_channels) % 4:
            raise NotImplementedError("channel counts must be multiples of 4 (16-byte rows)")
        heads, hd = num_attention_heads, attention_head_dim
        d = heads * hd
        out_channels = out_channels or in_channels
        self.inner_dim = d
        ps3 = (patch_size_t, patch_size, patch_size)
        self._patch_vol = patch_size_t * patch_size * patch_size
        if (in_channels * self._patch_vol) % 4 or ((conditioning_tensor_in_channels or in_channels) * self._patch_vol) % 4:
            raise NotImplementedError("channels x patch volume must be a multiple of 4 (16-byte rows)")
        self.x_embedder = _PatchEmbedP(in_channels, d, ps3)
        if conditioning_tensor_intermediate_proj_dim is None:
            conditioning_tensor_intermediate_proj_dim = d
        self.context_embedder = _PatchEmbedP(conditioning_tensor_in_channels, d, ps3)
        self.context_refiner = _TokenRefinerP(conditioning_tensor_intermediate_proj_dim, heads, hd, num_refiner_layers)
        self.time_text_embed = _CombinedTimestepTextProjP(d, d)
        self.time_elapsed_embed = _TimestepEmbeddingP(256, 2 * d) if incl_time_elapsed else None
        if spatial_deg2rad:
            rope_spatial_grid_start_pos = [float(np.deg2rad(v)) for v in rope_spatial_grid_start_pos]
            rope_spatial_grid_end_pos = [float(np.deg2rad(v)) for v in rope_spatial_grid_end_pos]
        self.rope_spatial_grid_start_pos = rope_spatial_grid_start_pos
        self.rope_spatial_grid_end_pos = rope_spatial_grid_end_pos
        assert sum(rope_axes_dim) == hd, "sum(rope_axes_dim) must equal attention_head_dim"
        assert sum(conditioning_tensor_rope_axes_dim) == hd
        self.transformer_blocks = nn.ModuleList([_DualBlockP(heads, hd, mlp_ratio) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList([_SingleBlockP(heads, hd, mlp_ratio) for _ in range(num_single_layers)])
        self.norm_out = _AdaLinearP(d, 2 * d)
        self.proj_out = nn.Linear(d, self._patch_vol * out_channels)  # :758
        self.requires_grad_(False)
        # scale_attn_by_lat (models/LaDCast_3D_model.py:682-693): normalised cos-latitude weights of the (hard-wired) 15 x 30 latent grid,
        # added to the attention scores as a (1, 1, 1, keys) float mask; a plain attribute (not a parameter / buffer) as in the reference
        self.scale_attn_by_lat = bool(scale_attn_by_lat)
        self.attn_lat_weights = None
        if scale_attn_by_lat:
            w = np.cos(np.deg2rad(np.linspace(-83.25, 84.75, 15)))
            w = w / w.mean()  # evaluate/utils.py:40-48
            self.attn_lat_weights = torch.from_numpy(w / w.sum()).float().repeat_interleave(30).view(1, 1, 1, -1)
        self._kbias = {}
        self._plan = None
        self._plan_gen = 0  # bumped on every plan rebuild (plan_identity): graphs captured outside the model key on it
        self._ws = {}
        self._rope = {}
        self._te_cache = None
        self._te_buf = {}
        self._graphs = {}
        self._capture_stream = None
        self.use_hip_graph = False
        self.gemm_precision = "fp32"

    def enable_hip_graph(self, flag: bool = True):
        """Capture the ~110 kernel launches of one forward into a hipGraph per input shape and replay it
        (the launches are otherwise host-bound: ~8 us of Python per launch against 5-15 us kernels).
        Inputs are copied into static buffers; the returned sample is a fresh tensor."""
        if flag and self._foreign_processors():
            raise NotImplementedError("hipGraph capture is for the built-in attention path; a user-supplied attention processor runs eagerly")
        self.use_hip_graph = bool(flag)
        if not flag:
            self._graphs = {}
        return self

    def set_attn_lat_weights(self, weights: torch.Tensor):
        """replace `attn_lat_weights` ((1, 1, 1, h * w) or (h * w,)) and drop the device copies / graphs made from the old ones"""
        self.attn_lat_weights = weights.detach().float().reshape(1, 1, 1, -1).cpu()
        self._kbias = {}
        self._graphs = {}
        return self

    def _key_bias(self, frames, dev):
        """the per-key additive score bias of an attention call over `frames` latent frames of keys (:873-880: the weights tiled per
        frame), padded for the split kernels; None when scale_attn_by_lat is off"""
        if not self.scale_attn_by_lat:
            return None
        key = (frames, str(dev))
        if key not in self._kbias:
            self._kbias[key] = hip.pad_key_bias(self.attn_lat_weights.reshape(-1).repeat(frames).to(dev))
        return self._kbias[key]

    def set_gemm_precision(self, mode: str):
        """"fp32": exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).  "bf16x3": split-bf16 error-compensated
        contraction on the bf16 matrix cores (hi*hi + hi*lo + lo*hi, fp32 accumulate), ~4e-6 rel-L2 per
        forward vs fp32 -- see DESIGN.md section 4.  Applies to the token-stream GEMMs and both attention
        contractions; softmax, norms, RoPE, modulation vectors and the sampler state stay fp32 / fp64.
        "bf16": the mixed-precision mode of BASELINE configs[4] ("fp16/bf16 mixed") - the same kernels and operand images
        with the lo terms skipped: ONE bf16 MFMA per product (operands rounded to bf16, fp32 accumulate), what
        ``torch.autocast(bfloat16)`` does to the reference's Linears and SDPA.  Everything the reference keeps in fp32 under
        autocast stays fp32 here as in the other modes (conditioning embedding ``temb``, models/LaDCast_3D_model.py:953; all
        (B, D)-vector Linears; norms, softmax statistics, RoPE, residual stream, fp64 sampler state) - and the GEMM outputs
        are NOT rounded to bf16, so the mode is tighter than autocast - measured: 3.2e-3 rel-L2 per 375M forward vs the fp32 oracle against
        5.4e-3 for the oracle under the reference's autocast recipe (tests/test_gpu_model.py, oracle/autocast.py).  Stated tolerances
        (measured x 2, per stage): ladcast_amd/precision.py."""
        if mode not in ("fp32", "bf16x3", "bf16"):
            raise ValueError("gemm precision must be 'fp32', 'bf16x3' or 'bf16'")
        if mode != self.gemm_precision:
            self.gemm_precision = mode
            self._plan = None
            self._graphs = {}
        return self

    # -- diffusers-style processor surface (models/LaDCast_3D_model.py:763-827) --------------
    @property
    def attn_processors(self):
        out = {}
        for name, mod in self.named_modules():
            if isinstance(mod, _AttentionP):
                out[f"{name}.processor"] = mod.get_processor()
        return out

    def set_attn_processor(self, processor):
        """models/LaDCast_3D_model.py:793-827.  An instance of `LaDCastAttnProcessor2_0` (the default) selects the fused HIP attention
        path.  ANY OTHER processor is honoured, never ignored: the layer's attention stage (q/k/v projections, q/k norm, rotary
        embedding, SDPA and - dual blocks - the output projections) is then the processor's own torch code, called with the reference's
        protocol `proc(attn, hidden_states, encoder_hidden_states=, attention_mask=, image_rotary_emb=, cond_image_rotary_emb=)`
        (only the keywords its signature accepts, as diffusers' `Attention.forward` does) on fp32 device tensors, `attn` being this
        model's parameter container (diffusers attribute names, `nn.Linear` projections, callable RMSNorms).  That is a slow path by
        nature (torch kernels, ~3x the launches); it needs `set_gemm_precision("fp32")` and eager launching, and the forward raises
        `NotImplementedError` otherwise.  The fast way to change the attention arithmetic is the C ABI."""
        mods = {f"{n}.processor": m for n, m in self.named_modules() if isinstance(m, _AttentionP)}
        if isinstance(processor, dict):
            if len(processor) != len(mods):
                raise ValueError(
                    f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                    f" number of attention layers: {len(mods)}. Please make sure to pass {len(mods)} processor classes."
                )
            for k, m in mods.items():
                m.set_processor(processor[k])
        else:
            for m in mods.values():
                m.set_processor(processor)
        if self._foreign_processors():
            if self.use_hip_graph:
                self.enable_hip_graph(False)  # captured graphs hold the fused path
            self._graphs = {}

    def _foreign_processors(self):
        return [m for m in self.modules() if isinstance(m, _AttentionP) and m.foreign_processor is not None]

    def _call_processor(self, attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb, cond_image_rotary_emb):
        """diffusers `Attention.forward`: hand the processor the keyword arguments its `__call__` signature accepts"""
        import inspect

        proc = attn.foreign_processor
        kw = dict(encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, image_rotary_emb=image_rotary_emb,
                  cond_image_rotary_emb=cond_image_rotary_emb)
        params = inspect.signature(proc.__call__).parameters
        if not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
            kw = {k: v for k, v in kw.items() if k in params}
        out = proc(attn, hidden_states, **kw)
        if not (isinstance(out, tuple) and len(out) == 2):
            raise TypeError("an attention processor must return (hidden_states, encoder_hidden_states) (models/LaDCast_3D_model.py:78-86,221)")
        return out

    # -- plan: fused weights ---------------------------------------------------------------
    def _apply(self, fn, *a, **k):  # any .to()/.cuda() invalidates the fused copies + caches
        self._plan = None
        self._ws = {}
        self._rope = {}
        self._te_cache = None
        self._te_buf = {}
        self._graphs = {}
        self._kbias = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plan = None
        self._graphs = {}
        return super().load_state_dict(*a, **k)

    @staticmethod
    def _fuse_qkv(q, k, v):
        return (
            torch.cat([q.weight, k.weight, v.weight], dim=0).contiguous(),
            torch.cat([q.bias, k.bias, v.bias], dim=0).contiguous(),
        )

    def _build_plan(self):
        self._upcast_to_fp32()  # a bf16 / fp16 model loads: parameters are up-cast once, with a warning
        if not next(self.parameters()).is_cuda:
            raise RuntimeError("LaDCastTransformer3DModel must live on a HIP device (no CPU fallback)")
        plan = SimpleNamespace(attn={})
        for name, mod in self.named_modules():
            if isinstance(mod, _AttentionP):
                e = SimpleNamespace()
                e.wqkv, e.bqkv = self._fuse_qkv(mod.to_q, mod.to_k, mod.to_v)
                if mod.add_q_proj is not None:
                    e.wqkv_c, e.bqkv_c = self._fuse_qkv(mod.add_q_proj, mod.add_k_proj, mod.add_v_proj)
                plan.attn[id(mod)] = e
        d = self.inner_dim
        plan.wx = self.x_embedder.proj.weight.reshape(d, -1).contiguous()
        plan.wc = self.context_embedder.proj.weight.reshape(d, -1).contiguous()
        # split-bf16 / bf16 modes: the patch embeds run on the MFMA kernel with K zero-padded to a whole 128-byte k-step
        # (32 split values, 64 plain bf16 values)
        kq = 64 if self.gemm_precision == "bf16" else 32
        plan.kx_pad = -(-plan.wx.shape[1] // kq) * kq
        plan.kc_pad = -(-plan.wc.shape[1] // kq) * kq
        # Every AdaLN modulation driven by the conditioning embedding (dual blocks: 6D per stream, single blocks: 3D,
        # output head: 2D) is one [sum N, D] matrix: one HBM-bound GEMV launch per forward instead of one per block.
        mods = []
        for blk in self.transformer_blocks:
            mods += [blk.norm1.linear, blk.norm1_context.linear]
        mods += [blk.norm.linear for blk in self.single_transformer_blocks]
        mods.append(self.norm_out.linear)
        plan.mod_w = torch.cat([m.weight for m in mods], dim=0).contiguous()
        plan.mod_b = torch.cat([m.bias for m in mods], dim=0).contiguous()
        plan.mod_off, off = {}, 0
        for m in mods:
            plan.mod_off[id(m)] = off
            off += m.weight.shape[0]
        # split-bf16 mode: every token-stream GEMM weight gets a pre-split [N][K/8][hi|lo] copy (same bytes as fp32)
        plan.split = self.gemm_precision in ("bf16x3", "bf16")  # pre-converted operands: split-bf16 groups / plain bf16 rows
        plan.one_term = self.gemm_precision == "bf16"
        plan.packed = {}
        if plan.split:
            ws = [e.wqkv for e in plan.attn.values()] + [e.wqkv_c for e in plan.attn.values() if hasattr(e, "wqkv_c")]
            ws.append(self.context_refiner.proj_in.weight)
            ws.append(self.proj_out.weight)
            for blk in self.context_refiner.token_refiner.refiner_blocks:
                ws += [blk.ff.net[0].proj.weight, blk.ff.net[2].weight]
            for blk in self.transformer_blocks:
                ws += [blk.attn.to_out[0].weight, blk.attn.to_add_out.weight, blk.ff.net[0].proj.weight, blk.ff.net[2].weight,
                       blk.ff_context.net[0].proj.weight, blk.ff_context.net[2].weight]
            for blk in self.single_transformer_blocks:
                ws += [blk.proj_mlp.weight, blk.proj_out.weight]
            pack = hip.pack_weight_bf16 if plan.one_term else hip.pack_weight_bf16x2
            for w in ws:
                plan.packed[id(w)] = pack(w)
            for w, kp in ((plan.wx, plan.kx_pad), (plan.wc, plan.kc_pad)):
                wp = torch.zeros(w.shape[0], kp, device=w.device, dtype=w.dtype)
                wp[:, : w.shape[1]] = w
                plan.packed[id(w)] = pack(wp)
        self._plan = plan
        self._plan_gen += 1

    # -- cached tables -----------------------------------------------------------------------
    def _patches(self, x):
        """(B, C, T, H, W) -> (B, C p_t p p, T / p_t, H / p, W / p): a patch's values go to the channel axis in the order (c, dt, dh, dw) - the
        flattening of the Conv3d patch-embed weight [D][C][p_t][p][p] (models/embeddings.py:52-59), so the patch embed is the same per-token GEMM as
        at patch size 1.  A pure re-ordering (one device copy; launches only, capturable); identity at patch size 1."""
        p, pt = self.config.patch_size, self.config.patch_size_t
        if p == 1 and pt == 1:
            return x
        B, C, T, H, W = x.shape
        if T % pt or H % p or W % p:
            raise ValueError(f"a (T, H, W) = {(T, H, W)} tensor is not a whole number of ({pt}, {p}, {p}) patches")
        return x.reshape(B, C, T // pt, pt, H // p, p, W // p, p).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, C * pt * p * p, T // pt, H // p, W // p).contiguous()

    def _unpatch(self, y, C_out):
        """the inverse on the output head's (B, C_out p_t p p, T', H', W') result: the reference's un-patchify permutation (:1047-1062)"""
        p, pt = self.config.patch_size, self.config.patch_size_t
        if p == 1 and pt == 1:
            return y
        B, _, T, H, W = y.shape
        return y.reshape(B, C_out, pt, p, p, T, H, W).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, C_out, T * pt, H * p, W * p).contiguous()

    def _rope_tables(self, r, t_in, h, w, dev):
        key = (r, t_in, h, w, str(dev))
        if key not in self._rope:
            c = self.config
            cond_t = torch.arange(-t_in + 1, 1, dtype=torch.float32)
            pred_t = torch.arange(1, r + 1, dtype=torch.float32)
            lat = torch.linspace(self.rope_spatial_grid_start_pos[0], self.rope_spatial_grid_end_pos[0], steps=h, dtype=torch.float32)
            lon = torch.linspace(self.rope_spatial_grid_start_pos[1], self.rope_spatial_grid_end_pos[1], steps=w, dtype=torch.float32)
            if getattr(c, "nope", False):  # get_1d_rotary_pos_embed(head_dim, temporal coordinate), a frame's row repeated over its h * w tokens (:897-918)
                n = h * w
                pc, ps = (t.repeat_interleave(n, dim=0).contiguous() for t in rotary_1d(c.attention_head_dim, pred_t, c.rope_theta))
                cc, cs = (t.repeat_interleave(n, dim=0).contiguous() for t in rotary_1d(c.attention_head_dim, cond_t, c.rope_theta))
            else:
                pc, ps = rope_tables_from_grid(c.rope_axes_dim, [pred_t, lat, lon], c.rope_theta)
                cc, cs = rope_tables_from_grid(c.conditioning_tensor_rope_axes_dim, [cond_t, lat, lon], c.rope_theta)
            pc, ps, cc, cs = (t.to(dev) for t in (pc, ps, cc, cs))
            # compact (cos_i, sin_i) tables for the fused QKV epilogue: pred rows, cond rows, and the joint table of the single-stream
            # blocks (pred rows, then cond rows)
            pk, ck = hip.compact_rope_table(pc, ps), hip.compact_rope_table(cc, cs)
            jk = torch.cat([pk, ck], dim=0).contiguous()
            self._rope[key] = (pc, ps, cc, cs, pk, ck, jk)
        return self._rope[key]

    def _time_elapsed_embedding(self, time_elapsed, dev):
        """(rows, 2D) device tensor = time_elapsed_embed(year_sincos(time_elapsed)); cached while the caller
        keeps passing the same timestamp tensor (the sampler does, 39 times per chunk)."""
        key = (time_elapsed.data_ptr(), time_elapsed._version, tuple(time_elapsed.shape), str(time_elapsed.device))
        if self._te_cache is not None and self._te_cache[0] == key:
            return self._te_cache[1]
        # the timestamps are needed on the HOST (calendar arithmetic).  Reading a device tensor back stalls the host until everything queued
        # on the stream - the previous sampler chunk - has run; callers that made the tensor from host integers attach them as
        # `tensor.host_values` (pipelines/utils.py does) and no read-back happens.  Reference: a `.item()` sync in every forward.
        stamps = getattr(time_elapsed, "host_values", None)
        if stamps is None or len(stamps) != time_elapsed.numel():
            stamps = time_elapsed.reshape(-1).tolist()
        stamps = [int(v) for v in stamps]
        n, d2 = len(stamps), 2 * self.inner_dim
        emb = get_year_sincos_embedding(stamps, 256)
        emb = hip.upload_nonblocking(emb, dev)  # no host stall either
        te = self.time_elapsed_embed
        if n not in self._te_buf:  # persistent buffers: a captured graph keeps reading the same addresses
            self._te_buf[n] = (torch.empty(n, d2, device=dev, dtype=torch.float32), torch.empty(n, d2, device=dev, dtype=torch.float32))
        mid, out = self._te_buf[n]
        hip.linear_small(emb, te.linear_1.weight, mid, rows=n, N=d2, K=256, bias=te.linear_1.bias, act_out=hip.ACT_SILU)
        hip.linear_small(mid, te.linear_2.weight, out, rows=n, N=d2, K=d2, bias=te.linear_2.bias)
        self._te_cache = (key, out, time_elapsed)  # keep the tensor alive so data_ptr cannot be recycled
        return out

    # -- building blocks ---------------------------------------------------------------------
    def _timestep_mlps(self, timestep, Bt, pooled, B, ws):
        """Both CombinedTimestepTextProjEmbeddings of one forward (the refiner's and the model's): their timestep MLPs
        depend only on the timestep, and the refiner's text MLP starts from the first pooled context - three independent
        first layers in one launch, the two timestep second layers in the next, then the refiner's sum.
        (diffusers CombinedTimestepTextProjEmbeddings: conditioning = timestep_embedder(proj(t)) + text_embedder(pooled))"""
        D = self.inner_dim
        r, m = self.context_refiner.time_text_embed, self.time_text_embed
        P = hip.linear_small_problem
        # the sinusoidal timestep embedding (diffusers Timesteps) is computed while the first Linears stage their input row
        TS = hip.ACT_IN_TIMESTEP_SINCOS
        hip.linear_small_grouped([
            P(timestep, r.timestep_embedder.linear_1.weight, ws.t1, rows=Bt, N=D, K=256, bias=r.timestep_embedder.linear_1.bias, act_in=TS, act_out=hip.ACT_SILU),
            P(timestep, m.timestep_embedder.linear_1.weight, ws.t1m, rows=Bt, N=D, K=256, bias=m.timestep_embedder.linear_1.bias, act_in=TS, act_out=hip.ACT_SILU),
            P(pooled, r.text_embedder.linear_1.weight, ws.p1, rows=B, N=D, K=D, bias=r.text_embedder.linear_1.bias, act_out=hip.ACT_SILU),
        ])
        hip.linear_small_grouped([
            P(ws.t1, r.timestep_embedder.linear_2.weight, ws.t2, rows=Bt, N=D, K=D, bias=r.timestep_embedder.linear_2.bias),
            P(ws.t1m, m.timestep_embedder.linear_2.weight, ws.t2m, rows=Bt, N=D, K=D, bias=m.timestep_embedder.linear_2.bias),
        ])
        tx = r.text_embedder
        hip.linear_small(ws.p1, tx.linear_2.weight, ws.temb_r, rows=B, N=D, K=D, bias=tx.linear_2.bias, add=ws.t2, add_rows=Bt)

    def _text_mlp(self, emb, pooled, B, Bt, t2, ws, out, te=None):
        """text_embedder(pooled) + the timestep MLP output computed earlier; te (rows, 2D): the time-elapsed modulation
        temb * (1 + scale) + shift (models/LaDCast_3D_model.py:958-969) as the second Linear's epilogue"""
        D = self.inner_dim
        tx = emb.text_embedder
        hip.linear_small(pooled, tx.linear_1.weight, ws.p1, rows=B, N=D, K=D, bias=tx.linear_1.bias, act_out=hip.ACT_SILU)
        hip.linear_small(ws.p1, tx.linear_2.weight, out, rows=B, N=D, K=D, bias=tx.linear_2.bias, add=t2, add_rows=Bt, mod=te,
                         mod_rows=1 if te is None else te.shape[0])

    def _attention(self, ws, B, row0, Sx, Sc, out, ldo, o_bs, seg_x, seg_c, out_split=False, key_bias=None, normed=False):
        """Attention over token rows [row0, row0 + Sx + Sc) of the fused qkv buffer -> out.
        fp32 mode: q/k RMSNorm + RoPE per segment in place (seg = (norm_q, norm_k, cos, sin): rows [row0, row0+Sx) use seg_x, the
        next Sc rows seg_c; ldc_qk_rmsnorm_rope) + ldc_attn_fwd.  Split-bf16 / bf16 modes: the QKV projection's epilogue has already
        written the normed, rotated, scaled and split operand rows (`_qkv_epi`), so this is ldc_attn_fwd_split alone."""
        D, H = self.inner_dim, self.config.num_attention_heads
        qkv = ws.qkv
        full = qkv.shape[1]
        S = Sx + Sc
        q = qkv[:, row0:, 0:D]
        k = qkv[:, row0:, D : 2 * D]
        v = qkv[:, row0:, 2 * D : 3 * D]
        if self.gemm_precision in ("bf16x3", "bf16"):
            hip.attn_fwd_split(q, k, v, out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=full * 3 * D, ldo=ldo, o_bs=o_bs, out_split=out_split,
                               one_term=self.gemm_precision == "bf16", key_bias=key_bias)
            return
        segs = [] if normed else [sg for sg in ((Sx, seg_x), (Sc, seg_c)) if sg[0] > 0]  # normed: the QKV projection's epilogue did it
        r0 = row0
        for rows, (nq, nk, c, s_) in segs:
            hip.qk_rmsnorm_rope(qkv[:, :, 0:D], qkv[:, :, D : 2 * D], B=B, row0=r0, rows=rows, H=H, ld=3 * D, bs=full * 3 * D,
                                wq=nq.weight, wk=nk.weight, eps=nq.eps, cos=c, sin=s_)
            r0 += rows
        hip.attn_fwd(q, k, v, out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=full * 3 * D, ldo=ldo, o_bs=o_bs, key_bias=key_bias)

    def _qkv_epi(self, norm_q, norm_k, rope):
        """epilogue descriptor of a fused QKV projection (split modes): per-head RMSNorm weights + compact rotary table (None = no RoPE)"""
        if norm_q.eps != norm_k.eps:
            raise NotImplementedError("the fused QKV epilogue takes one RMSNorm eps for q and k")
        # (exact-fp32 mode: q is not scaled here - ldc_attn_fwd folds the softmax scale into its Q fragments)
        return hip.qkv_epilogue(norm_q.weight, norm_k.weight, rope, eps=norm_q.eps, heads=self.config.num_attention_heads,
                                qscale=0.0 if self.gemm_precision in ("bf16x3", "bf16") else 1.0)

    # -- forward -------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(
        self,
        hidden_states: torch.Tensor,
        timestep: torch.Tensor,
        conditioning_tensors: torch.Tensor,
        time_elapsed: Optional[torch.Tensor] = None,
        attention_kwargs=None,
        return_dict: bool = True,
        coords=None,
        conditioning=None,
    ):
        """`conditioning` (extension, used by this package's samplers): `(pack, index)` from `prepare_conditioning` - the sample-independent
        part of the forward, evaluated for all noise levels of a sampler chunk in one batch; `timestep` / `conditioning_tensors` /
        `time_elapsed` must be the ones the pack was prepared from (entry `index`).  Without it the call computes that part itself."""
        if self._plan is None:
            self._build_plan()
        plan = self._plan
        dev = self.device
        cfg = self.config
        D, H = self.inner_dim, cfg.num_attention_heads
        B, C_in, R, Hh, Ww = hidden_states.shape
        T_in = conditioning_tensors.shape[2]
        Cc = conditioning_tensors.shape[1]
        Nx, Nc = R * Hh * Ww, T_in * Hh * Ww
        S = Nx + Nc
        C_out = cfg.out_channels or cfg.in_channels
        hidden_states = hidden_states.to(device=dev, dtype=torch.float32).contiguous()
        conditioning_tensors = conditioning_tensors.to(device=dev, dtype=torch.float32)
        if conditioning_tensors.shape[0] != B:
            conditioning_tensors = conditioning_tensors.expand(B, -1, -1, -1, -1)
        conditioning_tensors = conditioning_tensors.contiguous()
        timestep = timestep.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        Bt = timestep.shape[0]
        if Bt not in (1, B):
            raise ValueError(f"timestep must have 1 or {B} entries, got {Bt}")
        te = None
        if time_elapsed is not None and self.time_elapsed_embed is not None:
            te = self._time_elapsed_embedding(time_elapsed, dev)  # eager, cached per chunk; never inside a capture
            if te.shape[0] not in (1, B):
                raise ValueError("time_elapsed must have 1 or batch entries")

        if conditioning is not None:  # the sampler prepared the conditioning path for the whole chunk: only the sample-dependent part
            pack, idx = conditioning
            if tuple(pack.shape) != (B,) + tuple(conditioning_tensors.shape[1:]) or not 0 <= idx < pack.levels:
                raise ValueError("`conditioning` was prepared for another batch / conditioning shape")
            out = self._main_device(hidden_states, pack, idx)
            return (out,) if not return_dict else SimpleNamespace(sample=out)
        if self.use_hip_graph:
            gkey = (B, Bt, C_in, R, T_in, Hh, Ww, None if te is None else (te.data_ptr(), te.shape[0]))
            ent = self._graphs.get(gkey)
            if ent is None:
                sx, st, sk = torch.empty_like(hidden_states), torch.empty_like(timestep), torch.empty_like(conditioning_tensors)
                sx.copy_(hidden_states)
                st.copy_(timestep)
                sk.copy_(conditioning_tensors)
                # warm-up and capture on ONE side stream: per-stream workspaces (stream-K counters, attention operands)
                # are created and initialised by the warm-up, so no allocation / memset ends up inside the graph
                self.capture_stream().wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(self._capture_stream):
                    self._forward_device(sx, st, sk, te)
                torch.cuda.synchronize()
                hip.rearm_attention_workspaces(dev)  # ticket counters of the balanced fp32 attention: zeroed before a (re)capture, with nothing in flight (hip.py)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode="thread_local"):
                    sout = self._forward_device(sx, st, sk, te)
                ent = (graph, sx, st, sk, sout)
                self._graphs[gkey] = ent
            graph, sx, st, sk, sout = ent
            sx.copy_(hidden_states)
            st.copy_(timestep)
            sk.copy_(conditioning_tensors)
            graph.replay()
            out = sout.clone()
        else:
            out = self._forward_device(hidden_states, timestep, conditioning_tensors, te)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    # -- launch-only entry points for callers that capture a larger graph (pipelines/edm_sampler.py) ------------------
    def time_elapsed_embedding(self, time_elapsed):
        """the cached (rows, 2D) time-elapsed embedding of `forward` (None when the model has none / no timestamps)"""
        if time_elapsed is None or self.time_elapsed_embed is None:
            return None
        if self._plan is None:
            self._build_plan()
        return self._time_elapsed_embedding(time_elapsed, self.device)

    def plan_identity(self):
        """changes whenever the packed weights are rebuilt (load_state_dict, .to(), set_gemm_precision): key for graphs
        captured outside the model"""
        if self._plan is None:
            self._build_plan()
        return self._plan_gen

    def capture_stream(self):
        """the ONE side stream every graph of this model is captured on (per-stream workspaces - stream-K counters, attention
        operands, scoring scratch - are keyed by stream: one stream = one set, however many graphs are captured)"""
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(device=self.device)
        return self._capture_stream

    @torch.no_grad()
    def forward_launch_only(self, hidden_states, timestep, conditioning_tensors, te, conditioning=None):
        """`forward` without host work: fp32 device tensors in their final shapes, `te` from `time_elapsed_embedding`;
        only kernel launches on the current stream (capturable).  Returns the sample tensor.  `conditioning = (pack, index)`:
        as in `forward`."""
        if self._plan is None:
            self._build_plan()
        if hidden_states.dtype != torch.float32 or not hidden_states.is_cuda or not hidden_states.is_contiguous():
            raise TypeError("forward_launch_only takes a contiguous fp32 device tensor (use forward() for anything else)")
        if conditioning is not None:
            return self._main_device(hidden_states, conditioning[0], conditioning[1])
        return self._forward_device(hidden_states, timestep.reshape(-1), conditioning_tensors, te)

    # sampler chunks evaluate the conditioning path for all their noise levels at once unless this is switched off (A/B measurements,
    # `bench.py --no-batched-conditioning`); results agree to fp32 rounding (another stream-K cut of the same sums), not bit for bit
    batch_conditioning = True
    COND_MAX_ROWS = 96  # (noise level, member) entries per conditioning pass: bounds its workspace (30 MB per entry at 375M)

    @torch.no_grad()
    def prepare_conditioning(self, timesteps, conditioning_tensors, te=None):
        """The sample-independent part of the forward (`_conditioning_device`: context embed, token refiner, `temb`, all AdaLN modulation
        vectors; models/LaDCast_3D_model.py:942-969) for EVERY noise level of a sampler chunk in one batch.  timesteps: (N,) the values
        the model will be called with (`c_noise` of the EDM sampler, `scheduler.timesteps` of the pipeline loop); conditioning_tensors:
        (B, C, T_in, h, w) fp32 device tensor; te: `time_elapsed_embedding(timestamps)`.  Launches only (capturable).  Returns the pack to
        hand to `forward(..., conditioning=(pack, i))` for noise level i; it lives in workspaces of the model and is valid until the
        next `prepare_conditioning` of the same shape."""
        if self._plan is None:
            self._build_plan()
        dev = self.device
        timesteps = timesteps.to(device=dev, dtype=torch.float32).reshape(-1)
        # the same cast `forward` applies: the kernels read raw fp32 device pointers (an fp64 / bf16 / host tensor must never reach them)
        conditioning_tensors = conditioning_tensors.to(device=dev, dtype=torch.float32)
        if conditioning_tensors.dim() != 5:
            raise ValueError("conditioning_tensors must be (B, C, T_in, h, w)")
        N, B = timesteps.shape[0], conditioning_tensors.shape[0]
        shape = tuple(conditioning_tensors.shape)
        per = max(1, self.COND_MAX_ROWS // B)  # noise levels per pass
        if per >= N:
            pk = self._conditioning_device(timesteps.repeat_interleave(B).contiguous(),
                                           conditioning_tensors.unsqueeze(0).expand(N, *shape).reshape(N * B, *shape[1:]).contiguous(), te)
            return SimpleNamespace(ctx=pk.ctx, mods=pk.mods, shape=shape, levels=N)
        Nc, D, NM = shape[2] * shape[3] * shape[4] // self._patch_vol, self.inner_dim, self._plan.mod_w.shape[0]  # context TOKENS (patches)
        key = ("pack", N, B, Nc)
        if key not in self._ws:
            self._ws[key] = (torch.empty(N * B, Nc, D, device=dev, dtype=torch.float32), torch.empty(N * B, NM, device=dev, dtype=torch.float32))
        ctx, mods = self._ws[key]
        for n0 in range(0, N, per):
            n = min(per, N - n0)
            pk = self._conditioning_device(timesteps[n0 : n0 + n].repeat_interleave(B).contiguous(),
                                           conditioning_tensors.unsqueeze(0).expand(n, *shape).reshape(n * B, *shape[1:]).contiguous(), te)
            ctx[n0 * B : (n0 + n) * B].copy_(pk.ctx)
            mods[n0 * B : (n0 + n) * B].copy_(pk.mods)
        return SimpleNamespace(ctx=ctx, mods=mods, shape=shape, levels=N)

    @staticmethod
    def _launchers(plan):
        """What `_conditioning_device` and `_main_device` launch their GEMMs with, for the plan's precision mode:
        (AS, CS, fmt, G, run, run1, run_qkv).  Split-bf16 / bf16 modes: activations that only feed GEMMs (LayerNorm outputs, attention
        outputs, MLP hidden states) are written ONCE in the operand format by their producer (ladcast_hip.h LDC_GEMM_A_SPLIT /
        _C_SPLIT) - same buffers, strides and column offsets as in fp32 mode.  AS / CS: descriptor flags "A is / C is written in the
        operand format"; fmt: what the producers of GEMM operands write - fp32, split-bf16 groups (bf16x3), plain bf16 rows (bf16: a
        row's K values in the first 2 K bytes of its fp32 row; only COLUMN offsets differ, 2 bytes per column instead of 4)."""
        split, packed = plan.split, plan.packed
        AS = (hip.GEMM_A_SPLIT | (hip.GEMM_BF16_1TERM if plan.one_term else 0)) if split else 0
        CS = hip.GEMM_C_SPLIT if split else 0
        fmt = hip.FMT_BF16 if plan.one_term else hip.FMT_SPLIT if split else hip.FMT_F32

        def G(A, W, C, **kw):  # weight in the format of the active precision mode
            return hip.gemm_problem(A, packed[id(W)] if split else W, C, **kw)

        def run(problems):
            hip.gemm_grouped(problems, split_bf16=split)

        def run1(A, W, C, **kw):
            hip.gemm_grouped([G(A, W, C, **kw)], split_bf16=split)

        def run_qkv(problems, epis):
            # QKV projections: in the split modes their epilogue writes the attention operand rows (norm, RoPE, scale, split); in the
            # exact-fp32 mode it applies the per-head RMSNorm and the rotary embedding (plain fp32 rows).  Returns True when q / k leave
            # the launch normed and rotated, False when `_attention` still has to do it (ldc_qk_rmsnorm_rope)
            if split:
                hip.gemm_grouped_qkv(problems, epis)
                return True
            if hip.gemm_grouped_qkv_f32(problems, epis):
                return True
            hip.gemm_grouped(problems, split_bf16=False)
            return False

        return AS, CS, fmt, G, run, run1, run_qkv

    def _forward_device(self, hidden_states, timestep, conditioning_tensors, te):
        """one forward = its conditioning path (batch = the call's members) + the sample-dependent part"""
        pack = self._conditioning_device(timestep, conditioning_tensors, te)
        return self._main_device(hidden_states, SimpleNamespace(ctx=pack.ctx, mods=pack.mods, shape=tuple(conditioning_tensors.shape), levels=1), 0)

    def _conditioning_device(self, timestep, conditioning_tensors, te):
        """The part of the forward that does NOT see the sample: context patch embed, token refiner, conditioning embedding `temb`
        (+ time-elapsed modulation) and every temb-driven AdaLN modulation vector (models/LaDCast_3D_model.py:942-969).  It depends on
        (timestep, conditioning, timestamp) only, so a sampler evaluates it ONCE PER CHUNK for all its noise levels as one batch
        (`prepare_conditioning`) instead of once per network evaluation: the batch index is (noise level, member), the refiner's GEMMs
        see 20 x 450 rows instead of 450, and the 358 MB AdaLN matrix is streamed once per chunk instead of 39 times.  Kernel
        launches only (capturable).  Returns the pack: refined context `ctx` (Bc, Nc, D) and `mods` (Bc, NM) in a workspace of its own."""
        plan = self._plan
        dev = self.device
        cfg = self.config
        D, H = self.inner_dim, cfg.num_attention_heads
        conditioning_tensors = self._patches(conditioning_tensors)
        B, Cc, T_in, Hh, Ww = conditioning_tensors.shape
        Nx, Nc = 0, T_in * Hh * Ww
        S = Nc
        Bt = timestep.shape[0]
        kpad = max(plan.kx_pad, plan.kc_pad)  # differs between precision modes (k-step of the operand format)
        key = ("cond", B, Bt, Nc, kpad)
        if key not in self._ws:
            self._ws[key] = _Workspace(dev, B, Bt, 0, Nc, D, kpad, 4, plan.mod_w.shape[0])
        ws = self._ws[key]
        _, _, cc, cs, _, ck, _ = self._rope_tables(1, T_in, Hh, Ww, dev)  # the conditioning stream's tables do not depend on R
        SD = S * D
        h_c, nh_c = ws.h, ws.nh
        split = plan.split

        AS, CS, fmt, G, run, run1, run_qkv = self._launchers(plan)

        # 1b. context patch embed (k=1 Conv3d == per-token Linear), models/embeddings.py:52-59
        if split:
            KC = plan.kc_pad
            ctok = ws.ctok.view(-1)[: B * Nc * KC].view(B, Nc, KC)
            # token rows are written in the split format by the transpose itself -> the pre-split GEMM kernel
            hip.chan_to_token(conditioning_tensors, ctok, B=B, C=Cc, N=Nc, ldo=KC, fill_cols=KC, out_split=fmt)
            run1(ctok, plan.wc, ws.ctx0, M=Nc, N=D, K=KC, batch=B, a_bs=Nc * KC, c_bs=Nc * D, bias=self.context_embedder.proj.bias, flags=AS)
        else:
            hip.chan_to_token(conditioning_tensors, ws.ctok.view(-1)[: B * Nc * Cc].view(B, Nc, Cc), B=B, C=Cc, N=Nc, ldo=Cc)
            # K = 84: exact-fp32 tile-per-workgroup kernel (3 k-steps: nothing for stream-K to balance)
            hip.gemm(ws.ctok, plan.wc, ws.ctx0, M=Nc, N=D, K=Cc, batch=B, a_bs=Nc * Cc, c_bs=Nc * D, bias=self.context_embedder.proj.bias)

        # scale_attn_by_lat: per-key score bias, the same weights tiled over the frames of the keys (:873-880); key order inside a frame is
        # the token order, and the mask does not depend on the frame, so pred-then-cond (ours) and the reference's tiling agree
        kb_cond = None
        if self.scale_attn_by_lat:
            if Hh * Ww != self.attn_lat_weights.numel():
                raise ValueError("scale_attn_by_lat is hard-wired to the 15 x 30 latent grid (models/LaDCast_3D_model.py:684-692)")
            kb_cond = self._key_bias(T_in, dev)
        # user-supplied attention processors (set_attn_processor): called on torch views of the fp32 buffers, eager launches only
        foreign = bool(self._foreign_processors())
        m_cond = None
        if foreign:
            if split:
                raise NotImplementedError("a user-supplied attention processor runs on the fp32 path: call set_gemm_precision('fp32') "
                                          "(or change the attention arithmetic through the C ABI, include/ladcast_hip.h)")
            if torch.cuda.is_current_stream_capturing():
                raise NotImplementedError("a user-supplied attention processor cannot be captured into a hipGraph")
            if self.scale_attn_by_lat:  # the reference's (1, 1, 1, keys) float mask (:873-880)
                w = self.attn_lat_weights.to(dev)
                m_cond = w.repeat(1, 1, 1, T_in)

        # 2. context refiner, models/LaDCast_3D_model.py:375-390,280-302
        ref = self.context_refiner
        if split:  # the pooling pass also leaves the embedded context in the split format for proj_in (nh_c is free until the first norm)
            hip.mean_rows(ws.ctx0, ws.pooled, B=B, rows=Nc, D=D, ldx=D, x_bs=Nc * D, x_split=nh_c, lds=D, s_bs=SD, fmt=fmt)
        else:
            hip.mean_rows(ws.ctx0, ws.pooled, B=B, rows=Nc, D=D, ldx=D, x_bs=Nc * D)
        self._timestep_mlps(timestep, Bt, ws.pooled, B, ws)
        if split:
            run1(nh_c, ref.proj_in.weight, h_c, M=Nc, N=D, K=D, batch=B, a_bs=SD, c_bs=SD, bias=ref.proj_in.bias, flags=AS)
        else:
            run1(ws.ctx0, ref.proj_in.weight, h_c, M=Nc, N=D, K=D, batch=B, a_bs=Nc * D, c_bs=SD, bias=ref.proj_in.bias)
        for blk in ref.token_refiner.refiner_blocks:
            pa = plan.attn[id(blk.attn)]
            hip.layernorm_mod(h_c, nh_c, B=B, rows=Nc, D=D, ldx=D, x_bs=SD, ldy=D, y_bs=SD, scale=blk.norm1.weight, shift=blk.norm1.bias, mode=1, eps=blk.norm1.eps, out_split=fmt)
            if blk.attn.foreign_processor is not None:  # :280-286
                a, _ = self._call_processor(blk.attn, nh_c, None, m_cond, (cc, cs), None)
                ws.att.copy_(a)
            else:
                nd = run_qkv([G(nh_c, pa.wqkv, ws.qkv, M=Nc, N=3 * D, K=D, batch=B, a_bs=SD, c_bs=S * 3 * D, bias=pa.bqkv, flags=AS)],
                             [self._qkv_epi(blk.attn.norm_q, blk.attn.norm_k, ck)])
                self._attention(ws, B, Nx, Nc, 0, ws.att, D, SD, (blk.attn.norm_q, blk.attn.norm_k, cc, cs), None, key_bias=kb_cond, normed=nd)
            hip.linear_small(ws.temb_r, blk.norm_out.linear.weight, ws.mod_a, rows=B, N=2 * D, K=D, bias=blk.norm_out.linear.bias, act_in=hip.ACT_SILU)
            # gated attention residual + norm2 in one launch
            hip.gate_residual_layernorm(h_c, ws.att, ws.mod_a, nh_c, B=B, rows=Nc, D=D, ld_res=D, res_bs=SD, ld_y=D, y_bs=SD, gate_bs=2 * D,
                                        ld_out=D, out_bs=SD, weight=blk.norm2.weight, bias=blk.norm2.bias, eps=blk.norm2.eps, out_split=fmt)
            f0, f2 = blk.ff.net[0].proj, blk.ff.net[2]
            F = f0.weight.shape[0]
            run1(nh_c, f0.weight, ws.cat, M=Nc, N=F, K=D, batch=B, a_bs=SD, c_bs=Nc * F, bias=f0.bias, act=hip.ACT_SILU, flags=AS | CS)
            run1(ws.cat, f2.weight, h_c, M=Nc, N=D, K=F, batch=B, a_bs=Nc * F, c_bs=SD, bias=f2.bias,
                        gate=ws.mod_a[:, D:], gate_bs=2 * D, R=h_c, ldr=D, r_bs=SD, flags=AS)

        # 3. conditioning embedding, models/LaDCast_3D_model.py:953-969
        hip.mean_rows(h_c, ws.pooled, B=B, rows=Nc, D=D, ldx=D, x_bs=SD)
        self._text_mlp(self.time_text_embed, ws.pooled, B, Bt, ws.t2m, ws, ws.temb, te=te)
        NM = plan.mod_w.shape[0]
        hip.linear_small(ws.temb, plan.mod_w, ws.mods, rows=B, N=NM, K=D, bias=plan.mod_b, act_in=hip.ACT_SILU)

        return SimpleNamespace(ctx=ws.h, mods=ws.mods, rows=B, shape=(B, Cc, T_in, Hh, Ww))

    def _main_device(self, hidden_states, pack, idx):
        """The sample-dependent part of the forward: sample patch embed, dual-stream and single-stream blocks, output head.  `pack` =
        `_conditioning_device`'s result for a batch of (noise level, member) entries, `idx` = this call's noise level: rows
        [idx * B, (idx + 1) * B) of the pack.  Kernel launches only (no host sync, no shape-dependent Python state): capturable."""
        plan = self._plan
        dev = self.device
        cfg = self.config
        D, H = self.inner_dim, cfg.num_attention_heads
        hidden_states = self._patches(hidden_states)
        B, C_in, R, Hh, Ww = hidden_states.shape
        T_in = pack.shape[2] // cfg.patch_size_t  # (the pack remembers the conditioning tensor's shape as the caller passed it)
        Nx, Nc = R * Hh * Ww, T_in * Hh * Ww
        S = Nx + Nc
        C_out = (cfg.out_channels or cfg.in_channels) * self._patch_vol  # columns of the output head per token (:758)
        kpad = max(plan.kx_pad, plan.kc_pad)  # differs between precision modes (k-step of the operand format)
        key = (B, Nx, Nc, kpad)
        if key not in self._ws:
            self._ws[key] = _Workspace(dev, B, 1, Nx, Nc, D, kpad, C_out, 4)
        ws = self._ws[key]
        pc, ps, cc, cs, pk, ck, jk = self._rope_tables(R, T_in, Hh, Ww, dev)
        SD = S * D
        h_x, h_c = ws.h[:, :Nx], ws.h[:, Nx:]
        nh_x, nh_c = ws.nh[:, :Nx], ws.nh[:, Nx:]
        NM = plan.mod_w.shape[0]
        mods = pack.mods[idx * B : (idx + 1) * B]
        split = plan.split

        AS, CS, fmt, G, run, run1, run_qkv = self._launchers(plan)

        # 1a. sample patch embed (k=1 Conv3d == per-token Linear), models/embeddings.py:52-59; the refined context of this noise level
        # becomes the conditioning rows of the joint token buffer (the dual blocks update them in place)
        if split:
            KX = plan.kx_pad
            xtok = ws.xtok.view(-1)[: B * Nx * KX].view(B, Nx, KX)
            hip.chan_to_token(hidden_states, xtok, B=B, C=C_in, N=Nx, ldo=KX, fill_cols=KX, out_split=fmt)
            run1(xtok, plan.wx, h_x, M=Nx, N=D, K=KX, batch=B, a_bs=Nx * KX, c_bs=SD, bias=self.x_embedder.proj.bias, flags=AS)
        else:
            hip.chan_to_token(hidden_states, ws.xtok.view(-1)[: B * Nx * C_in].view(B, Nx, C_in), B=B, C=C_in, N=Nx, ldo=C_in)
            hip.gemm(ws.xtok, plan.wx, h_x, M=Nx, N=D, K=C_in, batch=B, a_bs=Nx * C_in, c_bs=SD, bias=self.x_embedder.proj.bias)
        h_c.copy_(pack.ctx[idx * B : (idx + 1) * B])

        # scale_attn_by_lat: per-key score bias, the same weights tiled over the frames of the keys (:873-880); key order inside a frame is
        # the token order, and the mask does not depend on the frame, so pred-then-cond (ours) and the reference's tiling agree
        kb_all = None
        if self.scale_attn_by_lat:
            if Hh * Ww != self.attn_lat_weights.numel():
                raise ValueError("scale_attn_by_lat is hard-wired to the 15 x 30 latent grid (models/LaDCast_3D_model.py:684-692)")
            kb_all = self._key_bias(R + T_in, dev)
        # user-supplied attention processors (set_attn_processor): called on torch views of the fp32 buffers, eager launches only
        foreign = bool(self._foreign_processors())
        m_all = None
        if foreign:
            if split:
                raise NotImplementedError("a user-supplied attention processor runs on the fp32 path: call set_gemm_precision('fp32') "
                                          "(or change the attention arithmetic through the C ABI, include/ladcast_hip.h)")
            if torch.cuda.is_current_stream_capturing():
                raise NotImplementedError("a user-supplied attention processor cannot be captured into a hipGraph")
            if self.scale_attn_by_lat:  # the reference's (1, 1, 1, keys) float mask (:873-880)
                w = self.attn_lat_weights.to(dev)
                m_all = w.repeat(1, 1, 1, T_in + R)


        def mod_of(linear, width):
            o = plan.mod_off[id(linear)]
            return mods[:, o : o + width]

        # 4. dual-stream blocks, models/LaDCast_3D_model.py:514-566
        for blk in self.transformer_blocks:
            pa = plan.attn[id(blk.attn)]
            mx, mc = mod_of(blk.norm1.linear, 6 * D), mod_of(blk.norm1_context.linear, 6 * D)
            # norm1 + norm1_context: the two streams are adjacent rows of ws.h -> one launch, two modulation sets
            hip.layernorm_mod(ws.h, ws.nh, B=B, rows=S, D=D, ldx=D, x_bs=SD, ldy=D, y_bs=SD, scale=mx[:, D:], shift=mx, split_row=Nx, scale2=mc[:, D:], shift2=mc,
                              mod_bs=NM, mode=0, eps=1e-6, out_split=fmt)
            if blk.attn.foreign_processor is not None:  # :531-545: the processor also applies to_out / to_add_out; gated residual here
                a, ca = self._call_processor(blk.attn, nh_x, nh_c, m_all, (pc, ps), (cc, cs))
                a, ca = a.contiguous(), ca.contiguous()
                hip.gate_residual(h_x, a, mx[:, 2 * D :], h_x, B=B, rows=Nx, D=D, ld_res=D, res_bs=SD, ld_y=D, y_bs=Nx * D, gate_bs=NM)
                hip.gate_residual(h_c, ca, mc[:, 2 * D :], h_c, B=B, rows=Nc, D=D, ld_res=D, res_bs=SD, ld_y=D, y_bs=Nc * D, gate_bs=NM)
            else:
                nd = run_qkv([
                    G(nh_x, pa.wqkv, ws.qkv, M=Nx, N=3 * D, K=D, batch=B, a_bs=SD, c_bs=S * 3 * D, bias=pa.bqkv, flags=AS),
                    G(nh_c, pa.wqkv_c, ws.qkv[:, Nx:], M=Nc, N=3 * D, K=D, batch=B, a_bs=SD, c_bs=S * 3 * D, bias=pa.bqkv_c, flags=AS),
                ], [self._qkv_epi(blk.attn.norm_q, blk.attn.norm_k, pk), self._qkv_epi(blk.attn.norm_added_q, blk.attn.norm_added_k, None)])
                self._attention(ws, B, 0, Nx, Nc, ws.att, D, SD, (blk.attn.norm_q, blk.attn.norm_k, pc, ps),
                                (blk.attn.norm_added_q, blk.attn.norm_added_k, None, None), out_split=fmt, key_bias=kb_all, normed=nd)
                o, oc = blk.attn.to_out[0], blk.attn.to_add_out
                run([
                    G(ws.att, o.weight, h_x, M=Nx, N=D, K=D, batch=B, a_bs=SD, c_bs=SD, bias=o.bias, gate=mx[:, 2 * D :], gate_bs=NM, R=h_x, ldr=D, r_bs=SD, flags=AS),
                    G(ws.att[:, Nx:], oc.weight, h_c, M=Nc, N=D, K=D, batch=B, a_bs=SD, c_bs=SD, bias=oc.bias, gate=mc[:, 2 * D :], gate_bs=NM, R=h_c, ldr=D, r_bs=SD, flags=AS),
                ])
            hip.layernorm_mod(ws.h, ws.nh, B=B, rows=S, D=D, ldx=D, x_bs=SD, ldy=D, y_bs=SD, scale=mx[:, 4 * D :], shift=mx[:, 3 * D :], split_row=Nx,
                              scale2=mc[:, 4 * D :], shift2=mc[:, 3 * D :], mod_bs=NM, mode=0, eps=1e-7, out_split=fmt)
            up, down = [], []
            for (hs, nhs, rows, ff, mod, off) in ((h_x, nh_x, Nx, blk.ff, mx, 0), (h_c, nh_c, Nc, blk.ff_context, mc, Nx)):
                f0, f2 = ff.net[0].proj, ff.net[2]
                F = f0.weight.shape[0]
                hid = ws.cat.view(-1)[off * B * F :]  # [B, rows, F] slab inside the concat scratch
                up.append(G(nhs, f0.weight, hid, M=rows, N=F, K=D, batch=B, a_bs=SD, c_bs=rows * F, bias=f0.bias, act=hip.ACT_GELU_TANH, flags=AS | CS))
                down.append(G(hid, f2.weight, hs, M=rows, N=D, K=F, batch=B, a_bs=rows * F, c_bs=SD, bias=f2.bias,
                              gate=mod[:, 5 * D :], gate_bs=NM, R=hs, ldr=D, r_bs=SD, flags=AS))
            run(up)
            run(down)

        # 5. single-stream blocks, models/LaDCast_3D_model.py:426-468
        for blk in self.single_transformer_blocks:
            pa = plan.attn[id(blk.attn)]
            mod = mod_of(blk.norm.linear, 3 * D)
            F = blk.proj_mlp.weight.shape[0]
            W5 = D + F
            # columns [D, 5 D) of the [attn | mlp] concat rows; in the plain-bf16 row format that is byte offset 2 D, not 4 D
            cat_mlp = ws.cat.view(torch.bfloat16)[:, :, D:] if plan.one_term else ws.cat[:, :, D:]
            hip.layernorm_mod(ws.h, ws.nh, B=B, rows=S, D=D, ldx=D, x_bs=SD, ldy=D, y_bs=SD, scale=mod[:, D:], shift=mod, mod_bs=NM, mode=0, eps=1e-6, out_split=fmt)
            if blk.attn.foreign_processor is not None:  # :443-457: pre_only attention, its output is the first D columns of the concat rows
                run([G(ws.nh, blk.proj_mlp.weight, cat_mlp, M=S, N=F, K=D, batch=B, a_bs=SD, ldc=W5, c_bs=S * W5, bias=blk.proj_mlp.bias,
                       act=hip.ACT_GELU_TANH, flags=AS | CS)])
                a, ca = self._call_processor(blk.attn, ws.nh[:, :Nx], ws.nh[:, Nx:], m_all, (pc, ps), (cc, cs))
                ws.cat[:, :Nx, :D].copy_(a)
                ws.cat[:, Nx:, :D].copy_(ca)
            else:
                nd = run_qkv([
                    G(ws.nh, blk.proj_mlp.weight, cat_mlp, M=S, N=F, K=D, batch=B, a_bs=SD, ldc=W5, c_bs=S * W5, bias=blk.proj_mlp.bias, act=hip.ACT_GELU_TANH,
                      flags=AS | CS),
                    G(ws.nh, pa.wqkv, ws.qkv, M=S, N=3 * D, K=D, batch=B, a_bs=SD, c_bs=S * 3 * D, bias=pa.bqkv, flags=AS),
                ], [None, self._qkv_epi(blk.attn.norm_q, blk.attn.norm_k, jk)])
                self._attention(ws, B, 0, Nx, Nc, ws.cat, W5, S * W5, (blk.attn.norm_q, blk.attn.norm_k, pc, ps),
                                (blk.attn.norm_q, blk.attn.norm_k, cc, cs), out_split=fmt, key_bias=kb_all, normed=nd)
            run1(ws.cat, blk.proj_out.weight, ws.h, M=S, N=D, K=W5, batch=B, a_bs=S * W5, c_bs=SD, bias=blk.proj_out.bias,
                        gate=mod[:, 2 * D :], gate_bs=NM, R=ws.h, ldr=D, r_bs=SD, flags=AS)

        # 6. output head, models/LaDCast_3D_model.py:1044-1062 (patch size 1: un-patchify == transpose)
        mo = mod_of(self.norm_out.linear, 2 * D)
        hip.layernorm_mod(h_x, nh_x, B=B, rows=Nx, D=D, ldx=D, x_bs=SD, ldy=D, y_bs=SD, scale=mo, shift=mo[:, D:], mod_bs=NM, mode=0, eps=1e-7, out_split=fmt)
        run1(nh_x, self.proj_out.weight, ws.otok, M=Nx, N=C_out, K=D, batch=B, a_bs=SD, c_bs=Nx * C_out, bias=self.proj_out.bias, flags=AS)
        out = torch.empty(B, C_out, R, Hh, Ww, device=dev, dtype=torch.float32)
        hip.token_to_chan(ws.otok, out, B=B, C=C_out, N=Nx, ldi=C_out)
        return self._unpatch(out, cfg.out_channels or cfg.in_channels)
