"""ctypes binding of the C ABI in ``include/ladcast_hip.h``.

PyTorch is plumbing here: it owns device memory and the HIP stream.  Every wrapper takes
torch CUDA(=HIP) tensors, passes raw device pointers + sizes + the current stream to the
shared library and raises ``RuntimeError`` on a non-zero status.  Importing this module
without the built library raises immediately -- the product path never falls back to
PyTorch ops.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDC_LIB_PATH: diagnostic builds only (e.g. the in-kernel stamp build made by `make stamps`)
LIB_PATH = os.environ.get("LDC_LIB_PATH") or os.path.join(_HERE, "libladcast_hip.so")

ACT_NONE, ACT_SILU, ACT_GELU_TANH, ACT_RELU = 0, 1, 2, 3
ACT_IN_TIMESTEP_SINCOS = 16  # act_in of the small linears: x = one timestep per row, the input row = its sinusoidal embedding
GEMM_A_SPLIT, GEMM_C_SPLIT, GEMM_BF16_1TERM = 1, 2, 4
GEMM_F32_REGSTAGE = 8  # exact-fp32 problems: stay on the register-staged stream-K kernel (include/ladcast_hip.h)
ATTN_OUT_SPLIT, ATTN_BF16_1TERM, ATTN_OUT_BF16 = 1, 2, 4
ABI_VERSION = 5  # LDC_ABI_VERSION of include/ladcast_hip.h this binding was written against
FMT_F32, FMT_SPLIT, FMT_BF16 = 0, 1, 2  # activation formats of the producers' `out_split` arguments (True == FMT_SPLIT)


class GemmDesc(Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int),
        ("lda", c_int), ("ldw", c_int), ("ldc", c_int), ("ldr", c_int),
        ("a_bs", c_longlong), ("c_bs", c_longlong), ("r_bs", c_longlong),
        ("gate_bs", c_int), ("act", c_int), ("flags", c_int), ("reserved", c_int),
    ]


class GemmProblem(Structure):
    _fields_ = [("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("gate", c_void_p), ("R", c_void_p), ("C", c_void_p), ("d", GemmDesc)]


MAX_GROUPED = 4


class QkvEpilogue(Structure):  # ldc_qkv_epilogue
    _fields_ = [("wq", c_void_p), ("wk", c_void_p), ("rope", c_void_p), ("reserved", c_void_p), ("eps", c_float), ("qscale", c_float),
                ("heads", c_int), ("rope_row0", c_int)]


class LinearSmallProblem(Structure):  # ldc_linear_small_problem
    _fields_ = [("x", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("add", c_void_p), ("y", c_void_p), ("x_rows", c_int), ("add_rows", c_int),
                ("rows", c_int), ("N", c_int), ("K", c_int), ("act_in", c_int), ("act_out", c_int), ("reserved", c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ladcast_amd/csrc`).  ladcast_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    P, I, L, F, D = c_void_p, c_int, c_longlong, c_float, c_double
    sig = {
        "ldc_abi_version": (I, []),
        "ldc_build_arch": (c_char_p, []),
        "ldc_sizeof_gemm_desc": (I, []),
        "ldc_gemm_bias_act": (I, [P, P, P, P, P, P, POINTER(GemmDesc), P]),
        "ldc_sizeof_gemm_problem": (I, []),
        "ldc_gemm_grouped_workspace_bytes": (L, []),
        "ldc_gemm_grouped_workspace_init": (I, [P, L, P]),
        "ldc_gemm_grouped": (I, [POINTER(GemmProblem), I, P, L, P]),
        "ldc_gemm_grouped_bf16x3": (I, [POINTER(GemmProblem), I, P, L, P]),
        "ldc_sizeof_qkv_epilogue": (I, []),
        "ldc_gemm_grouped_bf16x3_qkv": (I, [POINTER(GemmProblem), POINTER(QkvEpilogue), I, P, L, P]),
        "ldc_gemm_grouped_qkv_f32": (I, [POINTER(GemmProblem), POINTER(QkvEpilogue), I, P, L, P]),
        "ldc_attn_qkv_prepare_split": (I, [P, P, P, I, I, I, I, L, I, P, P, P, P, P, P, P, P, F, P]),
        "ldc_attn_fwd_split_workspace_bytes": (L, [I, I, I]),
        "ldc_attn_fwd_split_workspace_max_bytes": (L, []),
        "ldc_attn_fwd_split": (I, [P, P, P, P, I, I, I, I, L, I, L, P, I, P, L, P]),
        "ldc_pack_weight_bf16x2": (I, [P, P, I, I, I, P]),
        "ldc_pack_weight_bf16": (I, [P, P, I, I, I, P]),
        "ldc_linear_small": (I, [P, I, P, P, P, I, P, I, I, I, I, I, P]),
        "ldc_linear_small_grouped": (I, [POINTER(LinearSmallProblem), I, P]),
        "ldc_linear_small_mod": (I, [P, I, P, P, P, I, P, I, P, I, I, I, I, I, P]),
        "ldc_gate_residual_layernorm": (I, [P, P, P, P, I, I, I, I, L, I, L, I, I, L, P, P, F, I, P]),
        "ldc_attn_fwd": (I, [P, P, P, P, I, I, I, I, L, I, L, P, P]),
        "ldc_attn_fwd_ws": (I, [P, P, P, P, I, I, I, I, L, I, L, P, P, L, P]),
        "ldc_attn_fwd_workspace_bytes": (L, []),
        "ldc_qk_rmsnorm_rope": (I, [P, P, I, I, I, I, I, L, P, P, F, P, P, P]),
        "ldc_sphere_conv_nhwc_split": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P]),
        "ldc_sphere_conv_plan": (I, [I, I, I, I, I, I, I, P, P]),
        "ldc_rmsnorm_rows_split": (I, [P, P, P, P, P, P, L, I, I, I, I, I, I, F, I, P]),
        "ldc_pixel_unshuffle_shortcut_split": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
        "ldc_pixel_shuffle_shortcut_split": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
        "ldc_pixel_shuffle_to_chan": (I, [P, P, I, I, I, I, I, P]),
        "ldc_split_rows": (I, [P, P, L, I, I, I, I, P]),
        "ldc_upsample_nearest2x_rows": (I, [P, P, P, I, I, I, I, I, I, I, I, P]),
        "ldc_sphere_dwconv_nhwc_fmt": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
        "ldc_relu_linear_attn_nhwc_fmt": (I, [P, P, I, I, I, I, I, F, I, P, L, P]),
        "ldc_ensemble_scores_workspace_bytes": (L, [I, I, I]),
        "ldc_ensemble_scores": (I, [P, L, L, P, L, P, L, P, I, I, I, I, I, P, P, P, P, L, P]),
        "ldc_layernorm_mod": (I, [P, P, I, I, I, I, L, I, L, P, P, I, I, F, I, P]),
        "ldc_layernorm_mod2": (I, [P, P, I, I, I, I, L, I, L, P, P, I, P, P, I, I, F, I, P]),
        "ldc_mean_rows": (I, [P, P, I, I, I, I, L, P]),
        "ldc_mean_rows_split": (I, [P, P, P, I, I, I, I, L, I, L, I, P]),
        "ldc_chan_to_token_split": (I, [P, P, I, I, I, I, I, I, P]),
        "ldc_gate_residual": (I, [P, P, P, P, I, I, I, I, L, I, L, I, P]),
        "ldc_chan_to_token": (I, [P, P, I, I, I, I, I, P]),
        "ldc_token_to_chan": (I, [P, P, I, I, I, I, P]),
        "ldc_timestep_embedding": (I, [P, P, I, P]),
        "ldc_temb_modulate": (I, [P, P, I, I, I, P]),
        "ldc_chan_affine": (I, [P, P, P, P, F, L, I, L, I, P]),
        "ldc_edm_scale_f64_to_f32": (I, [P, D, P, L, P]),
        "ldc_edm_init_state": (I, [P, D, P, L, P]),
        "ldc_edm_euler": (I, [P, P, D, D, D, D, P, P, L, P]),
        "ldc_edm_churn": (I, [P, P, c_double, P, L, P]),
        "ldc_edm_heun": (I, [P, P, P, P, D, D, D, D, L, P]),
        "ldc_f64_to_f32": (I, [P, P, L, P]),
        "ldc_dpm_step": (I, [P, P, P, P, P, F, F, F, F, F, I, L, P]),
        "ldc_ddim_step": (I, [P, P, P, P, P, F, F, F, F, F, F, I, I, L, P]),
        "ldc_ddpm_step": (I, [P, P, P, P, P, F, F, F, F, F, F, I, L, P]),
        "ldc_scale_f32": (I, [P, F, P, L, P]),
        "ldc_axpby_f32": (I, [P, F, P, F, P, L, P]),
        "ldc_sphere_conv_nhwc": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
        "ldc_sphere_dwconv_nhwc": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
        "ldc_grouped_conv1x1_nhwc": (I, [P, P, P, L, I, I, I, P]),
        "ldc_relu_linear_attn_nhwc": (I, [P, P, I, I, I, I, I, F, P, L, P]),
        "ldc_relu_linear_attn_workspace_bytes": (L, [I, I, I]),
        "ldc_rmsnorm_rows": (I, [P, P, P, P, P, L, I, I, I, I, F, I, P]),
        "ldc_pixel_unshuffle_shortcut": (I, [P, P, P, I, I, I, I, I, P]),
        "ldc_pixel_shuffle_shortcut": (I, [P, P, P, I, I, I, I, I, P]),
        "ldc_chan_regroup": (I, [P, P, L, I, I, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ldc_abi_version() != ABI_VERSION:
        raise RuntimeError("libladcast_hip.so ABI version mismatch")
    if lib.ldc_sizeof_gemm_desc() != ctypes.sizeof(GemmDesc) or lib.ldc_sizeof_gemm_problem() != ctypes.sizeof(GemmProblem):
        raise RuntimeError("ldc_gemm_desc / ldc_gemm_problem layout mismatch between header and binding")
    if lib.ldc_sizeof_qkv_epilogue() != ctypes.sizeof(QkvEpilogue):
        raise RuntimeError("ldc_qkv_epilogue layout mismatch between header and binding")
    return lib, sig


lib, SIGNATURES = _load()


def _check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}")


def upload_nonblocking(t, device):
    """host -> device without stalling the host: a pageable source makes the copy synchronous, i.e. the host would wait for everything
    queued on the stream - the previous sampler chunk - before it can prepare the next one.  Pinned source + non_blocking: the copy
    is only stream-ordered (torch's pinned-memory allocator does not hand the block out again before the copy has run)."""
    device = torch.device(device)
    if device.type != "cuda" or t.device.type != "cpu":
        return t.to(device)
    return (t if t.is_pinned() else t.pin_memory()).to(device, non_blocking=True)


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ladcast_amd kernels need device tensors (no CPU fallback)")


# ---------------------------------------------------------------------------
def gemm(A, W, C, *, M, N, K, batch=1, lda=None, ldw=None, ldc=None, a_bs=0, c_bs=0, bias=None, gate=None,
         gate_bs=0, R=None, ldr=0, r_bs=0, act=ACT_NONE):
    """C[b] = epilogue(A[b] . W^T); A/C/R may be views (data_ptr carries the offset)."""
    _dev(A, W, C, bias, gate, R)
    d = GemmDesc(M, N, K, batch, lda if lda is not None else K, ldw if ldw is not None else K,
                 ldc if ldc is not None else N, ldr, a_bs, c_bs, r_bs, gate_bs, act, 0, 0)
    _check(lib.ldc_gemm_bias_act(_p(A), _p(W), _p(bias), _p(gate), _p(R), _p(C), ctypes.byref(d), _stream()), "ldc_gemm_bias_act")


_grouped_ws = {}


def _grouped_workspace(device):
    key = str(device)
    key = (key, torch.cuda.current_stream(device).cuda_stream)  # one workspace per (device, stream)
    if key not in _grouped_ws:
        ws = torch.empty(lib.ldc_gemm_grouped_workspace_bytes() // 4, device=device, dtype=torch.float32)
        _check(lib.ldc_gemm_grouped_workspace_init(c_void_p(ws.data_ptr()), ws.numel() * 4, _stream()), "ldc_gemm_grouped_workspace_init")
        _grouped_ws[key] = ws
    return _grouped_ws[key]


def gemm_problem(A, W, C, *, M, N, K, batch=1, lda=None, ldw=None, ldc=None, a_bs=0, c_bs=0, bias=None, gate=None, gate_bs=0, R=None,
                 ldr=0, r_bs=0, act=ACT_NONE, flags=0):
    """one entry of a grouped launch (same argument meaning as `gemm`); flags: GEMM_A_SPLIT | GEMM_C_SPLIT (bf16x3 only); GEMM_F32_REGSTAGE
    (exact fp32 only): the register-staged kernel instead of the ring kernel"""
    _dev(A, W, C, bias, gate, R)
    d = GemmDesc(M, N, K, batch, lda if lda is not None else K, ldw if ldw is not None else K, ldc if ldc is not None else N, ldr,
                 a_bs, c_bs, r_bs, gate_bs, act, flags, 0)
    pv = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    return GemmProblem(pv(A), pv(W), pv(bias), pv(gate), pv(R), pv(C), d), (A, W, C, bias, gate, R)


def gemm_grouped(problems, split_bf16=False):
    """Up to MAX_GROUPED independent GEMMs in one persistent stream-K launch (split tiles are reduced inside the launch).
    split_bf16: every problem's W is a `pack_weight_bf16x2` buffer and the bf16x3 kernel is used."""
    n = len(problems)
    if not 1 <= n <= MAX_GROUPED:
        raise ValueError(f"gemm_grouped takes 1..{MAX_GROUPED} problems")
    arr = (GemmProblem * n)(*[p[0] for p in problems])
    ws = _grouped_workspace(problems[0][1][2].device)
    fn = lib.ldc_gemm_grouped_bf16x3 if split_bf16 else lib.ldc_gemm_grouped
    _check(fn(arr, n, c_void_p(ws.data_ptr()), ws.numel() * 4, _stream()), "ldc_gemm_grouped" + ("_bf16x3" if split_bf16 else ""))


def gemm_sk(A, W, C, split_bf16=False, **kw):
    """single GEMM through the stream-K scheduler"""
    gemm_grouped([gemm_problem(A, W, C, **kw)], split_bf16=split_bf16)


def pack_weight_bf16x2(W):
    """fp32 [N, K] (K % 8 == 0) -> packed split-bf16 weight buffer (int32 view, N*K elements = same bytes)"""
    _dev(W)
    W = W.contiguous()
    N, K = W.shape
    out = torch.empty(N * K, device=W.device, dtype=torch.int32)
    _check(lib.ldc_pack_weight_bf16x2(_p(W), _p(out), N, K, K, _stream()), "ldc_pack_weight_bf16x2")
    return out


def pack_weight_bf16(W):
    """fp32 [N, K] (K % 8 == 0) -> plain bf16 [N, K] weight buffer (the W operand of the single-term bf16 mode)"""
    _dev(W)
    W = W.contiguous()
    N, K = W.shape
    out = torch.empty(N * K, device=W.device, dtype=torch.bfloat16)
    _check(lib.ldc_pack_weight_bf16(_p(W), _p(out), N, K, K, _stream()), "ldc_pack_weight_bf16")
    return out


def linear_small(x, W, y, *, rows, N, K, x_rows=None, bias=None, add=None, add_rows=1, act_in=ACT_NONE, act_out=ACT_NONE, mod=None, mod_rows=1):
    """mod ([mod_rows][2 N]): y = y * (1 + mod[:, :N]) + mod[:, N:] as the epilogue (ldc_linear_small_mod)"""
    _dev(x, W, y, bias, add, mod)
    xr = x_rows if x_rows is not None else rows
    if mod is not None:
        _check(lib.ldc_linear_small_mod(_p(x), xr, _p(W), _p(bias), _p(add), add_rows, _p(mod), mod_rows, _p(y), rows, N, K, act_in, act_out, _stream()),
               "ldc_linear_small_mod")
        return
    _check(lib.ldc_linear_small(_p(x), xr, _p(W), _p(bias), _p(add), add_rows, _p(y), rows, N, K, act_in, act_out, _stream()), "ldc_linear_small")


def gate_residual_layernorm(resid, y, gate, out, *, B, rows, D, ld_res, res_bs, ld_y, y_bs, gate_bs, ld_out, out_bs, weight, bias, eps, out_split=False):
    """resid += gate * y in place, then out = LayerNorm(resid) * weight + bias (one launch)"""
    _dev(resid, y, gate, out, weight, bias)
    _check(lib.ldc_gate_residual_layernorm(_p(resid), _p(y), _p(gate), _p(out), B, rows, D, ld_res, res_bs, ld_y, y_bs, gate_bs, ld_out, out_bs,
                                           _p(weight), _p(bias), eps, int(out_split), _stream()), "ldc_gate_residual_layernorm")


LINEAR_SMALL_MAX_GROUPED = 4  # LDC_LINEAR_SMALL_MAX_GROUPED


def linear_small_problem(x, W, y, *, rows, N, K, x_rows=None, bias=None, add=None, add_rows=1, act_in=ACT_NONE, act_out=ACT_NONE):
    """one entry of linear_small_grouped (same arguments as linear_small)"""
    _dev(x, W, y, bias, add)
    q = LinearSmallProblem(_p(x), _p(W), _p(bias), _p(add), _p(y), x_rows if x_rows is not None else rows, add_rows, rows, N, K, act_in, act_out, 0)
    return q, (x, W, y, bias, add)


def linear_small_grouped(problems):
    """up to LINEAR_SMALL_MAX_GROUPED independent small linears in one launch"""
    n = len(problems)
    if not 1 <= n <= LINEAR_SMALL_MAX_GROUPED:
        raise ValueError(f"linear_small_grouped takes 1..{LINEAR_SMALL_MAX_GROUPED} problems")
    arr = (LinearSmallProblem * n)(*[p[0] for p in problems])
    _check(lib.ldc_linear_small_grouped(arr, n, _stream()), "ldc_linear_small_grouped")


_attn_f32_ws = {}


def _attn_f32_workspace(device):
    """counters + slabs of the exact-fp32 attention's balanced schedule (ldc_attn_fwd_ws), one per (device, stream), zero-filled once
    (the launches leave the counters zero) and never replaced: captured hipGraphs hold its pointer.  As `_attn_workspace`: the first
    call on a stream must be made outside a capture."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _attn_f32_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("attn_fwd: the first call on a stream must be made outside a graph capture (run one warm-up forward)")
        ws = torch.zeros((lib.ldc_attn_fwd_workspace_bytes() + 3) // 4, device=device, dtype=torch.float32)
        _attn_f32_ws[key] = ws
    return ws


def attn_fwd(Q, K, V, O, *, B, S, H, ld_qkv, qkv_bs, ldo, o_bs, key_bias=None, use_workspace=True):
    """exact-fp32 attention on fp32 q / k / v views (ldc_attn_fwd_ws); key_bias: [S] additive score bias per key (or None).
    use_workspace=False: never the balanced schedule (ldc_attn_fwd; A/B, tests)"""
    _dev(Q, K, V, O, key_bias)
    if key_bias is not None and key_bias.numel() < S:
        raise ValueError("key_bias must hold one value per key")
    if not use_workspace:
        _check(lib.ldc_attn_fwd(_p(Q), _p(K), _p(V), _p(O), B, S, H, ld_qkv, qkv_bs, ldo, o_bs, _p(key_bias), _stream()), "ldc_attn_fwd")
        return
    ws = _attn_f32_workspace(Q.device)
    st = lib.ldc_attn_fwd_ws(_p(Q), _p(K), _p(V), _p(O), B, S, H, ld_qkv, qkv_bs, ldo, o_bs, _p(key_bias), _p(ws), ws.numel() * 4, _stream())
    if st != 0 and not torch.cuda.is_current_stream_capturing():
        ws.zero_()  # a launch that did not complete may leave ticket counters non-zero: re-arm them before anyone launches again
    _check(st, "ldc_attn_fwd_ws")


def rearm_attention_workspaces(device=None):
    """Zero the ticket counters + slabs of the balanced exact-fp32 attention for every stream of `device` (all devices when None).  The launches
    leave their counters zero, so this is needed only after a faulted / aborted launch, and is what the models call before they (re)capture a
    hipGraph.  Must not be called while a graph that uses the workspace is running.  One workspace serves ONE stream: a graph captured on
    stream A and replayed on stream B shares A's counters - replay it on A, or give B its own capture."""
    want = None if device is None else torch.device(device).index
    for ws in _attn_f32_ws.values():
        if want is None or ws.device.index == want:
            ws.zero_()


_score_ws = {}
_score_keep = []


def ensemble_scores(forecast, truth, clim, lat_weight, out, *, M, C, H, W, member_stride, channel_stride, truth_channel_stride,
                    clim_channel_stride=0, nan_channel=-1, skill_map=None, spread_map=None):
    """out: [5][C] = acc, mse, crps_spread, crps_skill, crps (ladcast_hip.h: ldc_ensemble_scores)"""
    _dev(forecast, truth, clim, lat_weight, out, skill_map, spread_map)
    need = int(lib.ldc_ensemble_scores_workspace_bytes(C, H, W))
    key = (str(forecast.device), torch.cuda.current_stream(forecast.device).cuda_stream)
    if key not in _score_ws or _score_ws[key].numel() * 4 < need:
        if key in _score_ws:
            _score_keep.append(_score_ws[key])  # a captured hipGraph may still point at the smaller one (as _rla_keep)
        _score_ws[key] = torch.empty(need // 4 + 1, device=forecast.device, dtype=torch.float32)
    ws = _score_ws[key]
    _check(lib.ldc_ensemble_scores(_p(forecast), member_stride, channel_stride, _p(truth), truth_channel_stride, _p(clim),
                                   clim_channel_stride, _p(lat_weight), M, C, H, W, nan_channel, _p(out), _p(skill_map), _p(spread_map),
                                   _p(ws), ws.numel() * 4, _stream()), "ldc_ensemble_scores")


def compact_rope_table(cos, sin):
    """the reference's [rows][128] cos / sin tables (every value twice: get_1d_rotary_pos_embed's repeat_interleave(2)) -> the
    [rows][64][2] (cos_i, sin_i) table of `qkv_epilogue`; refuses tables that are not of that form"""
    if not (torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])):
        raise ValueError("rotary tables must hold every (cos, sin) value twice (adjacent-pair RoPE)")
    return torch.stack([cos[:, 0::2], sin[:, 0::2]], dim=-1).reshape(cos.shape[0], cos.shape[1]).contiguous()


def qkv_epilogue(wq=None, wk=None, rope=None, *, eps=1e-7, heads, rope_row0=0, qscale=0.0):
    """epilogue of one QKV-projection problem of `gemm_grouped_qkv` (ldc_qkv_epilogue); rope = `compact_rope_table(cos, sin)`;
    returns (struct, keep-alive tuple)"""
    _dev(wq, wk, rope)
    pv = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    return QkvEpilogue(pv(wq), pv(wk), pv(rope), None, eps, qscale, heads, rope_row0), (wq, wk, rope)


ERR_UNSUPPORTED = -3  # LDC_ERR_UNSUPPORTED
_DEFAULT_QSCALE = ctypes.c_float(0.08838834764831845 * 1.4426950408889634).value  # log2(e) / sqrt(128) as the library rounds it


def gemm_grouped_qkv_f32(problems, epilogues):
    """exact-fp32 `gemm_grouped` where problem i with a non-None epilogue is a QKV projection whose epilogue applies the per-head RMSNorm
    and the rotary embedding (ldc_gemm_grouped_qkv_f32; epilogues built with qscale=1).  Returns False - and launches nothing - when the
    ring kernel does not serve the launch (LDC_ERR_UNSUPPORTED): the caller then runs `gemm_grouped` + `qk_rmsnorm_rope`."""
    n = len(problems)
    if not 1 <= n <= MAX_GROUPED or len(epilogues) != n:
        raise ValueError(f"gemm_grouped_qkv_f32 takes 1..{MAX_GROUPED} problems and one epilogue (or None) per problem")
    arr = (GemmProblem * n)(*[p[0] for p in problems])
    epi = (QkvEpilogue * n)(*[(e[0] if e is not None else QkvEpilogue()) for e in epilogues])
    ws = _grouped_workspace(problems[0][1][2].device)
    st = lib.ldc_gemm_grouped_qkv_f32(arr, epi, n, c_void_p(ws.data_ptr()), ws.numel() * 4, _stream())
    if st == ERR_UNSUPPORTED:
        return False
    _check(st, "ldc_gemm_grouped_qkv_f32")
    return True


def gemm_grouped_qkv(problems, epilogues, force_fallback=False):
    """`gemm_grouped(split_bf16=True)` where problem i with a non-None epilogue is a fused QKV projection whose C receives the
    attention operand rows of `attn_fwd_split` (bias -> per-head RMSNorm -> rotary embedding -> q scale -> hi / lo split).
    When the fused kernel does not serve the launch (LDC_ERR_UNSUPPORTED: K not a whole k-step, more tiles than hand-off counters)
    this runs what include/ladcast_hip.h documents for that status - the plain grouped GEMM, then `ldc_attn_qkv_prepare_split` in
    place on every QKV problem's output (split-bf16 mode only; `force_fallback` takes that route unconditionally, for tests)."""
    n = len(problems)
    if not 1 <= n <= MAX_GROUPED or len(epilogues) != n:
        raise ValueError(f"gemm_grouped_qkv takes 1..{MAX_GROUPED} problems and one epilogue (or None) per problem")
    arr = (GemmProblem * n)(*[p[0] for p in problems])
    epi = (QkvEpilogue * n)(*[(e[0] if e is not None else QkvEpilogue()) for e in epilogues])
    ws = _grouped_workspace(problems[0][1][2].device)
    st = ERR_UNSUPPORTED if force_fallback else lib.ldc_gemm_grouped_bf16x3_qkv(arr, epi, n, c_void_p(ws.data_ptr()), ws.numel() * 4, _stream())
    if st != ERR_UNSUPPORTED:
        _check(st, "ldc_gemm_grouped_bf16x3_qkv")
        return
    if any(p[0].d.flags & GEMM_BF16_1TERM for p in problems):
        raise RuntimeError("ldc_gemm_grouped_bf16x3_qkv: shape not served by the fused kernel, and the plain-GEMM + ldc_attn_qkv_prepare_split "
                           "route only writes split-bf16 operand rows (status -3 in the single-term bf16 mode)")
    _check(lib.ldc_gemm_grouped_bf16x3(arr, n, c_void_p(ws.data_ptr()), ws.numel() * 4, _stream()), "ldc_gemm_grouped_bf16x3 (QKV fallback)")
    for (p, keep), e in zip(problems, epilogues):
        if e is None:
            continue
        q, (wq, wk, rope) = e
        if q.qscale not in (0.0, _DEFAULT_QSCALE):  # ldc_attn_qkv_prepare_split applies the default scale only: never two answers for one call
            raise RuntimeError("ldc_gemm_grouped_bf16x3_qkv: shape not served by the fused kernel, and the plain-GEMM + ldc_attn_qkv_prepare_split "
                               f"route applies the default q scale log2(e)/sqrt(128), not qscale={q.qscale!r}")
        d, C = p.d, keep[2]
        D, S = q.heads * 128, d.M
        if d.N != 3 * D:
            raise RuntimeError("ldc_gemm_grouped_bf16x3_qkv failed with status -1")  # what the fused launch reports for a wrong width
        cos = sin = None
        if rope is not None:  # the compact (cos_i, sin_i) table back in the [rows][128] form of ldc_attn_qkv_prepare_split
            t = rope.reshape(-1, 64, 2)[q.rope_row0 : q.rope_row0 + S]
            cos, sin = t[..., 0].repeat_interleave(2, dim=1).contiguous(), t[..., 1].repeat_interleave(2, dim=1).contiguous()
        base = C.data_ptr()
        _check(lib.ldc_attn_qkv_prepare_split(c_void_p(base), c_void_p(base + 4 * D), c_void_p(base + 8 * D), d.batch, S, q.heads, d.ldc, d.c_bs, S,
                                              _p(wq), _p(wk), _p(cos), _p(sin), None, None, None, None, q.eps, _stream()), "ldc_attn_qkv_prepare_split")


def attn_qkv_prepare_split(Q, K, V, *, B, S, H, ld_qkv, qkv_bs, split_row, seg0=(None, None, None, None), seg1=(None, None, None, None), eps=1e-7):
    """fp32 q / k / v views of a fused buffer -> the operand rows of `attn_fwd_split`, in place (segN = (wq, wk, cos, sin))"""
    _dev(Q, K, V, *seg0, *seg1)
    _check(lib.ldc_attn_qkv_prepare_split(_p(Q), _p(K), _p(V), B, S, H, ld_qkv, qkv_bs, split_row, _p(seg0[0]), _p(seg0[1]), _p(seg0[2]), _p(seg0[3]),
                                          _p(seg1[0]), _p(seg1[1]), _p(seg1[2]), _p(seg1[3]), eps, _stream()), "ldc_attn_qkv_prepare_split")


_attn_ws = {}


def _attn_workspace(device, nbytes):
    """scratch of the attention's TAIL schedule (more units than CUs), one per (device, stream), allocated ONCE at the largest size any call
    shape can ask for (ldc_attn_fwd_split_workspace_max_bytes, 17.8 MB) and never replaced: its pointer is baked into every hipGraph
    captured on that stream, and graphs of several batch sizes stay alive side by side (a regrown workspace would leave the older graphs
    writing their partials into freed memory).  A first call inside a capture would take the block from the graph's private pool and
    publish it to eager callers: refused - samplers warm up on their capture stream first, like the GEMM workspace."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _attn_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("attn_fwd_split: the first call on a stream must be made outside a graph capture (run one warm-up forward)")
        ws = torch.empty((lib.ldc_attn_fwd_split_workspace_max_bytes() + 3) // 4, device=device, dtype=torch.float32)
        _attn_ws[key] = ws
    if ws.numel() * 4 < nbytes:
        raise RuntimeError(f"attn_fwd_split: call shape asks for {nbytes} workspace bytes, more than ldc_attn_fwd_split_workspace_max_bytes")
    return ws


def attn_fwd_split(Q, K, V, O, *, B, S, H, ld_qkv, qkv_bs, ldo, o_bs, out_split=False, one_term=False, key_bias=None, use_workspace=True):
    """attention on row-major split-bf16 operand rows (ldc_attn_fwd_split); key_bias: additive score bias per key, padded to a multiple
    of 32 entries (`pad_key_bias`), or None.  use_workspace=False: never the TAIL schedule (A/B, tests)"""
    _dev(Q, K, V, O, key_bias)
    nbytes = lib.ldc_attn_fwd_split_workspace_bytes(B, S, H) if use_workspace else 0
    ws = _attn_workspace(Q.device, nbytes) if nbytes > 0 else None
    if key_bias is not None and key_bias.numel() < 32 * ((S + 31) // 32):
        raise ValueError("key_bias must be padded to a multiple of 32 keys (hip.pad_key_bias)")
    flags = (ATTN_OUT_BF16 if int(out_split) == FMT_BF16 else ATTN_OUT_SPLIT if out_split else 0) | (ATTN_BF16_1TERM if one_term else 0)
    _check(lib.ldc_attn_fwd_split(_p(Q), _p(K), _p(V), _p(O), B, S, H, ld_qkv, qkv_bs, ldo, o_bs, _p(key_bias), flags, _p(ws), 0 if ws is None else ws.numel() * 4,
                                  _stream()), "ldc_attn_fwd_split")


def pad_key_bias(bias):
    """[S] -> [32 * ceil(S / 32)] (zero-padded, contiguous): the form `attn_fwd_split` reads 16 bytes at a time"""
    S = bias.numel()
    out = torch.zeros(32 * ((S + 31) // 32), device=bias.device, dtype=torch.float32)
    out[:S] = bias.reshape(-1)
    return out


def qk_rmsnorm_rope(q, k, *, B, row0, rows, H, ld, bs, wq, wk, eps, cos=None, sin=None):
    _dev(q, k, wq, wk, cos, sin)
    _check(lib.ldc_qk_rmsnorm_rope(_p(q), _p(k), B, row0, rows, H, ld, bs, _p(wq), _p(wk), eps, _p(cos), _p(sin), _stream()),
           "ldc_qk_rmsnorm_rope")


def layernorm_mod(x, y, *, B, rows, D, ldx, x_bs, ldy, y_bs, scale=None, shift=None, mod_bs=0, mode=0, eps=1e-6, out_split=False,
                  split_row=None, scale2=None, shift2=None):
    """split_row: rows [split_row, rows) take scale2 / shift2 (two token streams normalised by one launch)"""
    _dev(x, y, scale, shift, scale2, shift2)
    if split_row is None:
        _check(lib.ldc_layernorm_mod(_p(x), _p(y), B, rows, D, ldx, x_bs, ldy, y_bs, _p(scale), _p(shift), mod_bs, mode, eps,
                                     int(out_split), _stream()),
               "ldc_layernorm_mod")
    else:
        _check(lib.ldc_layernorm_mod2(_p(x), _p(y), B, rows, D, ldx, x_bs, ldy, y_bs, _p(scale), _p(shift), split_row, _p(scale2), _p(shift2),
                                      mod_bs, mode, eps, int(out_split), _stream()),
               "ldc_layernorm_mod2")


def mean_rows(x, y, *, B, rows, D, ldx, x_bs, x_split=None, lds=None, s_bs=None, fmt=FMT_SPLIT):
    """x_split: also write x in the split (fmt FMT_SPLIT) or plain-bf16 (FMT_BF16) activation format to this buffer (row stride lds,
    batch stride s_bs, in floats)"""
    _dev(x, y, x_split)
    if x_split is None:
        _check(lib.ldc_mean_rows(_p(x), _p(y), B, rows, D, ldx, x_bs, _stream()), "ldc_mean_rows")
    else:
        _check(lib.ldc_mean_rows_split(_p(x), _p(y), _p(x_split), B, rows, D, ldx, x_bs, lds if lds is not None else ldx,
                                       s_bs if s_bs is not None else x_bs, int(fmt), _stream()), "ldc_mean_rows_split")


def gate_residual(resid, y, gate, out, *, B, rows, D, ld_res, res_bs, ld_y, y_bs, gate_bs):
    _dev(resid, y, gate, out)
    _check(lib.ldc_gate_residual(_p(resid), _p(y), _p(gate), _p(out), B, rows, D, ld_res, res_bs, ld_y, y_bs, gate_bs, _stream()),
           "ldc_gate_residual")


def chan_to_token(x, out, *, B, C, N, ldo, fill_cols=None, out_split=False):
    _dev(x, out)
    _check(lib.ldc_chan_to_token_split(_p(x), _p(out), B, C, N, ldo, ldo if fill_cols is None else fill_cols, int(out_split), _stream()),
           "ldc_chan_to_token")


def token_to_chan(x, out, *, B, C, N, ldi):
    _dev(x, out)
    _check(lib.ldc_token_to_chan(_p(x), _p(out), B, C, N, ldi, _stream()), "ldc_token_to_chan")


def timestep_embedding(t, out, n):
    _dev(t, out)
    _check(lib.ldc_timestep_embedding(_p(t), _p(out), n, _stream()), "ldc_timestep_embedding")


def temb_modulate(temb, te, *, B, D, te_rows):
    _dev(temb, te)
    _check(lib.ldc_temb_modulate(_p(temb), _p(te), B, D, te_rows, _stream()), "ldc_temb_modulate")


def chan_affine(x, y, mean, std, target_std, *, outer, C, inner, inverse):
    _dev(x, y, mean, std)
    _check(lib.ldc_chan_affine(_p(x), _p(y), _p(mean), _p(std), target_std, outer, C, inner, int(inverse), _stream()), "ldc_chan_affine")


def edm_scale_f64_to_f32(x, c_in, out):
    _dev(x, out)
    _check(lib.ldc_edm_scale_f64_to_f32(_p(x), c_in, _p(out), x.numel(), _stream()), "ldc_edm_scale_f64_to_f32")


def edm_init_state(noise, sigma0, x):
    _dev(noise, x)
    _check(lib.ldc_edm_init_state(_p(noise), sigma0, _p(x), x.numel(), _stream()), "ldc_edm_init_state")


def edm_churn(x_cur, noise, coef, x_hat):
    """x_hat = x_cur + coef * noise (fp64; the stochastic branch of the EDM sampler)"""
    _dev(x_cur, noise, x_hat)
    if noise.dtype != torch.float64 or x_cur.dtype != torch.float64:
        raise TypeError("edm_churn works on the fp64 sampler state")
    _check(lib.ldc_edm_churn(_p(x_cur), _p(noise), coef, _p(x_hat), x_cur.numel(), _stream()), "ldc_edm_churn")


def edm_euler(x_hat, F, c_skip, c_out, t_hat, dt, x_next, d_cur):
    _dev(x_hat, F, x_next, d_cur)
    _check(lib.ldc_edm_euler(_p(x_hat), _p(F), c_skip, c_out, t_hat, dt, _p(x_next), _p(d_cur), x_hat.numel(), _stream()), "ldc_edm_euler")


def edm_heun(x_hat, x_next, F, d_cur, c_skip, c_out, t_next, dt):
    _dev(x_hat, x_next, F, d_cur)
    _check(lib.ldc_edm_heun(_p(x_hat), _p(x_next), _p(F), _p(d_cur), c_skip, c_out, t_next, dt, x_hat.numel(), _stream()), "ldc_edm_heun")


def f64_to_f32(x, y):
    _dev(x, y)
    _check(lib.ldc_f64_to_f32(_p(x), _p(y), x.numel(), _stream()), "ldc_f64_to_f32")


def dpm_step(sample, F, m1, x0, prev, c_skip, c_out, a, b, inv_r0, order):
    _dev(sample, F, m1, x0, prev)
    _check(lib.ldc_dpm_step(_p(sample), _p(F), _p(m1), _p(x0), _p(prev), c_skip, c_out, a, b, inv_r0, order, sample.numel(), _stream()),
           "ldc_dpm_step")


def ddim_step(sample, F, noise, x0, prev, sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, dir_coef, std_dev, clip_range, prediction_type, reclip):
    _dev(sample, F, noise, x0, prev)
    _check(lib.ldc_ddim_step(_p(sample), _p(F), _p(noise), _p(x0), _p(prev), sqrt_alpha_t, sqrt_beta_t, sqrt_alpha_prev, dir_coef, std_dev,
                             clip_range, prediction_type, int(bool(reclip)), sample.numel(), _stream()), "ldc_ddim_step")


def ddpm_step(sample, F, noise, x0, prev, sqrt_alpha_t, sqrt_beta_t, x0_coef, sample_coef, std_dev, clip_range, prediction_type):
    _dev(sample, F, noise, x0, prev)
    _check(lib.ldc_ddpm_step(_p(sample), _p(F), _p(noise), _p(x0), _p(prev), sqrt_alpha_t, sqrt_beta_t, x0_coef, sample_coef, std_dev, clip_range,
                             prediction_type, sample.numel(), _stream()), "ldc_ddpm_step")


def scale_f32(x, s, y):
    _dev(x, y)
    _check(lib.ldc_scale_f32(_p(x), s, _p(y), x.numel(), _stream()), "ldc_scale_f32")


def axpby_f32(x, a, y, b, out):
    _dev(x, y, out)
    _check(lib.ldc_axpby_f32(_p(x), a, _p(y), b, _p(out), x.numel(), _stream()), "ldc_axpby_f32")


# -- DCAE (NHWC) -------------------------------------------------------------------------------
def sphere_conv_nhwc(X, Wt, Y, *, B, H, W, cin, cout, ldx=None, ldy=None, bias=None, R=None, ldr=0, ksize=3, act=ACT_NONE):
    _dev(X, Wt, Y, bias, R)
    _check(lib.ldc_sphere_conv_nhwc(_p(X), _p(Wt), _p(bias), _p(R), _p(Y), B, H, W, cin, ldx if ldx is not None else cin, cout,
                                    ldy if ldy is not None else cout, ldr, ksize, act, _stream()), "ldc_sphere_conv_nhwc")


def conv_cin_padded(cin, cpk=32):
    """channels per tap of a packed conv weight: cpk * 2^j >= cin (cpk = channels per k-step: 32 split-bf16, 64 plain bf16)"""
    kt = -(-cin // cpk)
    p2 = 1
    while p2 < kt:
        p2 *= 2
    return cpk * p2


def sphere_conv_nhwc_split(X, Wp, Y, *, B, H, W, cin, cout, ldx, ldy=None, bias=None, R=None, ldr=0, ksize=3, act=ACT_NONE, in_fmt=FMT_SPLIT,
                           out_fmt=FMT_F32):
    """X: operand rows (FMT_SPLIT: Wp = pack_weight_bf16x2 of the [cout, k*k*conv_cin_padded(cin)] tap-major weight, zeros behind cin | FMT_BF16: Wp = pack_weight_bf16 of the taps padded to 64 * 2^j channels;
    ldx % 8 == 0); Y fp32 rows or (out_fmt = in_fmt) operand rows"""
    _dev(X, Wp, Y, bias, R)
    ws = _grouped_workspace(X.device)
    _check(lib.ldc_sphere_conv_nhwc_split(_p(X), _p(Wp), _p(bias), _p(R), _p(Y), B, H, W, cin, ldx, cout, ldy if ldy is not None else cout,
                                          ldr, ksize, act, int(in_fmt), int(out_fmt), _p(ws), ws.numel() * 4, _stream()), "ldc_sphere_conv_nhwc_split")


def sphere_conv_plan(B, H, W, cin, cout, ksize=3, in_fmt=FMT_SPLIT):
    """(halo, tile_rows, tile_w): which kernel `sphere_conv_nhwc_split` runs for the shape (ldc_sphere_conv_plan)"""
    bm, tw = c_int(0), c_int(0)
    halo = lib.ldc_sphere_conv_plan(B, H, W, cin, cout, ksize, int(in_fmt), ctypes.byref(bm), ctypes.byref(tw))
    return bool(halo), bm.value, tw.value


def split_rows(x, ys, *, rows, C, ldx=None, lds=None, fmt=FMT_SPLIT):
    _dev(x, ys)
    _check(lib.ldc_split_rows(_p(x), _p(ys), rows, C, ldx if ldx is not None else C, lds if lds is not None else -(-C // 8) * 8, int(fmt), _stream()),
           "ldc_split_rows")


def sphere_dwconv_nhwc(x, wt, y, *, B, H, W, C, ldx=None, ldy=None, bias=None, ksize=3, glu=False, out_fmt=FMT_F32):
    _dev(x, wt, y, bias)
    cy = C // 2 if glu else C
    _check(lib.ldc_sphere_dwconv_nhwc_fmt(_p(x), _p(wt), _p(bias), _p(y), B, H, W, C, ldx if ldx is not None else C,
                                          ldy if ldy is not None else cy, ksize, int(glu), int(out_fmt), _stream()), "ldc_sphere_dwconv_nhwc_fmt")


def grouped_conv1x1_nhwc(x, wt, y, *, M, groups, ldx, ldy):
    _dev(x, wt, y)
    _check(lib.ldc_grouped_conv1x1_nhwc(_p(x), _p(wt), _p(y), M, groups, ldx, ldy, _stream()), "ldc_grouped_conv1x1_nhwc")


def relu_linear_attn_nhwc(qkv, y, *, B, P, groups, ldq, ldy, eps, out_fmt=FMT_F32, sliced=True):
    """ReLU linear attention over consecutive 96-channel (q | k | v) groups.  sliced (default): 128-pixel slices, partial KV matrices
    in a scratch buffer of ldc_relu_linear_attn_workspace_bytes, two launches; sliced=False: one workgroup per (frame, group)"""
    _dev(qkv, y)
    ws, nbytes = None, 0
    if sliced:
        nbytes = lib.ldc_relu_linear_attn_workspace_bytes(B, P, groups)  # 0 below 1024 pixels: one launch there
        ws = torch.empty(nbytes // 4, device=qkv.device, dtype=torch.float32) if nbytes else None
    _check(lib.ldc_relu_linear_attn_nhwc_fmt(_p(qkv), _p(y), B, P, groups, ldq, ldy, eps, int(out_fmt), _p(ws), nbytes,
                                             _stream()),
           "ldc_relu_linear_attn_nhwc_fmt")


def rmsnorm_rows(x, w, y, *, rows, C, eps, b=None, resid=None, ldx=None, ldr=None, ldy=None, act=ACT_NONE, ys=None, lds=None, fmt=FMT_SPLIT):
    """y: fp32 rows and / or ys: operand rows (fmt = FMT_SPLIT | FMT_BF16, lds = C rounded up to 8 by default)"""
    _dev(x, w, b, resid, y, ys)
    _check(lib.ldc_rmsnorm_rows_split(_p(x), _p(w), _p(b), _p(resid), _p(y), _p(ys), rows, C, ldx if ldx is not None else C,
                                      ldr if ldr is not None else C, ldy if ldy is not None else C,
                                      (lds if lds is not None else -(-C // 8) * 8) if ys is not None else 0, int(fmt), eps, act, _stream()),
           "ldc_rmsnorm_rows_split")


def pixel_unshuffle_shortcut(cv, x, y, *, B, H2, W2, cout, cin, ys=None, lds=None, fmt=FMT_SPLIT):
    _dev(cv, x, y, ys)
    _check(lib.ldc_pixel_unshuffle_shortcut_split(_p(cv), _p(x), _p(y), _p(ys), B, H2, W2, cout, cin,
                                                  (lds if lds is not None else -(-cout // 8) * 8) if ys is not None else 0, int(fmt), _stream()),
           "ldc_pixel_unshuffle_shortcut_split")


def pixel_shuffle_to_chan(cv, out, *, B, H, W, cout, keep):
    """cv [B*H*W, 4*cout] NHWC conv rows -> out [B, keep, 2H, 2W] NCHW: pixel_shuffle without shortcut, first `keep` channels (ldc_pixel_shuffle_to_chan)"""
    _dev(cv, out)
    _check(lib.ldc_pixel_shuffle_to_chan(_p(cv), _p(out), B, H, W, cout, keep, _stream()), "ldc_pixel_shuffle_to_chan")


def pixel_shuffle_shortcut(cv, x, y, *, B, H, W, cout, cin, ys=None, lds=None, fmt=FMT_SPLIT):
    _dev(cv, x, y, ys)
    _check(lib.ldc_pixel_shuffle_shortcut_split(_p(cv), _p(x), _p(y), _p(ys), B, H, W, cout, cin,
                                                (lds if lds is not None else -(-cout // 8) * 8) if ys is not None else 0, int(fmt), _stream()),
           "ldc_pixel_shuffle_shortcut_split")


def upsample_nearest2x_rows(x, y, *, B, H, W, C, ldx=None, ldy=None, ys=None, lds=None, fmt=FMT_SPLIT):
    """NHWC rows [B*H*W, ldx] -> [B*2H*2W, C] nearest-neighbour x2 (F.interpolate(scale_factor=2, mode="nearest")): fp32 rows `y` and / or operand rows `ys`"""
    _dev(x, y, ys)
    _check(lib.ldc_upsample_nearest2x_rows(_p(x), _p(y), _p(ys), B, H, W, C, ldx if ldx is not None else C, ldy if ldy is not None else C,
                                           (lds if lds is not None else (ys.shape[1] if ys is not None else 0)), int(fmt), _stream()), "ldc_upsample_nearest2x_rows")


def chan_regroup(x, y, *, M, cin, cout):
    _dev(x, y)
    _check(lib.ldc_chan_regroup(_p(x), _p(y), M, cin, cout, _stream()), "ldc_chan_regroup")
