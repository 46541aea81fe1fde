"""The reference's mixed-precision execution, restated for the CPU oracle (TEST INFRASTRUCTURE, see ``oracle/__init__.py``).

BASELINE configs[4] names a "fp16/bf16 mixed" run.  In the reference that is ``torch.autocast`` around the unchanged fp32-weight
models (``configs/ladcast_375M.yaml:74`` ``mixed_precision: bf16``; ``train_new_encdec.py:231,312``) with three fp32 islands the
authors wrote by hand as ``torch.autocast(device_type, torch.float32)``: the conditioning embedding of the transformer
(``models/LaDCast_3D_model.py:953-969``) and the two attention products of the DCAE (``models/DCAE.py:162-175,180-186``).

``reference_autocast(policy)`` runs oracle code that way:

* ``policy="cpu"`` is literally ``torch.autocast("cpu", torch.bfloat16)``: torch's own CPU cast lists decide which ops round to bf16.
* ``policy="cuda"`` (default) additionally applies the one CUDA cast-list rule that matters on this path and that the CPU list lacks:
  ``layer_norm`` runs in fp32 and returns fp32 (``torch/csrc/autocast_mode.cpp``: CUDA "fp32" list), so the AdaLN modulation
  ``LN(x) * (1 + scale) + shift`` is formed in fp32 as it is on the reference's device.  (The other fp32-list ops on the path -
  ``pow`` / ``rsqrt`` of RMSNorm, softmax inside SDPA - are already written with explicit fp32 statistics by diffusers / fused.)

``fp32_island()`` is the hand-written island.  On the torch releases the reference was developed with (< 2.4) a nested CUDA
``autocast(dtype=float32)`` stays enabled and casts the inputs of the autocast-eligible ops UP to fp32; newer releases disable
autocast in that context instead, which makes ``F.linear(bf16 activation, fp32 weight)`` raise.  The only semantics under which
the reference runs at all is therefore "fp32 inputs, fp32 arithmetic" - the oracle's call sites enter ``fp32_island()`` and
``.float()`` the tensors that cross into it.  Outside ``reference_autocast`` both are no-ops (fp32 stays fp32 bit for bit).
"""
from __future__ import annotations

import contextlib
import warnings

import torch
import torch.nn.functional as F

_POLICY = None  # None (plain fp32 oracle) | "cpu" | "cuda"


def active() -> bool:
    return _POLICY is not None


@contextlib.contextmanager
def fp32_island():
    """``with torch.autocast(device_type, torch.float32):`` of the reference (see the module docstring)"""
    if _POLICY is None:
        yield
        return
    with torch.autocast("cpu", enabled=False):
        yield


@contextlib.contextmanager
def reference_autocast(policy: str = "cuda", dtype: torch.dtype = torch.bfloat16):
    global _POLICY
    if policy not in ("cpu", "cuda"):
        raise ValueError("policy must be 'cpu' or 'cuda'")
    if _POLICY is not None:
        raise RuntimeError("reference_autocast does not nest")
    orig_ln = F.layer_norm

    def layer_norm_fp32(input, normalized_shape, weight=None, bias=None, eps=1e-5):
        with torch.autocast("cpu", enabled=False):
            return orig_ln(input.float(), normalized_shape, None if weight is None else weight.float(),
                           None if bias is None else bias.float(), eps)

    _POLICY = policy
    if policy == "cuda":
        F.layer_norm = layer_norm_fp32  # nn.LayerNorm.forward looks the function up at call time
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with torch.autocast("cpu", dtype=dtype):
                yield
    finally:
        F.layer_norm = orig_ln
        _POLICY = None
