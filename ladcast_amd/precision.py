"""The three arithmetic modes of the hot path and their STATED tolerances - the one place they are defined.

``set_gemm_precision(mode)`` of both models, ``bench.py --precision``, ``include/ladcast_hip.h`` and the parity tests all refer to
this table; the numbers are rel-L2 against the fp32 CPU oracle on identical weights and inputs.

* ``fp32``   - exact-fp32 matrix cores everywhere: the reference's own arithmetic and, since round 6, the mode ``bench.py`` reports as ``value``.
* ``bf16x3`` - split-bf16 contraction (hi*hi + hi*lo + lo*hi, fp32 accumulation); the fast mode (``bench.py``'s ``bf16x3_mode`` block).  Both stay inside the north
  star's 1e-4 budget per sampler chunk / rollout.
* ``bf16``   - the mixed-precision mode of BASELINE configs[4] ("fp16/bf16 mixed"): ONE bf16 MFMA per product on operands rounded to
  bf16; residual stream, norms, softmax statistics, (B, D)-vector Linears, the DCAE's linear attention and the sampler state stay
  fp32 / fp64.  Outside the 1e-4 budget by construction (2^-9 per operand rounding).  Its entries are measured on the GPU
  (profiles/r03_*_gpu_tests.log) x 2, per stage, and the tests additionally require the mode to be at least as close to the fp32
  oracle as the oracle itself run under the reference's own mixed precision (``oracle/autocast.py``: torch.autocast(bfloat16) with
  the fp32 islands of models/LaDCast_3D_model.py:953 and models/DCAE.py:162,180).
"""

MODES = ("fp32", "bf16x3", "bf16")

STATED_TOLERANCE = {
    "fp32": {"forward": 2e-5, "chunk": 1e-4, "rollout": 1e-4, "dcae": 1e-4},
    "bf16x3": {"forward": 3e-5, "chunk": 1e-4, "rollout": 1e-4, "dcae": 1e-4},
    "bf16": {
        # AR transformer (375M and tiny widths): one forward; network output inside a sampler chunk (its input has drifted); the
        # sample of a 20-step chunk with the 39-forward Heun sampler / the 20-forward DPM-Solver++(2M) loop
        "forward": 7e-3,
        # the same forward on weights pushed to trained-model statistics (tests/synth.py::stress_ar_: peaked softmax, |logit| ~ 60, outlier
        # channels x50, AdaLN gates O(1)): measured 7.6e-3 (profiles/r04_*_gpu_tests.log) x 2.  fp32 / bf16x3 keep their "forward" entries there
        # (measured 1.8e-6 / 1.4e-5)
        "forward_trained_statistics": 1.5e-2,
        "chunk_network_output": 1e-2,
        "chunk_edm": 4e-3,
        "chunk_pipeline": 8e-3,
        # DCAE, one frame each way (tiny and full size)
        "dcae_encode": 2e-2,
        "dcae_decode": 1.2e-2,
        # end to end: encode -> chained chunks -> decode, decoded fields of chunk c (every chunk, no growth allowed beyond it)
        "rollout_decoded": 2.5e-2,
    },
}


def tolerance(mode: str, stage: str) -> float:
    if mode not in STATED_TOLERANCE:
        raise ValueError(f"precision mode must be one of {MODES}")
    return STATED_TOLERANCE[mode][stage]
