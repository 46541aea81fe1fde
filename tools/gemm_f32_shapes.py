"""Exact-fp32 ring GEMM (gemm_bf16x3_v3_kernel<BM, 0, false>) on the 375M / 1.6B models' launch shapes, one line per grouped launch:
us per launch and TFLOP/s after 2 s of settling.  A/B aid: with the A/B library (make -C ladcast_amd/csrc ab; LDC_LIB_PATH=...)
LDC_F32_RING_BM=256 forces 256-row tiles.  usage: python tools/gemm_f32_shapes.py [1.6B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ladcast_amd.hip as hip

D, F = (2048, 8192) if "1.6B" in sys.argv else (1536, 6144)
SHAPES = {
    "refiner qkv": [(450, 3 * D, D)], "dual qkv": [(1800, 3 * D, D), (450, 3 * D, D)], "dual out": [(1800, D, D), (450, D, D)],
    "dual ff up": [(1800, F, D), (450, F, D)], "dual ff down": [(1800, D, F), (450, D, F)],
    "single qkv+mlp": [(2250, F, D), (2250, 3 * D, D)], "single out": [(2250, D, D + F)], "square 4096 x 2048": [(4096, 2048, D)],
}
WARM_S = float(os.environ.get("WARM_S", "2.0"))
print(f"D = {D}, LDC_F32_RING_BM = {os.environ.get('LDC_F32_RING_BM', '(128)')}, library {os.environ.get('LDC_LIB_PATH', 'shipped')}")
tot_us = tot_fl = 0.0
for name, probs in SHAPES.items():
    flops = sum(2.0 * M * N * K for M, N, K in probs)
    ops = [(torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")) for M, N, K in probs]
    ps = [hip.gemm_problem(a, w, torch.empty(a.shape[0], w.shape[0], device="cuda"), M=a.shape[0], N=w.shape[0], K=a.shape[1]) for a, w in ops]
    fn = lambda: hip.gemm_grouped(ps, split_bf16=False)  # noqa: E731
    t_end = time.time() + WARM_S
    while time.time() < t_end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 10.0
    print(f"  {name:20s} {str(probs):50s} {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s")
    if not name.startswith("square"):
        tot_us, tot_fl = tot_us + us, tot_fl + flops
print(f"  model shapes together: {tot_us:.1f} us, {tot_fl / tot_us / 1e6:.1f} TFLOP/s")
