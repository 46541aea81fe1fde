"""Pre-split SphereConv2d 3x3 implicit GEMM (ldc_sphere_conv_nhwc_split) at the DCAE's layer shapes: sustained time per launch and the
relative error against the exact-fp32 conv kernel of the same library.  Run once per library variant (LDC_LIB_PATH=...): the
measurement variants of `make variant` (k-step order, no-DMA diagnostic) are compared this way, in one gpurun call."""
import os, sys, time
sys.path.insert(0, os.environ.get("LDC_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
from ladcast_amd.models.sphere_conv import pack_dense_weight, pack_dense_weight_bf16x3

tag = os.path.basename(os.environ.get("LDC_LIB_PATH", "libladcast_hip.so"))
shapes = [(252, 252, 120, 240), (504, 504, 60, 120), (504, 1008, 60, 120), (504, 504, 30, 60), (1008, 1008, 15, 30), (252, 89, 120, 240)]
for B in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]:
    for (ci, co, H, W) in shapes:
        torch.manual_seed(0)
        w = torch.randn(co, ci, 3, 3, device="cuda") / (9 * ci) ** 0.5
        x = torch.randn(B * H * W, ci, device="cuda"); y = torch.empty(B * H * W, co, device="cuda"); bias = torch.randn(co, device="cuda")
        yr = torch.empty_like(y)
        wf, wp = pack_dense_weight(w), pack_dense_weight_bf16x3(w)
        c8 = -(-ci // 8) * 8
        xs = torch.empty(B * H * W, c8, device="cuda"); hip.split_rows(x, xs, rows=B * H * W, C=ci)
        sp = lambda: hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=bias, ksize=3)
        hip.sphere_conv_nhwc(x, wf, yr, B=B, H=H, W=W, cin=ci, cout=co, bias=bias, ksize=3)
        sp(); torch.cuda.synchronize()
        err = ((y - yr).double().norm() / yr.double().norm()).item()
        t_end = time.time() + 0.7
        while time.time() < t_end:
            for _ in range(10): sp()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): sp()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 50
        fl = 2.0 * B * H * W * co * 9 * ci
        print(f"{tag:28s} B={B} {ci:4d}->{co:4d} @ {H:3d}x{W:3d}: {us:8.1f} us {fl / us / 1e6:6.1f} TF/s  rel-L2 vs fp32 kernel {err:.2e}", flush=True)
