"""SphereConv2d 3x3 implicit GEMM: exact-fp32 MFMA kernel vs the pre-split kernel (split rows in; halo-staged or gathered as ldc_sphere_conv_plan
decides) at the DCAE's layer shapes (sustained).  LDC_CONV_SMALL_TILES=<n> (A/B build): tile-height cross-over of the gathered kernel."""
import os, sys, time
sys.path.insert(0, os.environ.get("LDC_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
from ladcast_amd.models.sphere_conv import pack_dense_weight, pack_dense_weight_bf16x3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (ci, co, H, W) in [(252, 252, 120, 240), (504, 504, 60, 120), (504, 504, 30, 60), (1008, 1008, 15, 30)]:
    w = torch.randn(co, ci, 3, 3, device="cuda") / (9 * ci) ** 0.5
    x = torch.randn(B * H * W, ci, device="cuda"); y = torch.empty(B * H * W, co, device="cuda"); bias = torch.randn(co, device="cuda")
    wf, wp = pack_dense_weight(w), pack_dense_weight_bf16x3(w)
    f32 = lambda: hip.sphere_conv_nhwc(x, wf, y, B=B, H=H, W=W, cin=ci, cout=co, bias=bias, ksize=3)
    c8 = -(-ci // 8) * 8
    xs = torch.empty(B * H * W, c8, device="cuda"); hip.split_rows(x, xs, rows=B * H * W, C=ci)
    sp = lambda: hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=bias, ksize=3)
    res = []
    for fn in (f32, sp):
        t_end = time.time() + 0.7
        while time.time() < t_end:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 1e3 / 50)
    fl = 2.0 * B * H * W * co * 9 * ci
    print(f"B={B} {ci}->{co} @ {H}x{W}: fp32 {res[0]:8.1f} us {fl / res[0] / 1e6:6.1f} TF/s | pre-split bf16x3 {res[1]:8.1f} us {fl / res[1] / 1e6:6.1f} TF/s")
