"""One SphereConv2d 3x3 shape through ldc_sphere_conv_nhwc_split, a few launches: the program the PMC passes of tools/conv_pmc.sh wrap.
usage: conv_one.py cin cout H W [frames] [launches]"""
import os, sys
sys.path.insert(0, os.environ.get("LDC_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
from ladcast_amd.models.sphere_conv import pack_dense_weight_bf16x3

ci, co, H, W = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
torch.manual_seed(0)
w = torch.randn(co, ci, 3, 3, device="cuda") / (9 * ci) ** 0.5
x = torch.randn(B * H * W, ci, device="cuda"); y = torch.empty(B * H * W, co, device="cuda"); bias = torch.randn(co, device="cuda")
wp = pack_dense_weight_bf16x3(w)
c8 = -(-ci // 8) * 8
xs = torch.empty(B * H * W, c8, device="cuda"); hip.split_rows(x, xs, rows=B * H * W, C=ci)
for _ in range(n):
    hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=bias, ksize=3)
torch.cuda.synchronize()
alg = B * H * W * (c8 * 4 + co * 4) + wp.numel() * wp.element_size()
print(f"algorithmic bytes per launch: {alg / 1e6:.1f} MB (split rows in {B * H * W * c8 * 4 / 1e6:.1f} + fp32 rows out {B * H * W * co * 4 / 1e6:.1f} + packed weight {wp.numel() * wp.element_size() / 1e6:.1f})")
