import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
from ladcast_amd.models.sphere_conv import pack_dense_weight, pack_dense_weight_bf16x3
B, H, W, ci, co = [int(v) for v in sys.argv[1:6]]
print("plan", hip.sphere_conv_plan(B, H, W, ci, co), flush=True)
g = torch.Generator().manual_seed(7)
x = torch.randn(B * H * W, ci, generator=g).cuda()
w = (torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5).cuda()
b = torch.randn(co, generator=g).cuda()
c8 = -(-ci // 8) * 8
xs = torch.empty(B * H * W, c8, device="cuda")
hip.split_rows(x, xs, rows=B * H * W, C=ci)
torch.cuda.synchronize(); print("split ok", flush=True)
wp = pack_dense_weight_bf16x3(w)
torch.cuda.synchronize(); print("pack ok", wp.shape, wp.dtype, flush=True)
y32 = torch.empty(B * H * W, co, device="cuda")
hip.sphere_conv_nhwc(x, pack_dense_weight(w), y32, B=B, H=H, W=W, cin=ci, cout=co, bias=b, ksize=3)
torch.cuda.synchronize(); print("fp32 conv ok", flush=True)
y = torch.full((B * H * W, co), float("nan"), device="cuda")
hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=b, ksize=3)
torch.cuda.synchronize(); print("halo conv ok", flush=True)
d = (y - y32)
print("rel", (d.double().norm() / y32.double().norm()).item(), "nan", torch.isnan(y).sum().item())
bad = (d.abs() > 1e-3 * y32.abs().max()).nonzero()
print("bad elements", bad.shape[0])
if bad.shape[0]:
    rows = bad[:, 0].unique()
    pix = rows % (H * W)
    print("bad rows (first 40): (frame, y, x)", [(int(r // (H * W)), int(p // W), int(p % W)) for r, p in zip(rows[:40], pix[:40])])
    print("bad cols", bad[:, 1].unique()[:20].tolist(), "n rows", rows.numel())
