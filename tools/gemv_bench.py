"""the AdaLN modulation GEMV (38 D x D fp32 weights, one row per member) under repeated launches; LDC_LINEAR_SMALL_ITERS forces
the column groups per workgroup (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = torch.randn(rows, D, device="cuda"); W = torch.randn(38 * D, D, device="cuda"); b = torch.randn(38 * D, device="cuda"); y = torch.empty(rows, 38 * D, device="cuda")
fn = lambda: hip.linear_small(x, W, y, rows=rows, N=38 * D, K=D, bias=b, act_in=hip.ACT_SILU)
for _ in range(20): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): fn()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 200
print(f"D={D} rows={rows} iters={os.environ.get('LDC_LINEAR_SMALL_ITERS', 'auto')}: {us:.1f} us, {W.numel() * 4 / us / 1e6:.2f} TB/s")
