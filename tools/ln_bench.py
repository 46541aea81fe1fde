"""layernorm_mod timing at the model's shapes (sustained)"""
import os, sys, time
sys.path.insert(0, os.environ.get("LDC_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
D = 1536
for rows in (450, 1800, 2250, 18000):
    x = torch.randn(1, rows, D, device="cuda"); y = torch.empty_like(x); mod = torch.randn(1, 6 * D, device="cuda")
    fn = lambda: hip.layernorm_mod(x, y, B=1, rows=rows, D=D, ldx=D, x_bs=rows * D, ldy=D, y_bs=rows * D, scale=mod[:, D:], shift=mod, mod_bs=6 * D, mode=0, eps=1e-6, out_split=True)
    for _ in range(50): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(500): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 500
    print(f"rows={rows}: {us:.1f} us  {2 * rows * D * 4 / us / 1e6:.2f} TB/s")
