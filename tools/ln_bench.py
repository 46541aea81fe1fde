"""layernorm_mod timing at the model's shapes: 50 launches captured in one hipGraph (eager launching of a 9 us kernel from Python
measures the host)"""
import os, sys
sys.path.insert(0, os.environ.get("LDC_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # LDC_PKG_ROOT: A/B against a snapshot
import torch
import ladcast_amd.hip as hip
D = 1536
for rows in (450, 2250, 18000):
    x = torch.randn(1, rows, D, device="cuda"); y = torch.empty_like(x); mod = torch.randn(1, 6 * D, device="cuda")
    fn = lambda: hip.layernorm_mod(x, y, B=1, rows=rows, D=D, ldx=D, x_bs=rows * D, ldy=D, y_bs=rows * D, scale=mod[:, D:], shift=mod, mod_bs=6 * D, mode=0, eps=1e-6, out_split=True)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(50): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): g.replay()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 1000
    print(f"rows={rows}: {us:.2f} us  {2 * rows * D * 4 / us / 1e6:.2f} TB/s")
