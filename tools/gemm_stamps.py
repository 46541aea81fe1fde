"""Timeline of the bf16x3 DMA GEMM from in-kernel wall-clock stamps (diagnostic build: `make -C ladcast_amd/csrc stamps`).
usage: LDC_LIB_PATH=ladcast_amd/libladcast_hip_stamps.so python tools/gemm_stamps.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ladcast_amd.hip as hip

M, N, K = (int(v) for v in sys.argv[1:4])
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
Wp = hip.pack_weight_bf16x2(W)
if os.environ.get("A_SPLIT", "1") != "0":  # as the model: pre-split activations -> the 16x16x32 kernel
    Ap = hip.pack_weight_bf16x2(A)
    _prob = [hip.gemm_problem(Ap, Wp, C, M=M, N=N, K=K, flags=hip.GEMM_A_SPLIT)]
else:
    _prob = [hip.gemm_problem(A, Wp, C, M=M, N=N, K=K)]
run = lambda: hip.gemm_grouped(_prob, split_bf16=True)
import time
t_end = time.time() + float(os.environ.get("WARM_S", "2"))
while time.time() < t_end:  # the chip settles its clock under sustained load
    for _ in range(20):
        run()
    torch.cuda.synchronize()
torch.cuda.synchronize()
ws = hip._grouped_workspace(A.device)
raw = ws.view(torch.int64)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
raw[65536:65536 + 256 * 16].zero_()
for _ in range(20):
    run()
s.record()
run()
e.record()
torch.cuda.synchronize()
st = raw[65536:65536 + 256 * 16].cpu().numpy().reshape(256, 16).astype(np.float64)
used = st[:, 0] > 0
st = st[used]
t0 = st[:, 0].min()
print(f"M={M} N={N} K={K}: event time {s.elapsed_time(e)*1e3:.1f} us, workgroups {used.sum()}")
names = {0: "entry", 1: "seg0 first DMA landed", 2: "seg0 loop done", 3: "seg0 published/ticket", 4: "seg0 done",
         5: "seg1 first DMA landed", 6: "seg1 loop done", 7: "seg1 published/ticket", 8: "seg1 done",
         9: "seg2+ first DMA", 10: "seg2+ loop done", 11: "seg2+ ticket", 12: "seg2+ done", 15: "exit"}
for i in sorted(names):
    col = st[:, i]
    ok = col > 0
    if not ok.any():
        continue
    v = (col[ok] - t0) / 100.0  # 100 MHz -> us
    print(f"  {names[i]:26s} n={ok.sum():3d}  min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us")
clk = (st[:, 14] - st[:, 13]) / np.maximum(st[:, 2] - st[:, 1], 1) * 100.0  # MHz: shader cycles per 100 MHz tick
print(f"  in-kernel clock over segment 0's loop: median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f}); "
      f"shader cycles in that loop: median {np.median(st[:, 14] - st[:, 13]):.0f}")
