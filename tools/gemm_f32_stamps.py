"""Per-k-step cycle account of the exact-fp32 ring GEMM (gemm_bf16x3_v3_kernel<128, 0, false>) from in-kernel stamps.
Diagnostic builds:  make -C ladcast_amd/csrc stamps [DIAG=-DLDC_GEMM_DIAG_NODMA|NOLDS|NOBARRIER SUFFIX=_x]
usage: LDC_LIB_PATH=ladcast_amd/libladcast_hip_stamps.so python tools/gemm_f32_stamps.py [M N K]
With M x N = 256 whole 128 x 128 tiles (default 2048 x 2048) every CU runs exactly ONE tile, so stamps 13 / 14 (s_memtime around segment 0's
loop) give shader cycles per k-step directly: the MFMA floor is 64 MFMAs x 32 cycles x 2 waves per SIMD = 4096 cycles."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ladcast_amd.hip as hip

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 2048, 1536)
A = torch.randn(M, K, device="cuda")
W = torch.randn(N, K, device="cuda")
C = torch.empty(M, N, device="cuda")
prob = [hip.gemm_problem(A, W, C, M=M, N=N, K=K)]
run = lambda: hip.gemm_grouped(prob, split_bf16=False)  # noqa: E731
t_end = time.time() + float(os.environ.get("WARM_S", "2"))
while time.time() < t_end:
    for _ in range(10):
        run()
    torch.cuda.synchronize()
raw = hip._grouped_workspace(A.device).view(torch.int64)
raw[65536:65536 + 256 * 16].zero_()
for _ in range(10):
    run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
run()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3
st = raw[65536:65536 + 256 * 16].cpu().numpy().reshape(256, 16).astype(np.float64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
kt = K // 32
tiles = -(-M // 128) * -(-N // 128)
print(f"fp32 ring GEMM {M} x {N} x {K}: {tiles} tiles x {kt} k-steps on {len(st)} workgroups, event time {us:.1f} us = {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s "
      f"[{os.environ.get('LDC_LIB_PATH', 'shipped library')}]")
for i, nm in ((0, "entry"), (1, "seg0 first DMA landed"), (2, "seg0 loop done"), (4, "seg0 done (epilogue / hand-off)"), (15, "exit")):
    v = (st[:, i][st[:, i] > 0] - t0) / 100.0
    if len(v):
        print(f"  {nm:34s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us")
if os.environ.get("PER_XCD"):  # which workgroups finish early / late: exit time by XCD (blockIdx & 7) and by position inside the XCD's run of unit ranges
    full = raw[65536:65536 + 256 * 16].cpu().numpy().reshape(256, 16).astype(np.float64)
    ex = (full[:, 15] - t0) / 100.0
    for x in range(8):
        v = ex[x::8][full[x::8, 0] > 0]
        if len(v):
            print(f"  XCD {x}: exit min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us   first 4 ranges {np.round(v[:4], 1)}  last 4 {np.round(v[-4:], 1)}")
cyc = st[:, 14] - st[:, 13]
wall = (st[:, 2] - st[:, 1]) / 100.0
seg_k = kt if tiles * kt % len(st) == 0 and (tiles * kt // len(st)) % kt == 0 else None
clk = cyc / np.maximum(wall, 1e-9)
print(f"  segment 0's loop: median {np.median(cyc):.0f} shader cycles in {np.median(wall):.2f} us (clock {np.median(clk):.0f} MHz)"
      + (f" = {np.median(cyc) / seg_k:.0f} cycles per k-step (MFMA floor 4096 = {4096 * seg_k / np.median(cyc):.3f} of it)" if seg_k else ""))
