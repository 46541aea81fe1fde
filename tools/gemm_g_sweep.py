"""one pre-split bf16x3 GEMM shape under sustained load; run once per setting of the measurement aids LDC_BF16X3_G (number of unit
ranges) / LDC_BF16X3_BM (tile height), which the library reads once per process.  usage: python tools/gemm_g_sweep.py M N K [M N K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

v = [int(a) for a in sys.argv[1:]]
probs = [tuple(v[i : i + 3]) for i in range(0, len(v), 3)]
ps, flops = [], 0
for M, N, K in probs:
    A, W, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
    ps.append(hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=hip.GEMM_A_SPLIT))
    flops += 2 * M * N * K
fn = lambda: hip.gemm_grouped(ps, split_bf16=True)
t_end = time.time() + 1.5
while time.time() < t_end:
    for _ in range(20): fn()
    torch.cuda.synchronize()
ts = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 20)
t = sorted(ts)[len(ts) // 2]
print(f"G={os.environ.get('LDC_BF16X3_G', 'auto'):>4s} BM={os.environ.get('LDC_BF16X3_BM', 'auto'):>4s} {probs}: {t:7.1f} us  {flops / t / 1e6:6.1f} TF/s")
