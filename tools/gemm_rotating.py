"""Sustained bf16x3 GEMM loop that cycles through enough distinct weight buffers to exceed the 256 MB Infinity Cache:
every weight byte comes from HBM, as in the model (1.5 GB of weights per forward), while the clock is settled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

for name, probs in {"single qkv+mlp": [(2250, 6144, 1536), (2250, 4608, 1536)], "single out": [(2250, 1536, 7680)],
                    "dual ff down": [(1800, 1536, 6144), (450, 1536, 6144)]}.items():
    wbytes = sum(N * K * 4 for (_, N, K) in probs)
    for copies in (1, max(2, int(700e6 // wbytes) + 1)):
        sets = []
        for _ in range(copies):
            ps = []
            for (M, N, K) in probs:
                A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
                ps.append(hip.gemm_problem(A, hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K))
                del W
            sets.append(ps)
        i = 0
        t_end = time.time() + 1.5
        while time.time() < t_end:
            for _ in range(50):
                hip.gemm_grouped(sets[i % copies], split_bf16=True); i += 1
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200):
            hip.gemm_grouped(sets[i % copies], split_bf16=True); i += 1
        e.record(); torch.cuda.synchronize()
        print(f"{name:16s} weight sets {copies:3d} ({copies * wbytes / 1e6:6.0f} MB): {s.elapsed_time(e) * 1e3 / 200:7.1f} us")
        del sets
