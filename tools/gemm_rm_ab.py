"""A/B of the GEMM tile order (LDC_BF16X3_RM: super-row height; 0 = the one-super-row order of round 1, unset = launch_v3's choice),
interleaved rounds in ONE process under sustained load (cdna_hip_programming.md rule 24).  Prints per launch shape the median us per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

calls = {
    "refiner qkv": [(450, 4608, 1536)], "dual qkv": [(1800, 4608, 1536), (450, 4608, 1536)], "dual out": [(1800, 1536, 1536), (450, 1536, 1536)],
    "dual ff up": [(1800, 6144, 1536), (450, 6144, 1536)], "dual ff down": [(1800, 1536, 6144), (450, 1536, 6144)],
    "single qkv+mlp": [(2250, 6144, 1536), (2250, 4608, 1536)], "single out": [(2250, 1536, 7680)],
    "B8 single out": [(18000, 1536, 7680)],
}
settings = sys.argv[1:] or ["0", "auto", "3", "5", "9"]
rounds = int(os.environ.get("ROUNDS", "7"))
for name, probs in calls.items():
    ps, flops = [], 0
    for M, N, K in probs:
        A, W, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
        ps.append(hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=hip.GEMM_A_SPLIT))
        flops += 2 * M * N * K
    fn = lambda: hip.gemm_grouped(ps, split_bf16=True)
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    res = {s: [] for s in settings}
    for _ in range(rounds):
        for s in settings:
            if s == "auto":
                os.environ.pop("LDC_BF16X3_RM", None)
            else:
                os.environ["LDC_BF16X3_RM"] = s
            for _ in range(10): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100): fn()
            b.record(); torch.cuda.synchronize()
            res[s].append(a.elapsed_time(b) * 10)
    os.environ.pop("LDC_BF16X3_RM", None)
    line = "  ".join(f"rm={s}: {sorted(v)[len(v)//2]:7.1f} us (min {min(v):6.1f})" for s, v in res.items())
    print(f"{name:16s} {line}")
