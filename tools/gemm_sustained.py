"""bf16x3 GEMM per-shape time under SUSTAINED load (the chip lowers its clock after ~1 s of MFMA-dense work; a
20-iteration burst after idle reads 15-25 % optimistic -- the model runs in the settled state).
usage: python tools/gemm_sustained.py [--grouped] ; LDC_BF16X3_BM=128/256 forces a tile height"""
import os, sys, time
sys.path.insert(0, os.environ.get("LDC_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # LDC_PKG_ROOT: A/B against a snapshot
import torch
import ladcast_amd.hip as hip

# the 375M model's launches at one member: (list of (M, N, K)) per grouped call
calls = {
    "refiner qkv": [(450, 4608, 1536)], "refiner ff up": [(450, 6144, 1536, 1)], "refiner ff down": [(450, 1536, 6144)],
    "refiner proj_in": [(450, 1536, 1536)], "patch embeds": [(1800, 1536, 96), (450, 1536, 96)],
    "dual qkv": [(1800, 4608, 1536), (450, 4608, 1536)], "dual out": [(1800, 1536, 1536), (450, 1536, 1536)],
    "dual ff up": [(1800, 6144, 1536, 2), (450, 6144, 1536, 2)], "dual ff down": [(1800, 1536, 6144), (450, 1536, 6144)],
    "single qkv+mlp": [(2250, 6144, 1536, 2), (2250, 4608, 1536)], "single out": [(2250, 1536, 7680)],
    "4096^3": [(4096, 4096, 4096)],
}
warm_s = float(os.environ.get("WARM_S", "1.5"))
a_split = os.environ.get("A_SPLIT", "1") != "0"
only = os.environ.get("ONLY")  # comma-separated call names
for name, probs in calls.items():
    if only and name not in only.split(","):
        continue
    ps = []
    flops = 0
    for prob in probs:  # (M, N, K[, act]): act 1 = SiLU, 2 = GELU-tanh, with bias, as the model's MLP-up epilogues
        M, N, K = prob[:3]
        act = prob[3] if len(prob) > 3 else 0
        ekw = dict(act=act, bias=torch.randn(N, device="cuda")) if act else {}
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
        if a_split:  # as the model: activations pre-split by their producer -> the 16x16x32 kernel
            ps.append(hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=hip.GEMM_A_SPLIT | (hip.GEMM_C_SPLIT if act else 0), **ekw))
        else:
            ps.append(hip.gemm_problem(A, hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, **ekw))
        flops += 2 * M * N * K
    fn = lambda: hip.gemm_grouped(ps, split_bf16=True)
    t_end = time.time() + warm_s
    while time.time() < t_end:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 200
    print(f"{name:16s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s")
