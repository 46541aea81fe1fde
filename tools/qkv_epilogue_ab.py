"""What the fused QKV epilogue (bias -> RMSNorm -> RoPE -> scale -> split rows) costs on top of the plain GEMM epilogue, per launch shape of the
375M model, interleaved rounds in one process under sustained load.  The pack pass it replaces took 16-22 us per attention call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

H = 12
D = H * 128
calls = {"refiner qkv": [(450, 3 * D, D, "q")], "dual qkv": [(1800, 3 * D, D, "q"), (450, 3 * D, D, "q")],
         "single qkv+mlp": [(2250, 6144, D, "m"), (2250, 3 * D, D, "q")]}
wq, wk = torch.rand(128, device="cuda") + 0.5, torch.rand(128, device="cuda") + 0.5
rope = torch.rand(2250, 128, device="cuda")  # compact (cos_i, sin_i) table
for name, probs in calls.items():
    ps, epis = [], []
    for M, N, K, kind in probs:
        A, W, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
        b = torch.randn(N, device="cuda")
        fl = hip.GEMM_A_SPLIT | (hip.GEMM_C_SPLIT if kind == "m" else 0)
        ps.append(hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, bias=b, act=2 if kind == "m" else 0, flags=fl))
        epis.append(hip.qkv_epilogue(wq, wk, rope, heads=H) if kind == "q" else None)
    epis_nr = [hip.qkv_epilogue(wq, wk, None, heads=H) if e is not None else None for e in epis]  # norm + split only (no rotary table traffic)
    epis_nn = [hip.qkv_epilogue(None, None, None, heads=H) if e is not None else None for e in epis]  # split only
    variants = {"plain": lambda: hip.gemm_grouped(ps, split_bf16=True), "qkv epilogue": lambda: hip.gemm_grouped_qkv(ps, epis),
                "no rope": lambda: hip.gemm_grouped_qkv(ps, epis_nr), "split only": lambda: hip.gemm_grouped_qkv(ps, epis_nn)}
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for fn in variants.values():
            for _ in range(20): fn()
        torch.cuda.synchronize()
    res = {n: [] for n in variants}
    for _ in range(9):
        for n, fn in variants.items():
            for _ in range(5): fn()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100): fn()
            b_.record(); torch.cuda.synchronize()
            res[n].append(a.elapsed_time(b_) * 10)
    print(f"{name:16s} " + "  ".join(f"{n}: {sorted(v)[len(v)//2]:7.1f} us" for n, v in res.items()))
