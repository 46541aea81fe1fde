"""Yardstick: what the VENDOR bf16 GEMM (torch.matmul -> hipBLASLt) sustains on the 375M model's launch shapes
with random operands on this box, next to this repo's split-bf16 (3 MFMAs per product) and single-term kernels on the same shapes.
Measurement aid only: nothing under ladcast_amd/ imports torch.matmul or this file.  One JSON object on stdout.

Reading: the split kernel can at best reach (vendor rate) / 3 in algorithmic TFLOP/s if the vendor kernel marks what this chip
sustains on these (small, 1-2 wave) grids under its power limit."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

SHAPES = {
    "refiner qkv": [(450, 4608, 1536)], "dual qkv": [(1800, 4608, 1536), (450, 4608, 1536)], "dual out": [(1800, 1536, 1536), (450, 1536, 1536)],
    "dual ff up": [(1800, 6144, 1536), (450, 6144, 1536)], "dual ff down": [(1800, 1536, 6144), (450, 1536, 6144)],
    "single qkv+mlp": [(2250, 6144, 1536), (2250, 4608, 1536)], "single out": [(2250, 1536, 7680)], "4096^3": [(4096, 4096, 4096)], "8192^3": [(8192, 8192, 8192)],
}
WARM_S = float(os.environ.get("WARM_S", "2.0"))


def timed(fn, flops):
    t_end = time.time() + WARM_S  # settle the clock under load first
    while time.time() < t_end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    return round(us, 1), round(flops / us / 1e6, 1)


res = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "rows": {}}
for name, probs in SHAPES.items():
    flops = sum(2.0 * M * N * K for M, N, K in probs)
    row = {}
    ops = [(torch.randn(M, K, device="cuda").bfloat16(), torch.randn(N, K, device="cuda").bfloat16()) for M, N, K in probs]
    row["vendor_bf16_us"], row["vendor_bf16_tflops"] = timed(lambda: [torch.matmul(a, w.t()) for a, w in ops], flops)
    zops = [(torch.zeros_like(a), torch.zeros_like(w)) for a, w in ops]
    row["vendor_bf16_zero_operands_us"], row["vendor_bf16_zero_operands_tflops"] = timed(lambda: [torch.matmul(a, w.t()) for a, w in zops], flops)
    # exact fp32 (round 6: the arithmetic of the headline): the vendor's fp32 GEMM (torch.matmul on fp32 tensors -> hipBLASLt / rocBLAS) next to this
    # repo's fp32-input MFMA ring kernel (gemm_bf16x3_v3_kernel<128, 0, false>) on the same shapes
    if max(M for M, _, _ in probs) <= 4096:
        fops = [(torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")) for M, N, K in probs]
        row["vendor_f32_us"], row["vendor_f32_tflops"] = timed(lambda: [torch.matmul(a, w.t()) for a, w in fops], flops)
        fps = [hip.gemm_problem(a, w, torch.empty(a.shape[0], w.shape[0], device="cuda"), M=a.shape[0], N=w.shape[0], K=a.shape[1]) for a, w in fops]
        row["ours_f32_us"], row["ours_f32_tflops"] = timed(lambda: hip.gemm_grouped(fps, split_bf16=False), flops)
        row["ours_f32_over_vendor_f32"] = round(row["ours_f32_tflops"] / row["vendor_f32_tflops"], 3)
        del fops, fps
    if max(M for M, _, _ in probs) <= 4096:
        for mode, fl in (("split3", hip.GEMM_A_SPLIT), ("single_term", hip.GEMM_A_SPLIT | hip.GEMM_BF16_1TERM)):
            ps = []
            for M, N, K in probs:
                A, W, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.empty(M, N, device="cuda")
                if mode == "single_term":  # plain bf16 operand rows (first 2 K bytes of each fp32-sized row)
                    Ab = torch.zeros_like(A)
                    Ab.view(torch.bfloat16)[:, :K] = A.bfloat16()
                    ps.append(hip.gemm_problem(Ab, hip.pack_weight_bf16(W), C, M=M, N=N, K=K, flags=fl))
                else:
                    ps.append(hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=fl))
            row[f"ours_{mode}_us"], row[f"ours_{mode}_tflops"] = timed(lambda: hip.gemm_grouped(ps, split_bf16=True), flops)
        row["ours_split3_over_vendor_third"] = round(row["ours_split3_tflops"] / (row["vendor_bf16_tflops"] / 3.0), 3)
    res["rows"][name] = row
print(json.dumps(res))
