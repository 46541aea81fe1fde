"""a few launches of the pre-split-activation bf16x3 GEMM (16x16x32 kernel) for rocprofv3 --pmc runs: M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
M, N, K = [int(v) for v in sys.argv[1:4]]
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
p = [hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=hip.GEMM_A_SPLIT)]
for _ in range(6):
    hip.gemm_grouped(p, split_bf16=True)
torch.cuda.synchronize()
