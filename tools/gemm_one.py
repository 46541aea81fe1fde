"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
M, N, K = [int(v) for v in sys.argv[1:4]]
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
Wp = hip.pack_weight_bf16x2(W)
for _ in range(6):
    if mode == "bf16x3": hip.gemm_sk(A, Wp, C, split_bf16=True, M=M, N=N, K=K)
    else: hip.gemm_sk(A, W, C, M=M, N=N, K=K)
torch.cuda.synchronize()
