"""bf16x3 GEMM timed with operands cold (HBM) vs warm (left in L2 / Infinity Cache by the previous call).
In the model every weight is read once per forward (1.5 GB per forward: always cold), so the cold figure is the one
that predicts in-model time.  usage: python tools/gemm_cold.py [M N K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

shapes = [(2250, 1536, 1536), (2250, 4608, 1536), (2250, 10752, 1536), (2250, 1536, 6144), (2250, 1536, 7680), (450, 1536, 1536)]
if len(sys.argv) > 3:
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
flush = torch.empty(768 * 1024 * 1024 // 4, device="cuda")
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    Wp = hip.pack_weight_bf16x2(W)
    fn = lambda: hip.gemm_sk(A, Wp, C, split_bf16=True, M=M, N=N, K=K)
    for _ in range(3): fn()
    res = {}
    for mode in ("warm", "cold_w", "cold_all"):
        ts = []
        for _ in range(12):
            if mode != "warm":
                flush.add_(1.0)          # 768 MB read + write: evicts L2 and the 256 MB Infinity Cache
                if mode == "cold_w":
                    A.add_(0.0)          # activations were just written by the previous kernel in the model
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print(f"M={M} N={N} K={K}: warm {res['warm']:.1f} us | weights cold {res['cold_w']:.1f} us | all cold {res['cold_all']:.1f} us")
