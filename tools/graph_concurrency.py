"""Do two independent branches of a captured hipGraph run concurrently on this runtime?  (development aid)
branch 1: the AdaLN modulation GEMV (HBM-bound, ~85 us); branch 2: the x patch-embed-sized GEMM (MFMA-bound, ~45 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

D = 1536
dev = "cuda"
x = torch.randn(1, D, device=dev); Wm = torch.randn(38 * D, D, device=dev); ym = torch.empty(1, 38 * D, device=dev)
M, N, K = 1800, 1536, 1536
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
prob = [hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=hip.GEMM_A_SPLIT)]
REP = 20


def gemv():
    hip.linear_small(x, Wm, ym, rows=1, N=38 * D, K=D, act_in=hip.ACT_SILU)


def gemm():
    hip.gemm_grouped(prob, split_bf16=True)


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn(s)
    return g


side = torch.cuda.Stream()


def seq(s):
    for _ in range(REP):
        gemv(); gemm()


def only_gemv(s):
    for _ in range(REP):
        gemv()


def only_gemm(s):
    for _ in range(REP):
        gemm()


def forked(s):
    for _ in range(REP):
        side.wait_stream(s)
        with torch.cuda.stream(side):
            gemm()
        gemv()
        s.wait_stream(side)


for name, fn in (("gemv only", only_gemv), ("gemm only", only_gemm), ("sequential", seq), ("forked", forked)):
    g = capture(fn)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"{name:12s} {a.elapsed_time(b) * 1e3 / 10 / REP:8.1f} us per iteration")
