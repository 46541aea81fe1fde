"""Per-shape timing of the GEMM / attention kernels through the C ABI (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

def time_it(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = [(4096,4096,4096),(2250,1536,1536),(2250,4608,1536),(2250,6144,1536),(2250,1536,6144),(2250,1536,7680),(1800,1536,1536),(450,1536,1536),(450,4608,1536),(450,6144,1536),(450,1536,6144),(1800,6144,1536),(1800,1536,6144),(4500,4608,1536),(4500,1536,7680), (18000,1536,1536),(18000,6144,1536)]
for (M,N,K) in shapes:
    A = torch.randn(M,K,device="cuda"); W = torch.randn(N,K,device="cuda"); C = torch.empty(M,N,device="cuda")
    ms = time_it(lambda: hip.gemm(A,W,C,M=M,N=N,K=K))
    ms2 = time_it(lambda: hip.gemm_sk(A,W,C,M=M,N=N,K=K))
    Wp = hip.pack_weight_bf16x2(W)
    ms3 = time_it(lambda: hip.gemm_sk(A,Wp,C,split_bf16=True,M=M,N=N,K=K))
    print(f"gemm M={M:6d} N={N:5d} K={K:5d}: tile-per-WG {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:6.1f} TF/s | stream-K fp32 {ms2*1e3:8.1f} us {2*M*N*K/ms2/1e9:6.1f} TF/s | stream-K bf16x3 {ms3*1e3:8.1f} us {2*M*N*K/ms3/1e9:6.1f} TF/s  tiles={((M+127)//128)*((N+127)//128)}")
for (B,S,H) in [(1,2250,12),(1,450,12),(2,2250,12),(8,2250,12),(1,2250,16)]:
    D=H*128
    qkv = torch.randn(B,S,3*D,device="cuda"); O = torch.empty(B,S,D,device="cuda")
    ms = time_it(lambda: hip.attn_fwd(qkv[:,:,:D],qkv[:,:,D:2*D],qkv[:,:,2*D:],O,B=B,S=S,H=H,ld_qkv=3*D,qkv_bs=S*3*D,ldo=D,o_bs=S*D))
    ms2 = float("nan")  # the first-generation split attention was removed in round 2 (tools/attn_split_bench.py times the current kernels)
    print(f"attn B={B} S={S} H={H}: fp32 {ms*1e3:9.1f} us {4*B*H*S*S*128/ms/1e9:7.1f} TF/s | bf16x3 {ms2*1e3:9.1f} us {4*B*H*S*S*128/ms2/1e9:7.1f} TF/s")
