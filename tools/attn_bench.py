"""bf16x3 attention under sustained load: raw-operand kernel vs (norm+RoPE kernels) and pack + LDS-DMA kernel.
usage: python tools/attn_bench.py [B S H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

B, S, H = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (1, 2250, 12)
D = H * 128
Nx = S * 4 // 5
qkv = torch.randn(B, S, 3 * D, device="cuda"); O = torch.empty(B, S, D, device="cuda")
w = torch.ones(128, device="cuda"); cs = torch.randn(S, 128, device="cuda"); sn = torch.randn(S, 128, device="cuda")
pk = torch.empty(hip.attn_packed_bytes(B, S, H) // 4, device="cuda")
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)

def old_norm():
    hip.qk_rmsnorm_rope(q, k, B=B, row0=0, rows=Nx, H=H, ld=3 * D, bs=S * 3 * D, wq=w, wk=w, eps=1e-6, cos=cs, sin=sn)
    hip.qk_rmsnorm_rope(q, k, B=B, row0=Nx, rows=S - Nx, H=H, ld=3 * D, bs=S * 3 * D, wq=w, wk=w, eps=1e-6, cos=cs, sin=sn)
def old_attn(): hip.attn_fwd(q, k, v, O, ldo=D, o_bs=S * D, split_bf16=True, **kw)
def pack(): hip.attn_pack(q, k, v, pk, split_row=Nx, seg0=(w, w, cs, sn), seg1=(w, w, cs, sn), **kw)
def new_attn(): hip.attn_fwd_packed(pk, O, B=B, S=S, H=H, ldo=D, o_bs=S * D)

def timed(fn, warm_s=1.0, iters=200):
    t_end = time.time() + warm_s
    while time.time() < t_end:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

fl = 4 * B * H * S * S * 128
for name, fn in (("qk_rmsnorm_rope x2", old_norm), ("attn raw operands", old_attn), ("pack (norm+rope+split)", pack), ("attn packed + DMA", new_attn)):
    us = timed(fn)
    extra = f"  {fl / us / 1e6:6.1f} TF/s algorithmic" if "attn" in name else ""
    print(f"B={B} S={S} H={H}  {name:24s} {us:8.1f} us{extra}")
