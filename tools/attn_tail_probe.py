"""attention kernel alone at unit counts around the 256 CUs (what one round costs when every CU is busy; the TAIL schedule vs the plain grids)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

def run(S, H, B=1, ws=True, n=200):
    D = H * 128
    qkv = torch.randn(B, S, 3 * D, device="cuda")
    qkv[..., :D] *= 2.0
    out = torch.empty(B, S, D, device="cuda")
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)
    hip.attn_qkv_prepare_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], split_row=S, **kw)
    fn = lambda: hip.attn_fwd_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, ldo=D, o_bs=S * D, out_split=True, use_workspace=ws, **kw)
    t_end = time.time() + 1.0
    while time.time() < t_end:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / n)
    us = sorted(ts)[2]
    nq, nt = -(-S // 128), -(-S // 32)
    units = nq * H * B
    print(f"S={S} H={H} B={B} units={units} key tiles={nt} workspace={ws}: {us:7.1f} us  {4.0 * B * H * S * S * 128 / us / 1e6:5.0f} TF/s   us per (unit-round x key tile) = {us / nt / max(1.0, units / 256):.3f}")

for S, H in ((2250, 12), (2250, 14), (2048, 16), (2272, 16)):
    run(S, H)
run(2250, 16, ws=True)
run(2250, 16, ws=False)
run(2250, 12, B=2)
run(2250, 16, B=2, ws=True)
run(2250, 16, B=2, ws=False)
