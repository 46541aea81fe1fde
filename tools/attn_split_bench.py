"""the split attention kernel (operand rows from the QKV epilogue) at the model's shapes under sustained load, both arithmetic modes.
usage: python tools/attn_split_bench.py [B]   (the packed second-generation arm of the round-2 A/B was removed with its kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (S, H) in ((2250, 12), (450, 12), (2250, 16)):
    D = H * 128
    qkv = torch.randn(B, S, 3 * D, device="cuda")
    qkv[..., :D] *= 2.0
    out = torch.empty(B, S, D, device="cuda")
    q, k, v = qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :]
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)
    sp = qkv.clone()
    hip.attn_qkv_prepare_split(sp[:, :, :D], sp[:, :, D : 2 * D], sp[:, :, 2 * D :], split_row=S, **kw)
    fl = 4.0 * B * H * S * S * 128
    variants = {
        "split  x3": lambda: hip.attn_fwd_split(sp[:, :, :D], sp[:, :, D : 2 * D], sp[:, :, 2 * D :], out, ldo=D, o_bs=S * D, out_split=True, **kw),
        "x3 plain grid": lambda: hip.attn_fwd_split(sp[:, :, :D], sp[:, :, D : 2 * D], sp[:, :, 2 * D :], out, ldo=D, o_bs=S * D, out_split=True, use_workspace=False, **kw),
        "split  x1": lambda: hip.attn_fwd_split(sp[:, :, :D], sp[:, :, D : 2 * D], sp[:, :, 2 * D :], out, ldo=D, o_bs=S * D, out_split=True, one_term=True, **kw),
    }
    res = {n: [] for n in variants}
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for fn in variants.values():
            for _ in range(10): fn()
        torch.cuda.synchronize()
    for _ in range(7):
        for n, fn in variants.items():
            for _ in range(5): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): fn()
            b.record(); torch.cuda.synchronize()
            res[n].append(a.elapsed_time(b) * 20)
    print(f"B={B} S={S} H={H}: " + "  ".join(f"{n}: {sorted(v)[len(v)//2]:6.1f} us" + f" ({fl / sorted(v)[len(v)//2] / 1e6:5.0f} TF/s)" for n, v in res.items()))
