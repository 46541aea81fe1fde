"""exact-fp32 attention: the balanced schedule (ldc_attn_fwd_ws with a workspace) against the one-unit-per-workgroup grids - values against
a float64 reference, repeatability, and time under sustained load.  usage: attn_f32_bal.py [B S H ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

v = [int(a) for a in sys.argv[1:]]
shapes = [tuple(v[i : i + 3]) for i in range(0, len(v), 3)] or [(1, 2250, 12), (1, 2250, 16), (2, 2250, 12), (4, 2250, 12), (1, 450, 12), (1, 1000, 3), (3, 333, 5)]
for B, S, H in shapes:
    D = H * 128
    g = torch.Generator().manual_seed(S + H)
    qkv = (torch.randn(B, S, 3 * D, generator=g) * 1.5).cuda()
    bias = (0.5 * torch.randn(S, generator=g)).cuda()
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    q, k, vv = (qkv[:, :, i * D : (i + 1) * D].reshape(B, S, H, 128).transpose(1, 2).double() for i in range(3))
    for kb in (None, bias):
        sc = q @ k.transpose(-1, -2) / 128 ** 0.5
        if kb is not None:
            sc = sc + kb.double()
        want = (torch.softmax(sc, -1) @ vv).transpose(1, 2).reshape(B, S, D)
        outs = {}
        for name, use in (("plain", False), ("balanced", True)):
            O = torch.full((B, S, D), float("nan"), device="cuda")
            hip.attn_fwd(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], O, key_bias=kb, use_workspace=use, **kw)
            torch.cuda.synchronize()
            outs[name] = O
            err = ((O.double() - want).norm() / want.norm()).item()
            same = True
            for _ in range(5):
                O2 = torch.empty_like(O)
                hip.attn_fwd(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], O2, key_bias=kb, use_workspace=use, **kw)
                same = same and torch.equal(O, O2)
            print(f"B={B} S={S} H={H} bias={kb is not None} {name:9s}: rel-L2 vs float64 {err:.2e}  finite {bool(torch.isfinite(O).all())}  repeatable {same}", flush=True)
        print(f"    balanced vs plain: rel-L2 {((outs['balanced'] - outs['plain']).double().norm() / outs['plain'].double().norm()).item():.2e}")
    for name, use in (("plain", False), ("balanced", True)):
        O = torch.empty(B, S, D, device="cuda")
        fn = lambda: hip.attn_fwd(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], O, use_workspace=use, **kw)
        t_end = time.time() + 1.0
        while time.time() < t_end:
            for _ in range(20): fn()
            torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 30 * 1e3)
        t = sorted(ts)[2]
        print(f"    {name:9s} {t:8.1f} us  {4 * B * H * S * S * 128 / t / 1e6:6.1f} TFLOP/s  (units {-(-S // 128) * H * B})", flush=True)
