"""In-kernel timeline of the 375M model's six GEMM launch types (grouped problems, QKV epilogue, gates / residuals as in the model).

Stamp build:   make -C ladcast_amd/csrc stamps && LDC_LIB_PATH=ladcast_amd/libladcast_hip_stamps.so python tools/gemm_launch_stamps.py
Product build: python tools/gemm_launch_stamps.py --time-only     (HIP-event time per launch type, sustained, no stamps)
Idle-gap experiment (is the power limit an energy budget over milliseconds or an instantaneous cap?):
               python tools/gemm_launch_stamps.py --time-only --gap-cycles 20000
               -> every GEMM launch is followed by a spin kernel of that many shader cycles (near-idle chip); the GEMM's own time is
                  measured with events around the GEMM only.  If the GEMM gets faster with gaps, the cap is an energy budget.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ladcast_amd.hip as hip

ap = argparse.ArgumentParser()
ap.add_argument("--time-only", action="store_true")
ap.add_argument("--gap-cycles", type=int, default=0)
ap.add_argument("--warm-s", type=float, default=1.5)
ap.add_argument("--launch", default="all")
ap.add_argument("--one-term", action="store_true")
args = ap.parse_args()

D, H, Nx, Nc = 1536, 12, 1800, 450
S = Nx + Nc
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
pack = hip.pack_weight_bf16 if args.one_term else hip.pack_weight_bf16x2
AS = hip.GEMM_A_SPLIT | (hip.GEMM_BF16_1TERM if args.one_term else 0)
CS = hip.GEMM_C_SPLIT


def act(rows, cols):  # activations in the operand format
    x = rnd(rows, cols)
    return pack(x).view(torch.float32).view(rows, -1) if not args.one_term else pack(x).view(rows, cols)


def as_f32_rows(t, rows, cols):
    """operand rows occupy fp32-sized rows in the model's buffers (plain bf16 rows: the first half of each row)"""
    if not args.one_term:
        return t
    buf = torch.zeros(rows, cols, device=dev, dtype=torch.float32)
    buf.view(torch.bfloat16).view(rows, 2 * cols)[:, :cols] = t
    return buf


F = 4 * D
nh = as_f32_rows(act(S, D), S, D)
att = as_f32_rows(act(S, D), S, D)
hid = as_f32_rows(act(S, F), S, F)
cat = as_f32_rows(act(S, D + F), S, D + F)
h = rnd(S, D)
qkv = torch.empty(S, 3 * D, device=dev)
mods = rnd(1, 6 * D)
rope = rnd(S, 128)
wn = rnd(128)


def W(n, k):
    return pack(rnd(n, k) * 0.02), rnd(n) * 0.02


def G(A, Wb, C, **kw):
    return hip.gemm_problem(A, Wb[0], C, bias=Wb[1], **kw)


w_qkv_x, w_qkv_c, w_out_x, w_out_c = W(3 * D, D), W(3 * D, D), W(D, D), W(D, D)
w_up_x, w_up_c, w_dn_x, w_dn_c = W(F, D), W(F, D), W(D, F), W(D, F)
w_mlp, w_sqkv, w_sout = W(F, D), W(3 * D, D), W(D, D + F)
epi = lambda r: hip.qkv_epilogue(wn, wn, r, eps=1e-7, heads=H)  # noqa: E731
cat_out = torch.empty(S, D + F, device=dev)
hid_out = torch.empty(S, F, device=dev)
cat_mlp = cat_out.view(torch.bfloat16)[:, D:] if args.one_term else cat_out[:, D:]

launches = {
    "dual_qkv": lambda: hip.gemm_grouped_qkv([G(nh, w_qkv_x, qkv, M=Nx, N=3 * D, K=D, flags=AS), G(nh[Nx:], w_qkv_c, qkv[Nx:], M=Nc, N=3 * D, K=D, flags=AS)],
                                            [epi(rope), epi(None)]),
    "dual_out": lambda: hip.gemm_grouped([G(att, w_out_x, h, M=Nx, N=D, K=D, gate=mods[:, :D], R=h, ldr=D, flags=AS),
                                          G(att[Nx:], w_out_c, h[Nx:], M=Nc, N=D, K=D, gate=mods[:, D:], R=h[Nx:], ldr=D, flags=AS)], split_bf16=True),
    "dual_up": lambda: hip.gemm_grouped([G(nh, w_up_x, hid_out, M=Nx, N=F, K=D, act=hip.ACT_GELU_TANH, flags=AS | CS),
                                         G(nh[Nx:], w_up_c, hid_out[Nx:], M=Nc, N=F, K=D, act=hip.ACT_GELU_TANH, flags=AS | CS)], split_bf16=True),
    "dual_down": lambda: hip.gemm_grouped([G(hid, w_dn_x, h, M=Nx, N=D, K=F, gate=mods[:, :D], R=h, ldr=D, flags=AS),
                                           G(hid[Nx:], w_dn_c, h[Nx:], M=Nc, N=D, K=F, gate=mods[:, D:], R=h[Nx:], ldr=D, flags=AS)], split_bf16=True),
    "single_qkv_mlp": lambda: hip.gemm_grouped_qkv([G(nh, w_mlp, cat_mlp, M=S, N=F, K=D, ldc=D + F, act=hip.ACT_GELU_TANH, flags=AS | CS),
                                                    G(nh, w_sqkv, qkv, M=S, N=3 * D, K=D, flags=AS)], [None, epi(rope)]),
    "single_out": lambda: hip.gemm_grouped([G(cat, w_sout, h, M=S, N=D, K=D + F, gate=mods[:, :D], R=h, ldr=D, flags=AS)], split_bf16=True),
}
flops = {"dual_qkv": 2 * S * 3 * D * D, "dual_out": 2 * S * D * D, "dual_up": 2 * S * F * D, "dual_down": 2 * S * F * D,
         "single_qkv_mlp": 2 * S * (F + 3 * D) * D, "single_out": 2 * S * D * (D + F)}
names = {0: "entry", 1: "seg0 first DMA landed", 2: "seg0 loop done", 3: "seg0 published/ticket", 4: "seg0 done",
         5: "seg1 first DMA landed", 6: "seg1 loop done", 7: "seg1 published/ticket", 8: "seg1 done",
         9: "seg2+ first DMA", 10: "seg2+ loop done", 11: "seg2+ ticket", 12: "seg2+ done", 15: "exit"}
if os.environ.get("LDC_STAMPS_PROLOGUE"):  # library built with DIAG=-DLDC_GEMM_STAMPS_PROLOGUE: start-up of the first segment (<= 2-segment launches)
    names.update({9: "seg0 decoded (scalars)", 10: "seg0 sources set", 11: "seg0 prologue DMAs issued"})

for name, run in launches.items():
    if args.launch not in ("all", name):
        continue
    t_end = time.time() + args.warm_s
    while time.time() < t_end:  # the chip settles its clock under sustained load
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    n = 50
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record()
        run()
        e.record()
        if args.gap_cycles:
            torch.cuda._sleep(args.gap_cycles)
    torch.cuda.synchronize()
    ts = np.array([s.elapsed_time(e) * 1e3 for s, e in ev])
    print(f"{name:15s} {np.median(ts):7.1f} us median ({ts.min():.1f} min) per launch = {flops[name] / np.median(ts) / 1e6:6.1f} TFLOP/s"
          + (f"   [each launch followed by a {args.gap_cycles}-cycle spin kernel]" if args.gap_cycles else ""), flush=True)
    if args.time_only:
        continue
    raw = hip._grouped_workspace(torch.device("cuda:0")).view(torch.int64)
    raw[65536:65536 + 256 * 16].zero_()
    for _ in range(20):
        run()
    run()
    torch.cuda.synchronize()
    st = raw[65536:65536 + 256 * 16].cpu().numpy().reshape(256, 16).astype(np.float64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    for i in sorted(names):
        col = st[:, i]
        ok = col > 0
        if not ok.any():
            continue
        v = (col[ok] - t0) / 100.0
        print(f"    {names[i]:26s} n={ok.sum():3d}  min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f} us")
    clk = (st[:, 14] - st[:, 13]) / np.maximum(st[:, 2] - st[:, 1], 1) * 100.0
    print(f"    in-kernel clock over segment 0's loop: median {np.median(clk):.0f} MHz; shader cycles in that loop: median {np.median(st[:, 14] - st[:, 13]):.0f}")
