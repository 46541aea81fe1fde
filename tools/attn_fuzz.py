"""Random-shape check of both attention kernels (exact fp32: ldc_attn_fwd_ws; split bf16: ldc_attn_fwd_split, operand rows made by
ldc_attn_qkv_prepare_split) against float64 softmax(q k^T / sqrt(128) + bias) v: batch, sequence length (ragged query blocks / key
tiles), heads (so that every schedule comes up: one unit per workgroup, two per CU, balanced ranges, persistent + key-sliced tail), key
bias on / off; every call twice, bit for bit.  usage: attn_fuzz.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, n, worst32, worst3 = time.time() + seconds, 0, 0.0, 0.0
sched = {"f32 balanced": 0, "f32 plain": 0, "split tail": 0, "split plain": 0}
while time.time() < t_end:
    B = rng.choice([1, 1, 2, 3, 5])
    S = rng.choice([rng.randint(1, 200), rng.randint(200, 1200), rng.randint(1200, 2600), 2250, 450])
    H = rng.choice([1, 2, 3, 5, 8, 12, 16, 20])
    if B * H * S * S > 4e8:
        continue
    D = H * 128
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    qkv = torch.randn(B, S, 3 * D, generator=g)
    qkv[..., :D] *= rng.choice([1.0, 2.0, 4.0])
    bias = 0.5 * torch.randn(S, generator=g) if rng.random() < 0.4 else None
    d = qkv.cuda()
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].reshape(B, S, H, 128).transpose(1, 2).double().cuda() for i in range(3))
    sc = q @ k.transpose(-1, -2) / 128 ** 0.5
    if bias is not None:
        sc = sc + bias.double().cuda()
    want = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, S, D)
    kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    units, nt = -(-S // 128) * H * B, -(-S // 32)
    # exact fp32
    outs = []
    for _ in range(2):
        o = torch.full((B, S, D), float("nan"), device="cuda")
        hip.attn_fwd(d[:, :, :D], d[:, :, D:2 * D], d[:, :, 2 * D:], o, key_bias=None if bias is None else bias.cuda(), **kw)
        outs.append(o)
    torch.cuda.synchronize()
    e32 = ((outs[0].double() - want).norm() / want.norm()).item()
    ok = e32 < 3e-6 and torch.equal(outs[0], outs[1])
    sched["f32 balanced" if (units % 256 and units <= 1024 and units * nt >= 512) else "f32 plain"] += 1
    # split bf16 (three-term)
    sp = d.clone()
    hip.attn_qkv_prepare_split(sp[:, :, :D], sp[:, :, D:2 * D], sp[:, :, 2 * D:], split_row=S, **{k_: kw[k_] for k_ in ("B", "S", "H", "ld_qkv", "qkv_bs")})
    kb = None if bias is None else hip.pad_key_bias(bias.cuda())
    outs3 = []
    for _ in range(2):
        o = torch.full((B, S, D), float("nan"), device="cuda")
        hip.attn_fwd_split(sp[:, :, :D], sp[:, :, D:2 * D], sp[:, :, 2 * D:], o, key_bias=kb, **kw)
        outs3.append(o)
    torch.cuda.synchronize()
    e3 = ((outs3[0].double() - want).norm() / want.norm()).item()
    ok = ok and e3 < 2e-5 and torch.equal(outs3[0], outs3[1])
    sched["split tail" if hip.lib.ldc_attn_fwd_split_workspace_bytes(B, S, H) > 0 else "split plain"] += 1
    if not ok:
        print(f"FAIL B={B} S={S} H={H} bias={bias is not None}: fp32 rel-L2 {e32:.2e} (repeat {torch.equal(outs[0], outs[1])}), split {e3:.2e} (repeat {torch.equal(outs3[0], outs3[1])})", flush=True)
        sys.exit(1)
    n += 1; worst32 = max(worst32, e32); worst3 = max(worst3, e3)
print(f"{n} random shapes x 2 kernels: all within tolerance and bitwise repeatable; worst rel-L2 fp32 {worst32:.2e}, split bf16 {worst3:.2e}; schedules taken: {sched}")
