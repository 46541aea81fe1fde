"""Random-shape check of the halo-staged conv (csrc/conv_halo.hip) against the exact-fp32 conv kernel of the same library: image sizes, frame
counts, channel counts (tails, non-multiples of 8), output widths (ragged panels), with residual / activation / operand-row output drawn at
random; only shapes whose plan IS the halo kernel are run (incl. the two-workgroups-per-tile form).  usage: conv_halo_fuzz.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip
from ladcast_amd.models.sphere_conv import pack_dense_weight, pack_dense_weight_bf16, pack_dense_weight_bf16x3

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, n, worst, ksplit_cases, bf16_cases = time.time() + seconds, 0, 0.0, 0, 0
while time.time() < t_end:
    H, W = rng.choice([8, 15, 16, 17, 24, 30, 33, 48, 60]), 2 * rng.choice([8, 9, 12, 15, 16, 20, 24, 30, 33, 60])
    ci, co = 4 * rng.randint(1, 40), 4 * rng.randint(1, 80)
    tiles1 = -(-H // 16) * -(-W // 16) * -(-co // 128)
    B = rng.choice([max(1, -(-200 // tiles1)), max(1, -(-110 // tiles1)), max(1, -(-300 // tiles1))])
    one = rng.random() < 0.25
    fmt = hip.FMT_BF16 if one else hip.FMT_SPLIT
    halo, bm, tw = hip.sphere_conv_plan(B, H, W, ci, co, in_fmt=fmt)
    if not halo or B * H * W * max(ci, co) > 6e7:
        continue
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    x = torch.randn(B * H * W, ci, generator=g).cuda()
    w = (torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5).cuda()
    b = torch.randn(co, generator=g).cuda() if rng.random() < 0.8 else None
    r = torch.randn(B * H * W, co, generator=g).cuda() if rng.random() < 0.5 else None
    act = rng.choice([hip.ACT_NONE, hip.ACT_RELU, hip.ACT_SILU])
    if one:  # the single-term kernel multiplies the bf16-rounded operands exactly: compare against fp32 on those
        x, w = x.bfloat16().float(), w.bfloat16().float()
    c8 = -(-ci // 8) * 8
    xs = torch.empty(B * H * W, c8, device="cuda")
    hip.split_rows(x, xs, rows=B * H * W, C=ci, fmt=fmt)
    y32 = torch.empty(B * H * W, co, device="cuda")
    hip.sphere_conv_nhwc(x, pack_dense_weight(w), y32, B=B, H=H, W=W, cin=ci, cout=co, bias=b, R=r, ldr=co if r is not None else 0, ksize=3, act=act)
    wp = pack_dense_weight_bf16(w) if one else pack_dense_weight_bf16x3(w)
    y = torch.full((B * H * W, co), float("nan"), device="cuda")
    hip.sphere_conv_nhwc_split(xs, wp, y, B=B, H=H, W=W, cin=ci, ldx=c8, cout=co, bias=b, R=r, ldr=co if r is not None else 0, ksize=3, act=act, in_fmt=fmt)
    torch.cuda.synchronize()
    err = ((y - y32).double().norm() / y32.double().norm()).item()
    tol = 3e-6 if one else 2e-5
    if not (err < tol):
        print(f"FAIL B={B} H={H} W={W} cin={ci} cout={co} bf16={one} act={act} bias={b is not None} resid={r is not None} plan={bm, tw}: rel-L2 {err:.3e}", flush=True)
        sys.exit(1)
    n += 1; worst = max(worst, err if not one else 0.0); bf16_cases += one
    tiles = B * -(-H // (256 // tw)) * -(-W // tw) * -(-co // 128)
    ksplit_cases += tiles < 192
print(f"{n} random shapes on the halo kernel ({ksplit_cases} with two workgroups per tile, {bf16_cases} in the single-term bf16 mode): all within tolerance; worst split-bf16 rel-L2 vs the exact-fp32 kernel {worst:.2e}")
