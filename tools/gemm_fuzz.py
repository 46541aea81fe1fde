"""Random-shape check of the grouped split-bf16 GEMM (gemm_bf16x3_v3.hip through ldc_gemm_grouped_bf16x3): 1 - 4 problems per launch with
random M (ragged row tiles), N (ragged column panels, N % 4 != 0 -> generic epilogue), K (whole k-steps), bias / gate / residual /
activation / batch / operand-row output drawn at random, against a float64 product of the same fp32 inputs; every launch is repeated
and must reproduce bit for bit (the stream-K pieces are reduced by the last arriver in a fixed order).  usage: gemm_fuzz.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ladcast_amd.hip as hip

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, n, worst = time.time() + seconds, 0, 0.0
acts = [hip.ACT_NONE, hip.ACT_SILU, hip.ACT_GELU_TANH] if hasattr(hip, "ACT_GELU_TANH") else [hip.ACT_NONE, hip.ACT_SILU]
def ref_act(y, a):
    if a == hip.ACT_SILU: return y * torch.sigmoid(y)
    if a != hip.ACT_NONE: return torch.nn.functional.gelu(y, approximate="tanh")
    return y
while time.time() < t_end:
    cnt = rng.choice([1, 1, 2, 2, 3, 4])
    same_k = rng.random() < 0.6
    K0 = 32 * rng.randint(1, 96)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    probs, keep, want = [], [], []
    work = 0
    for i in range(cnt):
        M = rng.choice([rng.randint(1, 300), rng.randint(300, 2600), 128 * rng.randint(1, 20)])
        N = rng.choice([4 * rng.randint(1, 400), 128 * rng.randint(1, 24), 84])
        K = K0 if same_k else 32 * rng.randint(1, 96)
        batch = rng.choice([1, 1, 1, 2, 3])
        if work + batch * M * N * K > 6e10:
            continue
        work += batch * M * N * K
        A = torch.randn(batch, M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        bias = torch.randn(N, generator=g).cuda() if rng.random() < 0.7 else None
        gate = torch.randn(batch, N, generator=g).cuda() if rng.random() < 0.3 else None
        R = torch.randn(batch, M, N, generator=g).cuda() if rng.random() < 0.4 else None
        act = rng.choice(acts) if gate is None and R is None else hip.ACT_NONE
        C = torch.full((batch, M, N), float("nan"), device="cuda")
        As = torch.empty(batch * M, K, device="cuda")
        hip.split_rows(A.view(batch * M, K), As, rows=batch * M, C=K)
        kw = dict(M=M, N=N, K=K, batch=batch, a_bs=M * K, c_bs=M * N, bias=bias, act=act, flags=hip.GEMM_A_SPLIT)
        if gate is not None: kw.update(gate=gate, gate_bs=N)
        if R is not None: kw.update(R=R, ldr=N, r_bs=M * N)
        probs.append(hip.gemm_problem(As, hip.pack_weight_bf16x2(W), C, **kw))
        y = A.double() @ W.double().t()
        if bias is not None: y = y + bias.double()
        y = ref_act(y, act)
        if gate is not None: y = y * gate.double()[:, None, :]
        if R is not None: y = y + R.double()
        want.append(y); keep.append((C, M, N, K, batch, act, bias is not None, gate is not None, R is not None))
    if not probs:
        continue
    hip.gemm_grouped(probs, split_bf16=True)
    torch.cuda.synchronize()
    first = [k[0].clone() for k in keep]
    for k in keep: k[0].fill_(float("nan"))
    hip.gemm_grouped(probs, split_bf16=True)
    torch.cuda.synchronize()
    for (C, M, N, K, batch, act, hb, hg, hr), y, f in zip(keep, want, first):
        err = ((C.double() - y).norm() / y.norm().clamp_min(1e-30)).item()
        if not (err < 2e-5) or not torch.equal(C, f):
            print(f"FAIL group of {cnt} (same K {same_k}): M={M} N={N} K={K} batch={batch} act={act} bias={hb} gate={hg} resid={hr}: rel-L2 {err:.3e} repeatable {torch.equal(C, f)}", flush=True)
            sys.exit(1)
        worst = max(worst, err)
    n += 1
print(f"{n} random grouped launches: all within 2e-5 of the float64 product and bitwise repeatable; worst rel-L2 {worst:.2e}")
