"""Is every kernel bitwise reproducible when N processes share the GPU (wave preemption, kernels of several processes interleaved)?
Each of N concurrent processes runs each op REPS times on fixed seeded inputs and compares every result with its first; one line per
(process, op) with the number of differing repeats and the worst absolute difference.  usage: python tools/concurrency_race_probe.py [N] [REPS]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "worker" else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] != "worker" else 40

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rank, reps = int(sys.argv[2]), int(sys.argv[3])
    sys.path.insert(0, ROOT)
    import torch
    import ladcast_amd.hip as hip

    def rnd(*shape, seed=0, scale=1.0):
        return (scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))).cuda()

    def gemm_case(M, N_, K, batch=1, act=0, gate=False, qkv_heads=0, c_split=False, grouped=None):
        A = rnd(batch, M, K, seed=1)
        As = torch.empty_like(A)
        hip.split_rows(A.reshape(-1, K), As.reshape(-1, K), rows=batch * M, C=K, ldx=K, lds=K)
        probs = grouped or [(M, N_)]
        Ws = [hip.pack_weight_bf16x2(rnd(n_, K, seed=2 + i, scale=K ** -0.5)) for i, (_, n_) in enumerate(probs)]

        def fn():
            C = torch.zeros(batch, M, sum(n for _, n in probs), device="cuda")
            col, ps = 0, []
            for (m_, n_), W in zip(probs, Ws):
                ps.append(hip.gemm_problem(As, W, C[:, :, col:], M=m_, N=n_, K=K, batch=batch, a_bs=M * K, ldc=C.shape[2], c_bs=M * C.shape[2], act=act,
                                           flags=hip.GEMM_A_SPLIT | (hip.GEMM_C_SPLIT if c_split else 0)))
                col += n_
            hip.gemm_grouped(ps, split_bf16=True)
            return C
        return fn

    def attn_case(S, H, B=1):
        D = H * 128
        qkv = rnd(B, S, 3 * D, seed=5)
        kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)
        hip.attn_qkv_prepare_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], split_row=S, **kw)

        def fn():
            out = torch.zeros(B, S, D, device="cuda")
            hip.attn_fwd_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, ldo=D, o_bs=S * D, **kw)
            return out
        return fn

    def ln_case():
        x, sc, sh = rnd(1, 2250, 1536, seed=7), rnd(1, 1536, seed=8, scale=0.1), rnd(1, 1536, seed=9, scale=0.1)

        def fn():
            y = torch.zeros(1, 2250, 1536, device="cuda")
            hip.layernorm_mod(x, y, B=1, rows=2250, D=1536, ldx=1536, x_bs=2250 * 1536, ldy=1536, y_bs=2250 * 1536, scale=sc, shift=sh, mod_bs=1536, mode=0, eps=1e-6, out_split=hip.FMT_SPLIT)
            return y
        return fn

    ops = {
        "gemm dual up 2250x6144x1536 (256-row, gelu, C split)": gemm_case(2250, 6144, 1536, act=hip.ACT_GELU_TANH, c_split=True),
        "gemm dual down 2250x1536x6144 (128-row, split tiles)": gemm_case(2250, 1536, 6144),
        "gemm single out 2250x1536x7680": gemm_case(2250, 1536, 7680),
        "gemm single qkv+mlp 2250x(6144+4608)x1536 grouped": gemm_case(2250, 0, 1536, grouped=[(2250, 6144), (2250, 4608)]),
        "gemm head 1800x84x1536 (few tiles, split-K)": gemm_case(1800, 84, 1536),
        "attention split (1, 2250, 12)": attn_case(2250, 12),
        "attention split (2, 2250, 12): 4-wave form, two workgroups per CU": attn_case(2250, 12, B=2),
        "gemm single qkv+mlp batch 2 (256-row tiles)": gemm_case(2250, 0, 1536, batch=2, grouped=[(2250, 6144), (2250, 4608)]),
        "gemm dual down batch 2": gemm_case(2250, 1536, 6144, batch=2),
        "gemm single out batch 2": gemm_case(2250, 1536, 7680, batch=2),
        "attention split tail (1, 2250, 16)": attn_case(2250, 16),
        "layernorm_mod 2250x1536 -> split": ln_case(),
    }
    for name, fn in ops.items():
        ref = fn().clone()
        torch.cuda.synchronize()
        bad, worst = 0, 0.0
        for r0 in range(0, reps, 8):  # launched back to back, compared afterwards (a host sync after every launch hides timing effects)
            outs = [fn() for _ in range(min(8, reps - r0))]
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    worst = max(worst, (o - ref).abs().max().item())
        print(f"proc {rank}: {name:70s} {bad:3d} of {reps} repeats differ, worst abs diff {worst:.3e}", flush=True)
    sys.exit(0)

env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(REPS)], env=env, cwd=ROOT) for r in range(N)]
rc = [p.wait() for p in procs]
print("exit codes", rc)
