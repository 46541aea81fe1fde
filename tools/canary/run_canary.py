"""LDS canary experiment: one process runs tools/canary/lds_canary while N - 1 others loop ONE suspect kernel of the library.
usage: python tools/canary/run_canary.py <op> [N] [seconds] [canary KiB]      op: attn_b2 | attn_b1 | gemm128 | gemm256 | ln | gemv | forward_b2"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "worker":
    op, seconds = sys.argv[2], float(sys.argv[3])
    sys.path.insert(0, ROOT)
    import torch
    import ladcast_amd.hip as hip
    g = torch.Generator().manual_seed(1)
    if op in ("attn_f32_b2",):
        B, S, H = 2, 2250, 12
        D = H * 128
        qkv = torch.randn(B, S, 3 * D, generator=g).cuda()
        out = torch.empty(B, S, D, device="cuda")
        fn = lambda: hip.attn_fwd(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D, ldo=D, o_bs=S * D)
    elif op in ("attn_b2", "attn_b1", "attn_b2_1term"):
        B, S, H = (1 if op == "attn_b1" else 2), 2250, 12
        D = H * 128
        qkv = torch.randn(B, S, 3 * D, generator=g).cuda()
        kw = dict(B=B, S=S, H=H, ld_qkv=3 * D, qkv_bs=S * 3 * D)
        hip.attn_qkv_prepare_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], split_row=S, **kw)
        out = torch.empty(B, S, D, device="cuda")
        fn = lambda: hip.attn_fwd_split(qkv[:, :, :D], qkv[:, :, D : 2 * D], qkv[:, :, 2 * D :], out, ldo=D, o_bs=S * D, one_term=op.endswith("1term"), **kw)
    elif op in ("gemm128", "gemm256"):
        M, N, K = 2250, (1536 if op == "gemm128" else 6144), (6144 if op == "gemm128" else 1536)
        A = torch.randn(1, M, K, generator=g).cuda()
        As = torch.empty_like(A)
        hip.split_rows(A.reshape(-1, K), As.reshape(-1, K), rows=M, C=K, ldx=K, lds=K)
        W = hip.pack_weight_bf16x2((torch.randn(N, K, generator=g) / K**0.5).cuda())
        C = torch.empty(1, M, N, device="cuda")
        fn = lambda: hip.gemm_grouped([hip.gemm_problem(As, W, C, M=M, N=N, K=K, a_bs=M * K, c_bs=M * N, flags=hip.GEMM_A_SPLIT)], split_bf16=True)
    elif op == "ln":
        x, sc = torch.randn(2, 2250, 1536, generator=g).cuda(), (0.1 * torch.randn(2, 3072, generator=g)).cuda()
        y = torch.empty_like(x)
        fn = lambda: hip.layernorm_mod(x, y, B=2, rows=2250, D=1536, ldx=1536, x_bs=2250 * 1536, ldy=1536, y_bs=2250 * 1536, scale=sc[:, 1536:], shift=sc, mod_bs=3072, mode=0, eps=1e-6, out_split=hip.FMT_SPLIT)
    elif op == "gemv":
        x, W, b = torch.randn(2, 1536, generator=g).cuda(), (torch.randn(58368, 1536, generator=g) / 39).cuda(), torch.randn(58368, generator=g).cuda()
        y = torch.empty(2, 58368, device="cuda")
        fn = lambda: hip.linear_small(x, W, y, rows=2, N=58368, K=1536, bias=b, act_in=hip.ACT_SILU)
    else:
        import bench
        from ladcast_amd.models import LaDCastTransformer3DModel
        torch.manual_seed(1234)
        model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to("cuda").eval().set_gemm_precision("bf16x3")
        x = torch.randn(2, 84, 4, 15, 30, generator=g).cuda()
        known = torch.randn(2, 84, 1, 15, 30, generator=g).cuda()
        ts, t = torch.tensor([2018010100]).cuda(), torch.tensor([0.3]).cuda()
        fn = lambda: model(x, t, known, time_elapsed=ts)
    t_end = time.time() + seconds
    n = 0
    while time.time() < t_end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    print(f"worker [{op}]: {n} calls" + (f", qkv at 0x{qkv.data_ptr():x} .. 0x{qkv.data_ptr() + qkv.numel() * 4:x}" if "attn" in op else ""), flush=True)
    sys.exit(0)
op = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
kib = sys.argv[4] if len(sys.argv) > 4 else "16"
spin = sys.argv[5] if len(sys.argv) > 5 else "200"
cmode = sys.argv[6] if len(sys.argv) > 6 else "0"
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
workers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", op, str(seconds + 6)], env=env, cwd=ROOT, stdout=subprocess.DEVNULL) for _ in range(N - 1)]
time.sleep(5)  # let the workers build their inputs
print(f"== canary next to {N - 1} x [{op}]", flush=True)
if cmode.startswith("stream"):  # tools/canary/stream_canary.hip: the GEMV's streaming loads, checked word by word ("stream" | "stream_plain")
    subprocess.run(["/tmp/stream_canary", str(seconds), "0" if cmode.endswith("plain") else "1"])
else:
    subprocess.run(["/tmp/lds_canary", str(seconds), kib, spin, cmode])
for w in workers:
    w.wait()
