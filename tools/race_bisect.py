"""Which launch of a 375M forward stops being bitwise reproducible when N processes share the GPU?  Every C-ABI wrapper of ladcast_amd.hip is
wrapped: after each call the device tensors it was given (outputs among them) are checksummed (int64 sum of the bit patterns); the sequence
of (call, checksums) of repeat k is compared with repeat 0 of the same process and the first difference is printed.
usage: python tools/race_bisect.py [N] [REPS] [mode]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rank, reps, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sys.path.insert(0, ROOT)
    import types
    import torch
    import bench
    import ladcast_amd.hip as hip
    from ladcast_amd.models import LaDCastTransformer3DModel

    log = []

    def tensors_of(x, out):
        if isinstance(x, torch.Tensor):
            if x.is_cuda:
                out.append(x)
        elif isinstance(x, (list, tuple)):
            for y in x:
                tensors_of(y, out)

    def wrap(name, fn):
        def inner(*a, **k):
            r = fn(*a, **k)
            ts = []
            tensors_of(a, ts)
            tensors_of(list(k.values()), ts)
            sums = []
            for t in ts:
                if t.dtype in (torch.float32, torch.int32):
                    base = t if t.is_contiguous() else t.contiguous()
                    sums.append(base.view(torch.int32).sum(dtype=torch.int64))  # stays on the device: no host sync between launches
            keep = None
            if os.environ.get("RACE_KEEP") and name in ("linear_small", "gemm_grouped"):
                keep = [t.clone() for t in ts if t.dtype == torch.float32 and t.numel() <= 40_000_000]
            log.append((name, sums, keep))
            return r
        return inner

    skip = {"upload_nonblocking", "gemm_problem", "linear_small_problem", "qkv_epilogue", "compact_rope_table", "pad_key_bias", "conv_cin_padded"}
    for name in dir(hip):
        fn = getattr(hip, name)
        if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in skip and fn.__module__ == hip.__name__:
            setattr(hip, name, wrap(name, fn))
    torch.manual_seed(1234)
    model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to("cuda").eval().set_gemm_precision(mode)
    B = int(os.environ.get("RACE_B", "2"))
    x = torch.randn(B, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    known = (0.5 * torch.randn(B, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
    ts = torch.tensor([2018010100]).cuda()
    t = torch.tensor([0.3]).cuda()
    model(x, t, known, time_elapsed=ts)  # plan, workspaces
    ref = None
    for rep in range(reps):
        log.clear()
        out = model(x, t, known, time_elapsed=ts).sample
        torch.cuda.synchronize()
        cur = [(n, tuple(int(v) for v in (torch.stack(sm).cpu().tolist() if sm else []))) for n, sm, _ in log]
        keeps = [k for _, _, k in log]
        if ref is None:
            ref, ref_keeps = cur, keeps
            continue
        if cur != ref:
            for i, (a, b) in enumerate(zip(ref, cur)):
                if a != b:
                    which = [j for j, (u, v) in enumerate(zip(a[1], b[1])) if u != v]
                    print(f"proc {rank} rep {rep}: first difference at launch {i} of {len(ref)}: {a[0]} (tensor args {which} of {len(a[1])}; previous launch: {ref[i - 1][0] if i else '-'}; next: {ref[i + 1][0] if i + 1 < len(ref) else '-'})", flush=True)
                    if keeps[i] is not None and ref_keeps[i] is not None:
                        for j, (u, v) in enumerate(zip(ref_keeps[i], keeps[i])):
                            if u.shape == v.shape and not torch.equal(u, v):
                                dd = (u.reshape(-1).view(torch.int32) != v.reshape(-1).view(torch.int32)).nonzero().flatten()
                                ad = (u.reshape(-1)[dd] - v.reshape(-1)[dd]).abs()
                                last = u.shape[-1]
                                cols, rows = (dd % last), (dd // last)
                                print(f"    kept tensor {j} shape {tuple(u.shape)}: {dd.numel()} words differ; rows {rows.unique()[:8].tolist()} ({rows.unique().numel()} distinct), "
                                      f"cols min {int(cols.min())} max {int(cols.max())} distinct {cols.unique().numel()}, first cols {cols[:12].tolist()}, max |diff| {ad.max().item():.3e}, "
                                      f"median |diff| {ad.median().item():.3e}", flush=True)
                    break
    print(f"proc {rank}: done, {len(ref)} wrapped calls per forward", flush=True)
    sys.exit(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(REPS), mode], env=env, cwd=ROOT) for r in range(N)]
print("exit codes", [p.wait() for p in procs])
