"""N processes share the GPU; each runs REPS x prepare_conditioning (20 noise levels x 2 members, 375M) back to back and reports which of the
conditioning path's buffers differ from the first repeat.  usage: python tools/race_cond.py N REPS mode"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "worker":
    rank, reps, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from ladcast_amd.models import LaDCastTransformer3DModel
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    torch.manual_seed(1234)
    model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to("cuda").eval().set_gemm_precision(mode)
    known = (0.5 * torch.randn(2, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
    ts = torch.tensor([2018010100]).cuda()
    sch = EDMDPMSolverMultistepScheduler()
    sch.set_timesteps(20)
    cn = sch.precondition_noise(sch.sigmas[:-1]).cuda()
    te = model.time_elapsed_embedding(ts)
    model.prepare_conditioning(cn, known, te)
    torch.cuda.synchronize()
    wskey = [k for k in model._ws if isinstance(k, tuple) and k and k[0] == "cond"][0]
    ws = model._ws[wskey]
    names = ["ctok", "ctx0", "pooled", "t1", "t2", "t1m", "t2m", "p1", "temb_r", "nh", "qkv", "att", "mod_a", "cat", "h", "temb", "mods"]
    snaps = []
    for _ in range(reps):
        model.prepare_conditioning(cn, known, te)
        snaps.append({n: getattr(ws, n).clone() for n in names})
    torch.cuda.synchronize()
    bad = {}
    for i in range(1, reps):
        for n in names:
            if not torch.equal(snaps[i][n], snaps[0][n]):
                d = (snaps[i][n].float() - snaps[0][n].float()).abs()
                d = d[torch.isfinite(d)]
                bad.setdefault(n, []).append(round(d.max().item(), 9) if d.numel() else float("nan"))
    print(f"proc {rank} [{mode}]: buffers that differ in some repeat (of {reps}): " + (", ".join(f"{n} x{len(v)} (worst {max(v):.2e})" for n, v in bad.items()) or "none"), flush=True)
    sys.exit(0)
N, REPS, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(REPS), mode], env=env, cwd=ROOT) for r in range(N)]
rc = [p.wait() for p in procs]
