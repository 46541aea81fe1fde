"""N processes share the GPU; each runs REPS 375M forwards back to back (eager, no host sync in between) on fixed inputs and counts the
outputs that differ from its first.  usage: python tools/race_forward.py N REPS mode   (environment: LDC_LIB_PATH / LDC_* of the A/B build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "worker":
    rank, reps, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    B = int(os.environ.get("RACE_B", "1"))
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from ladcast_amd.models import LaDCastTransformer3DModel
    torch.manual_seed(1234)
    model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to("cuda").eval().set_gemm_precision(mode)
    x = torch.randn(B, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    known = (0.5 * torch.randn(B, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
    ts = torch.tensor([2018010100]).cuda()
    t = torch.tensor([0.3]).cuda()
    model(x, t, known, time_elapsed=ts)
    torch.cuda.synchronize()
    outs = [model(x, t, known, time_elapsed=ts).sample.clone() for _ in range(reps)]
    torch.cuda.synchronize()
    bad = [i for i in range(reps) if not torch.equal(outs[i], outs[0])]
    worst = max([(outs[i] - outs[0]).abs().max().item() for i in bad], default=0.0)
    print(f"proc {rank} [{mode}, B={B}]: {len(bad)} of {reps} forwards differ from the first, worst abs diff {worst:.3e} (output std {outs[0].std().item():.3f})", flush=True)
    sys.exit(0)
N, REPS, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(REPS), mode], env=env, cwd=ROOT) for r in range(N)]
rc = [p.wait() for p in procs]
