"""N processes share ONE GPU with a MIXED load: even ranks run 375M forwards (B = 2: the 4-wave split attention, the AdaLN GEMV inside every
forward), odd ranks run full-size DCAE encode + decode; every process repeats its work REPS times on fixed inputs and counts the results that
differ from its first.  All zero = every kernel of both models is bit-reproducible next to the other model's kernels of other processes.
usage: python tools/race_mixed.py N REPS mode"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "worker":
    rank, reps, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sys.path.insert(0, ROOT)
    import torch
    import bench
    torch.manual_seed(1234)
    if rank % 2 == 0:
        from ladcast_amd.models import LaDCastTransformer3DModel
        model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to("cuda").eval().set_gemm_precision(mode)
        x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
        known = (0.5 * torch.randn(2, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
        ts, t = torch.tensor([2018010100]).cuda(), torch.tensor([0.3]).cuda()
        fn = lambda: [model(x, t, known, time_elapsed=ts).sample.clone()]
        what = "375M forward, B = 2"
    else:
        from ladcast_amd.models import AutoencoderDC
        g = AutoencoderDC.from_config(bench.CONFIG_DCAE_84).cuda().eval().set_gemm_precision(mode)
        xf = torch.randn(2, 84, 120, 240, generator=torch.Generator().manual_seed(5)).cuda()
        st = torch.randn(1, 5, 120, 240, generator=torch.Generator().manual_seed(6)).cuda()
        def fn():
            z = g.encode(xf, static_conditioning_tensor=st).latent
            return [z.clone(), g.decode(z).sample.clone()]
        what = "DCAE encode + decode, 2 frames"
    first = fn()
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for _ in range(reps):
        out = fn()
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(out, first)):
            bad += 1
            worst = max(worst, max((a - b).abs().max().item() for a, b in zip(out, first)))
    print(f"proc {rank} [{what}, {mode}]: {bad} of {reps} repeats differ from the first, worst abs diff {worst:.3e}", flush=True)
    sys.exit(0)
N, REPS, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(REPS), mode], env=env, cwd=ROOT) for r in range(N)]
rc = [p.wait() for p in procs]
sys.exit(max(rc))
