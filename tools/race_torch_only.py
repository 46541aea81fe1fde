"""Control experiment for the GPU-sharing reproducibility question: NO kernel of this repository - N processes share the GPU, each runs
chains of stock torch kernels (elementwise producer -> matmul consumer -> layer_norm -> matmul), back to back without host syncs, and counts
the results that differ from its first.  If stock kernels show the same effect, the stale-read between dependent launches of one stream
under concurrent queues is a platform property, not a race in ladcast_amd's kernels.  usage: python tools/race_torch_only.py N REPS"""
import os, subprocess, sys
if sys.argv[1] == "worker":
    rank, reps = int(sys.argv[2]), int(sys.argv[3])
    import torch
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 2250, 1536, generator=g).cuda()
    w1 = (torch.randn(1536, 6144, generator=g) / 39).cuda()
    w2 = (torch.randn(6144, 1536, generator=g) / 78).cuda()
    sc = torch.randn(1536, generator=g).cuda()

    def chain():
        h = x
        for _ in range(6):
            n = torch.nn.functional.layer_norm(h, (1536,)) * (1 + 0.1 * sc) + 0.01 * sc  # small producer kernels
            u = torch.nn.functional.gelu(n @ w1, approximate="tanh")                    # big consumer
            h = h + u @ w2
        return h

    ref = chain().clone()
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for r0 in range(0, reps, 4):
        outs = [chain() for _ in range(min(4, reps - r0))]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                worst = max(worst, (o - ref).abs().max().item())
    print(f"torch-only proc {rank}: {bad} of {reps} chains differ from the first, worst abs diff {worst:.3e} (output std {ref.std().item():.3f})", flush=True)
    sys.exit(0)
N, REPS = int(sys.argv[1]), int(sys.argv[2])
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(REPS)], env=env) for r in range(N)]
rc = [p.wait() for p in procs]
