"""Soak: replay the captured 375M Heun chunk N times (alternating two workloads) and check every result of each workload against its
first bit for bit; prints the replay count, wall time and the number of mismatching replays (expected 0).
usage: python tools/soak_chunk.py [N=1500] [precision=bf16x3] [model=375M] [members=1]   (1.6B: the attention's TAIL schedule is on the path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ladcast_amd.models import LaDCastTransformer3DModel
from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
model_name = sys.argv[3] if len(sys.argv) > 3 else "375M"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.manual_seed(1234)
g = LaDCastTransformer3DModel.from_config(bench.CONFIGS[model_name]).cuda().eval().set_gemm_precision(prec).enable_hip_graph(True)
pipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
ts = torch.tensor([2018010100]).cuda()
known = [torch.randn(1, 84, 1, 15, 30, device="cuda") * s for s in (0.5, 0.9)]
run = lambda k, m: ensemble_AR_sampler(pipe, B, 4, 20, known_latents=known[k], timestamps=ts, sampler_type="edm", device="cuda", member_ids=[m + j for j in range(B)])
ref = [run(0, 0), run(1, 3)]
assert all(torch.isfinite(r).all() for r in ref)
bad, t0 = 0, time.time()
for i in range(n):
    k = i & 1
    if not torch.equal(run(k, 3 * k), ref[k]):
        bad += 1
dt = time.time() - t0
print(f"{model_name} x {B} member(s), {prec}: {n} chunk replays ({39 * n} forwards, {(39 * 36 + 20) * n} kernel launches) in {dt:.0f} s, {1e3 * dt / n:.1f} ms each: {bad} mismatching replays")
