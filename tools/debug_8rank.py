"""debug aid: is the 375M chunk bitwise reproducible when N processes share one GPU?  N independent single-rank processes at the same
time, each repeating the same 2-member chunk; variants: whole-chunk hipGraph / eager launches.  usage: debug_8rank.py N graph|eager [mode]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, variant = int(sys.argv[1]), sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
procs = []
for r_ in range(N):
    dump = f"/tmp/solo_{r_}.pt"
    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from datetime import datetime
import bench
from ladcast_amd.models import LaDCastTransformer3DModel
from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
torch.manual_seed(1234)
model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to("cuda").eval().set_gemm_precision({mode!r})
model.enable_hip_graph({variant == "graph"})
pipe = AutoRegressive2DPipeline(model, EDMDPMSolverMultistepScheduler())
ic = (0.5 * torch.randn(84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
kw = dict(num_inference_steps=20, return_seq_len=4, latent_transform_args={{"mean": [0.0] * 84, "std": [1.0] * 84, "target_std": 0.5}},
          total_lead_time_hour=24, sampler_type="edm", return_latent=True, known_latents_override=ic)
res = [roll_out_serial(None, [datetime(2018, 1, 1, 0)], pipe, ensemble_size=2, member_ids=[0, 8], **kw) for _ in range(6)]
torch.save(torch.stack(res), {dump!r})
"""
    procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT))
for p_ in procs:
    p_.wait()
import torch
solo = [torch.load(f"/tmp/solo_{r_}.pt") for r_ in range(N)]
ref = solo[0][0]
for r_ in range(N):
    flags = [torch.equal(solo[r_][j], ref) for j in range(6)]
    worst = max((solo[r_][j] - ref).abs().max().item() for j in range(6))
    print(f"N={N} {variant} {mode} process {r_}: 6 repeats equal to process 0's first: {flags}  worst abs diff {worst:.3e}", flush=True)
