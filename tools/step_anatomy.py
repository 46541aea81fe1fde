"""where one bench step (one sampler chunk through roll_out_serial) spends its wall time: host work before the graph launch, the graph on
the GPU, host work after it (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from datetime import datetime
import bench
from ladcast_amd.models import LaDCastTransformer3DModel
from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).to(dev).eval().set_gemm_precision("bf16x3")
model.enable_hip_graph(True)
pipe = AutoRegressive2DPipeline(model, EDMDPMSolverMultistepScheduler())
ic = (0.5 * torch.randn(84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).to(dev)
targs = {"mean": [0.0] * 84, "std": [1.0] * 84, "target_std": 0.5}


def step():
    return roll_out_serial(None, [datetime(2018, 1, 1, 0)], pipe, ensemble_size=1, num_inference_steps=20, return_seq_len=4, latent_transform_args=targs,
                           total_lead_time_hour=6, sampler_type=sys.argv[1] if len(sys.argv) > 1 else "edm", return_latent=True, known_latents_override=ic)


for _ in range(3):
    step()
torch.cuda.synchronize()
marks = {}
orig = torch.cuda.CUDAGraph.replay


def replay(self):
    marks["call"] = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(self)
    e1.record()
    marks["returned"] = time.perf_counter()
    marks["ev"] = (e0, e1)


torch.cuda.CUDAGraph.replay = replay
rows = []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    e0, e1 = marks["ev"]
    gpu = e0.elapsed_time(e1)
    rows.append(((marks["call"] - t0) * 1e3, (marks["returned"] - marks["call"]) * 1e3, gpu, (t1 - t0) * 1e3))
for r in rows:
    print("host before launch %6.2f ms | launch call %6.2f ms | graph on GPU %7.2f ms | step %7.2f ms | step - graph %5.2f ms" % (r[0], r[1], r[2], r[3], r[3] - r[2]))
