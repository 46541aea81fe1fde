"""VERDICT r04 item 7, second half: does running the DCAE decode of chunk k on a second stream, beside the AR sampler chunk k + 1, pay?
cfg5's share of one GPU: 375M AR (1 member by default, 20-step Heun chunk as one hipGraph) + decode of the chunk's 4 frames (one hipGraph).
Both models capture on their own side stream, so their stream-K / attention workspaces are distinct and the two graphs may run side by side.
Prints sequential vs overlapped time per (chunk + decode) and checks that the overlapped results equal the sequential ones bit for bit.
usage: python tools/overlap_probe.py [members] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ladcast_amd.models import AutoencoderDC, LaDCastTransformer3DModel
from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

members = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
torch.manual_seed(1234)
ar = LaDCastTransformer3DModel.from_config(bench.CONFIGS["375M"]).cuda().eval().set_gemm_precision(prec)
ar.enable_hip_graph(True)
ae = AutoencoderDC.from_config(bench.CONFIG_DCAE_84).cuda().eval().set_gemm_precision(prec).enable_hip_graph(True)
pipe = AutoRegressive2DPipeline(ar, EDMDPMSolverMultistepScheduler())
known = (0.5 * torch.randn(1, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
ts = torch.tensor([2018010100]).cuda()


def chunk(k):
    return ensemble_AR_sampler(pipe, members, 4, 20, known_latents=k, timestamps=ts, sampler_type="edm", device="cuda")


def decode(lat):  # (members, 84, 4, 15, 30) -> frames, at most GRAPH_MAX_FRAMES per call
    z = lat.permute(0, 2, 1, 3, 4).reshape(-1, 84, 15, 30).contiguous()
    return torch.cat([ae.decode(z[i : i + 4]).sample for i in range(0, z.shape[0], 4)])


N = 6
lat = chunk(known); dec = decode(lat); torch.cuda.synchronize()  # captures both graphs
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def sequential():
    outs, k = [], known
    for _ in range(N):
        lat = chunk(k)
        outs.append(decode(lat))
        k = lat[:1, :, -1:].contiguous()
    return outs


def overlapped():
    outs, k = [], known
    with torch.cuda.stream(s1):
        lat = chunk(k)
    for i in range(N):
        ev = torch.cuda.Event()
        with torch.cuda.stream(s1):
            ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            outs.append(decode(lat))  # chunk i's frames on stream 2 ...
        if i + 1 < N:
            with torch.cuda.stream(s1):
                k = lat[:1, :, -1:].contiguous()
                lat = chunk(k)  # ... beside chunk i + 1 on stream 1
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)
    return outs


res = {}
for name, fn in (("sequential", sequential), ("overlapped", overlapped), ("sequential", sequential), ("overlapped", overlapped)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.setdefault(name, []).append((dt / N, outs))
    print(f"{name:11s}: {1e3 * dt / N:8.2f} ms per (chunk + decode of {4 * members} frames), {members} member(s), {prec}", flush=True)
same = all(torch.equal(a, b) for a, b in zip(res["sequential"][0][1], res["overlapped"][0][1]))
print("overlapped results equal the sequential ones bit for bit:", same)
best = {k: min(v[0] for v in vs) for k, vs in res.items()}
print(f"gain of overlapping: {100 * (1 - best['overlapped'] / best['sequential']):.1f} % ({1e3 * best['sequential']:.2f} -> {1e3 * best['overlapped']:.2f} ms)")
