"""RCCL executed for real on ONE GPU: a world-size-1 `nccl` (= RCCL on ROCm) process group in a fresh child process.

It proves nothing about xGMI, but it runs every RCCL call site the N > 1 path has (evaluate/pred_rollout.py:358,398-403 is what they
replace): `init_process_group("nccl", device_id=...)` with the bench's timeout handling, the probe all-reduce, the device-side padded
`all_gather` of `gather_members` / `gather_work` (both the one-collective latent form and the chunked form), `torch.cuda.nccl.version()`,
`roll_out_sharded` around the product's `roll_out_serial` on a tiny model - and compares every result bit for bit with the same call made
with no process group.

    python -m benchlib.rccl_world1            # the child: prints ONE JSON line
    benchlib.rccl_world1.run(timeout)         # the parent side (bench.py, tests): starts the child, returns the parsed line
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(timeout=240.0):
    """start the child (this process may already have initialised the GPU: a child process is started and waited for, never an exec) and
    return its JSON line; {"ok": False, "error": ...} when it fails or times out - the caller decides whether that is fatal"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-m", "benchlib.rccl_world1"], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return dict(ok=False, error=f"the world-size-1 nccl child did not finish within {timeout:.0f} s")
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return dict(ok=False, error=f"child exit code {r.returncode}: {(r.stderr or r.stdout)[-600:]}")
    out = json.loads(lines[-1])
    out["child_seconds"] = round(time.perf_counter() - t0, 1)
    return out


def _child():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from datetime import datetime

    import torch
    import torch.distributed as dist

    from benchlib.launch import init_process_group

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"

    from ladcast_amd.models import LaDCastTransformer3DModel
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.pipelines.distributed import gather_members, gather_work, roll_out_sharded
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    # (1) everything once WITHOUT a process group: the reference results
    g = torch.Generator().manual_seed(7)
    members = torch.randn(1, 3, 84, 5, 15, 30, generator=g).to(dev)  # (n_init, members, C, 1 + steps, h, w): a rank's result block
    items = torch.randn(3, 84, 5, 15, 30, generator=g).to(dev)
    tiny = dict(in_channels=84, out_channels=84, num_attention_heads=2, attention_head_dim=128, num_layers=1, num_single_layers=1, num_refiner_layers=1,
                mlp_ratio=4, patch_size=1, patch_size_t=1, qk_norm="rms_norm", rope_theta=256.0, rope_axes_dim=(16, 56, 56),
                rope_spatial_grid_start_pos=(-499.5, 5.25), rope_spatial_grid_end_pos=(508.5, 353.25), spatial_deg2rad=True,
                conditioning_tensor_in_channels=84, conditioning_tensor_rope_axes_dim=(16, 56, 56), incl_time_elapsed=True)
    torch.manual_seed(1234)
    model = LaDCastTransformer3DModel.from_config(tiny).to(dev).eval().set_gemm_precision("fp32")
    pipe = AutoRegressive2DPipeline(model, EDMDPMSolverMultistepScheduler())
    ic = (0.5 * torch.randn(84, 1, 15, 30, generator=torch.Generator().manual_seed(2))).to(dev)
    kw = dict(pipeline=pipe, num_inference_steps=3, return_seq_len=4, latent_transform_args={"mean": [0.0] * 84, "std": [1.0] * 84, "target_std": 0.5},
              total_lead_time_hour=24, sampler_type="edm", return_latent=True, known_latents_override=ic)

    def sharded():
        return roll_out_sharded(roll_out_serial, 2, [datetime(2018, 1, 1, 0)], device=dev, input_fields=None, **kw)

    assert not dist.is_initialized()
    want = dict(members=gather_members(members, 3, member_dim=1), work=gather_work(items, 1, 3, device=dev, item_shape=(84, 5, 15, 30)),
                work_chunked=gather_work(items, 1, 3, device=dev, max_bytes=4 * 84 * 5 * 15 * 30), rollout=sharded())
    torch.cuda.synchronize()

    # (2) the same calls inside a world-size-1 RCCL group, brought up exactly as bench.py brings up the N > 1 group
    t0 = time.perf_counter()
    init_process_group("nccl", dev, 1, 0, 120.0)
    t_init = time.perf_counter() - t0
    assert dist.is_initialized() and dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    t0 = time.perf_counter()
    got = dict(members=gather_members(members, 3, member_dim=1), work=gather_work(items, 1, 3, device=dev, item_shape=(84, 5, 15, 30)),
               work_chunked=gather_work(items, 1, 3, device=dev, max_bytes=4 * 84 * 5 * 15 * 30), rollout=sharded())
    stats = torch.tensor([1.0, 2.0, 3.0], device=dev, dtype=torch.float64)
    allr = [torch.empty_like(stats)]
    dist.all_gather(allr, stats)  # the per-rank statistics exchange of the N > 1 bench line
    tmax = torch.tensor([4.5], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    t_coll = time.perf_counter() - t0
    equal = {k: bool(got[k].is_cuda and torch.equal(got[k], want[k])) for k in want}
    equal["stats_all_gather"] = bool(torch.equal(allr[0], stats)) and tmax.item() == 4.5
    version = ".".join(str(v) for v in torch.cuda.nccl.version())
    dist.destroy_process_group()
    print(json.dumps(dict(ok=all(equal.values()), backend="nccl", world_size=1, rccl_version=version, bit_equal_to_no_group=equal,
                          init_seconds=round(t_init, 2), collectives_seconds=round(t_coll, 3),
                          call_sites=["init_process_group(nccl, device_id, timeout)", "probe all_reduce", "gather_members (padded device-side all_gather)",
                                      "gather_work (one all_gather with a status word)", "gather_work (chunked all_gathers)",
                                      "roll_out_sharded(roll_out_serial) on a tiny model", "all_gather of per-rank statistics", "all_reduce MAX", "barrier",
                                      "destroy_process_group"],
                          note="ONE GPU: no byte crosses xGMI; this executes the RCCL call sites of the N > 1 path, it is not a scaling measurement")), flush=True)


if __name__ == "__main__":
    _child()
