"""world_size-2 gloo test of the member-sharding / gather path (CPU; the compute stand-in is the oracle
sampler, since the point is the partition + collective logic, not the kernels)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ladcast_amd.pipelines.distributed import gather_members, members_per_rank, shard_members


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ens, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import pipelines as OP
    from oracle.scheduler import EDMDPMSolverMultistepScheduler
    from tests.synth import make_ar, synth_known, tiny_ar_config

    m = make_ar(tiny_ar_config())
    pipe = OP.AutoRegressive2DPipeline(m, EDMDPMSolverMultistepScheduler())
    known, ts = synth_known(1), torch.tensor([2018010100])
    ids = shard_members(ens, rank, world)
    local = OP.ensemble_AR_sampler(pipe, len(ids), 2, 2, known_latents=known, timestamps=ts, sampler_type="edm", member_ids=ids)
    full = gather_members(local, ens, member_dim=0)
    if rank == 0:
        torch.save(full, result_path)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bookkeeping():
    assert shard_members(16, 3, 8) == [3, 11]
    assert shard_members(1, 1, 2) == [] and shard_members(3, 1, 2) == [1]
    assert members_per_rank(5, 2) == [3, 2]
    assert sorted(sum((shard_members(13, r, 4) for r in range(4)), [])) == list(range(13))
    x = torch.arange(6.0).reshape(3, 2)
    assert torch.equal(gather_members(x, 3), x)  # no process group: identity


def test_two_rank_sharded_ensemble_equals_single_process(tmp_path):
    ens = 3  # uneven split: rank 0 owns members {0, 2}, rank 1 owns {1}
    path = str(tmp_path / "full.pt")
    mp.spawn(_worker, args=(2, _free_port(), ens, path), nprocs=2, join=True)
    got = torch.load(path)
    from oracle import pipelines as OP
    from oracle.scheduler import EDMDPMSolverMultistepScheduler
    from tests.synth import make_ar, synth_known, tiny_ar_config

    m = make_ar(tiny_ar_config())
    pipe = OP.AutoRegressive2DPipeline(m, EDMDPMSolverMultistepScheduler())
    want = OP.ensemble_AR_sampler(pipe, ens, 2, 2, known_latents=synth_known(1), timestamps=torch.tensor([2018010100]), sampler_type="edm")
    assert got.shape == want.shape
    assert ((got - want).norm() / want.norm()).item() < 1e-5  # batch-size dependent BLAS blocking only
