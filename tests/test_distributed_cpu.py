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


# -- 2-D (initial time x member) sharding: evaluate/pred_rollout.py:349-358 re-cut so that ens < world and 20 members / 8 ranks work ----
from ladcast_amd.pipelines.distributed import gather_work, roll_out_sharded, shard_work, work_bounds  # noqa: E402


def test_work_sharding_bookkeeping():
    assert work_bounds(20, 8) == [0, 2, 5, 7, 10, 12, 15, 17, 20]
    sizes = lambda n, e, w: [sum(len(ids) for _, ids in shard_work(n, e, r, w)) for r in range(w)]  # noqa: E731
    assert sizes(1, 20, 8) == [2, 3, 2, 3, 2, 3, 2, 3]
    assert sizes(2, 20, 8) == [5] * 8  # the README's 20 members on 8 ranks balance exactly over two initial times
    assert sizes(1, 1, 8) == [0, 0, 0, 0, 0, 0, 0, 1]  # ens < world: seven ranks without work
    assert sizes(8, 16, 8) == [16] * 8  # cfg 3 with one initial time per rank == the reference's own split
    assert shard_work(2, 3, 1, 2) == [(1, [0, 1, 2])] and shard_work(2, 3, 0, 2) == [(0, [0, 1, 2])]
    assert shard_work(3, 3, 0, 2) == [(0, [0, 1, 2]), (1, [0])] and shard_work(3, 3, 1, 2) == [(1, [1, 2]), (2, [0, 1, 2])]
    for n, e, w in ((1, 1, 2), (3, 5, 4), (2, 20, 8), (5, 3, 8)):
        items = sum(([(t, k) for k in ids] for r in range(w) for t, ids in shard_work(n, e, r, w)), [])
        assert items == [(t, k) for t in range(n) for k in range(e)]  # every item once, time-major order
    x = torch.arange(24.0).reshape(6, 4)
    assert torch.equal(gather_work(x, 2, 3), x.reshape(2, 3, 4))  # no process group: a reshape


def _fake_rollout(ensemble_size, member_ids, pred_timestamp, calls=None, **kw):
    """roll_out_serial's tensor contract with a closed-form value per (initial time, member, lead step): (1, ens, C, 1+steps, h, w)"""
    assert len(pred_timestamp) == 1 and len(member_ids) == ensemble_size > 0
    if calls is not None:
        calls.append((pred_timestamp[0], list(member_ids)))
    out = torch.empty(1, ensemble_size, 2, 3, 2, 2)
    for j, k in enumerate(member_ids):
        out[0, j] = 1000.0 * pred_timestamp[0] + 10.0 * k + torch.arange(24.0).reshape(2, 3, 2, 2) / 100.0
    return out


def _worker_2d(rank, world, port, cases, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for n_init, ens in cases:
        calls = []
        full = roll_out_sharded(_fake_rollout, ensemble_size=ens, pred_timestamp=list(range(1, n_init + 1)), calls=calls)
        res.append((full, calls))
    torch.save(res, f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_2d_sharding_incl_empty_rank(tmp_path):
    """ens < world (one rank owns nothing: it must not launch and must not hang the gather), a non-divisible case, and an even one"""
    cases = [(1, 1), (3, 3), (2, 2), (1, 5)]
    path = str(tmp_path / "res")
    mp.spawn(_worker_2d, args=(2, _free_port(), cases, path), nprocs=2, join=True)
    r0, r1 = torch.load(path + ".0"), torch.load(path + ".1")
    for (n_init, ens), (f0, c0), (f1, c1) in zip(cases, r0, r1):
        want = torch.cat([_fake_rollout(ens, list(range(ens)), [t]) for t in range(1, n_init + 1)], dim=0)
        assert f0.shape == want.shape == (n_init, ens, 2, 3, 2, 2)
        assert torch.equal(f0, want) and torch.equal(f1, want)  # every rank holds the single-process result
        done = sorted((t, k) for t, ids in c0 + c1 for k in ids)
        assert done == [(t, k) for t in range(1, n_init + 1) for k in range(ens)]  # each item computed exactly once
    assert r0[0][1] == [] and r1[0][1] == [(1, [0])]  # case (1, 1): rank 0 had no work


def test_run_rollout_batches_and_files(tmp_path):
    """run_rollout without a process group: batches of initial times, one latent_YYYYMMDDHH.npy per initial time, reference layout"""
    from datetime import datetime
    import numpy as np
    from ladcast_amd.evaluate import pred_rollout as PR

    times = [datetime(2018, 1, d, 0) for d in (1, 2, 3)]
    seen = []

    def fake(ensemble_size, member_ids, pred_timestamp, **kw):
        seen.append((pred_timestamp[0], list(member_ids), kw["return_latent"], kw["latent_transform"], kw["encdec_model_type"]))
        return _fake_rollout(ensemble_size, member_ids, [pred_timestamp[0].day])

    orig = PR.roll_out_serial
    PR.roll_out_serial = fake
    try:
        res = PR.run_rollout(None, times, pipeline=None, encdec_model=None, latent_transform_args={}, output=str(tmp_path), ensemble_size=2,
                             total_lead_time_hour=12, batch_size=2)
    finally:
        PR.roll_out_serial = orig
    assert [s[:2] for s in seen] == [(t, [0, 1]) for t in times] and all(s[2:] == (True, "normalize", "ae") for s in seen)
    assert len(res) == 3 and res[1].shape == (2, 2, 3, 2, 2)
    for t, r in zip(times, res):
        f = np.load(os.path.join(str(tmp_path), t.strftime("latent_%Y%m%d%H.npy")))
        assert np.array_equal(f, r.numpy())


# -- IC perturbation (noise_level > 0) must not depend on the partition; chunked gather ---------------------------------------------
def _real_rollout_kwargs():
    """the PRODUCT's roll_out_serial on the CPU: host loop of the pipeline sampler with an elementwise toy network and a duck-typed
    scheduler (no HIP kernel on this path: latent_transform=None, sampler_type="pipeline", a given IC latent)"""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline
    from tests.synth import ToyNet
    from tests.synth import DuckDDIMScheduler, synth_known

    g = torch.Generator().manual_seed(5)
    return dict(input_fields=None, pipeline=AutoRegressive2DPipeline(ToyNet(84), DuckDDIMScheduler()), num_inference_steps=3, return_seq_len=2,
                latent_transform=None, latent_transform_args={"std": (torch.rand(84, generator=g) + 0.5).tolist()}, total_lead_time_hour=18,
                sampler_type="pipeline", return_latent=True, known_latents_override=synth_known(1)[0], noise_level=0.3)


def _worker_noise(rank, world, port, result_path):
    from datetime import datetime

    from ladcast_amd.pipelines import roll_out_serial

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different global RNG streams per process: the result must not see them
    times = [datetime(2018, 1, 1, 0), datetime(2018, 1, 2, 12)]
    full = roll_out_sharded(roll_out_serial, ensemble_size=3, pred_timestamp=times, **_real_rollout_kwargs())
    x = torch.arange(7 * 5 * 3, dtype=torch.float32).reshape(7, 5, 3) + 1000 * rank
    mine = work_bounds(14, world)
    chunked = gather_work(x[: mine[rank + 1] - mine[rank]], 2, 7, max_bytes=2 * 15 * 4, out_device="cpu")  # two rows per collective
    torch.save((full, chunked), f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_ic_noise_is_partition_invariant_and_gather_is_chunked(tmp_path):
    """2 initial times x 3 members on 2 ranks cut initial time 1 between the ranks (items 0-2 | 3-5 are whole
    times here, so also 1 x 3 on 2 ranks below); with noise_level > 0 every piece of an initial time must start from the SAME perturbed
    IC.  The 2-rank result equals the single-process result bit for bit, members of one initial time share slot 0, and initial times
    differ."""
    from datetime import datetime

    from ladcast_amd.pipelines import roll_out_serial

    path = str(tmp_path / "res")
    mp.spawn(_worker_noise, args=(2, _free_port(), path), nprocs=2, join=True)
    (f0, c0), (f1, c1) = torch.load(path + ".0"), torch.load(path + ".1")
    times = [datetime(2018, 1, 1, 0), datetime(2018, 1, 2, 12)]
    torch.manual_seed(7)
    single = roll_out_sharded(roll_out_serial, ensemble_size=3, pred_timestamp=times, **_real_rollout_kwargs())
    assert single.shape == (2, 3, 84, 4, 15, 30) and torch.isfinite(single).all()
    assert torch.equal(f0, single) and torch.equal(f1, single)
    # the perturbation is real, shared by the members of an initial time, different between initial times
    clean = roll_out_sharded(roll_out_serial, ensemble_size=3, pred_timestamp=times, **dict(_real_rollout_kwargs(), noise_level=0))
    assert not torch.equal(single[:, :, :, 1:], clean[:, :, :, 1:])
    kw = dict(_real_rollout_kwargs(), ic_noise_seed=0)
    a = roll_out_serial(pred_timestamp=times[:1], ensemble_size=1, member_ids=[2], **kw)
    assert torch.equal(a[0, 0], single[0, 2])  # a piece holding only member 2 of the first initial time
    assert not torch.equal(single[0, 0, :, 1:], single[1, 0, :, 1:])
    # the reference's own behaviour (global RNG) is still there when no seed is given
    torch.manual_seed(1)
    r1 = roll_out_serial(pred_timestamp=times[:1], ensemble_size=1, **_real_rollout_kwargs())
    torch.manual_seed(2)
    r2 = roll_out_serial(pred_timestamp=times[:1], ensemble_size=1, **_real_rollout_kwargs())
    assert not torch.equal(r1[:, :, :, 1:], r2[:, :, :, 1:])
    # chunked gather: 14 items of (5, 3) over 2 ranks in collectives of two rows
    want = torch.cat([torch.arange(7 * 5 * 3, dtype=torch.float32).reshape(7, 5, 3) + 1000 * r for r in range(2)]).reshape(2, 7, 5, 3)
    assert torch.equal(c0, want) and torch.equal(c1, want)


# -- single all_gather in latent mode; return_ensemble_mean under sharding -----------------------------------------
def _fake_decoded_rollout(ensemble_size, member_ids, pred_timestamp, return_ensemble_mean=False, **kw):
    """decoded-mode contract: slot 0 = the IC field (the same for every member), slots 1.. differ per member; honours
    return_ensemble_mean the way roll_out_serial does (one row holding the fp32 mean over the members, pipelines/utils.py:296-300,641)"""
    g = torch.Generator().manual_seed(1000 * pred_timestamp[0])
    ic = torch.randn(2, 3, 4, generator=g)
    rows = []
    for k in member_ids:
        x = torch.randn(2, 4, 3, 4, generator=torch.Generator().manual_seed(77 * pred_timestamp[0] + k))
        x[:, 0] = ic
        rows.append(x)
    out = torch.stack(rows)[None]
    if return_ensemble_mean:
        m = out.mean(dim=1, keepdim=True)
        m[:, 0, :, 0] = ic
        return m
    return out


def _worker_mean_and_single_gather(rank, world, port, result_path):
    from datetime import datetime

    from ladcast_amd.pipelines import roll_out_serial

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mean = roll_out_sharded(_fake_decoded_rollout, ensemble_size=5, pred_timestamp=[1, 2, 3], return_ensemble_mean=True)
    # latent mode through the PRODUCT's roll_out_serial: the item shape is known on every rank -> exactly one collective, an all_gather
    # (1 initial time x 1 member on 2 ranks: rank 0 has no work and must still take part with the right shape)
    calls = []
    orig_ag, orig_ar = dist.all_gather, dist.all_reduce
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), orig_ag(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), orig_ar(*a, **k))[1]
    try:
        kw = dict(_real_rollout_kwargs(), noise_level=0)
        one = roll_out_sharded(roll_out_serial, ensemble_size=1, pred_timestamp=[datetime(2018, 1, 1, 0)], **kw)
        three = roll_out_sharded(roll_out_serial, ensemble_size=3, pred_timestamp=[datetime(2018, 1, 1, 0)], **kw)
    finally:
        dist.all_gather, dist.all_reduce = orig_ag, orig_ar
    torch.save((mean, one, three, calls), f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ensemble_mean_and_single_all_gather(tmp_path):
    from datetime import datetime

    from ladcast_amd.pipelines import roll_out_serial
    from ladcast_amd.pipelines.distributed import latent_item_shape

    path = str(tmp_path / "res")
    mp.spawn(_worker_mean_and_single_gather, args=(2, _free_port(), path), nprocs=2, join=True)
    (m0, one0, three0, calls0), (m1, one1, three1, calls1) = torch.load(path + ".0"), torch.load(path + ".1")
    want = torch.cat([_fake_decoded_rollout(5, list(range(5)), [t], return_ensemble_mean=True) for t in (1, 2, 3)], dim=0)
    assert m0.shape == want.shape == (3, 1, 2, 4, 3, 4)
    assert torch.allclose(m0, want, rtol=0, atol=1e-6) and torch.equal(m0, m1)  # same fp32 mean of the same members
    assert torch.equal(m0[:, 0, :, 0], want[:, 0, :, 0])  # slot 0 (the IC) copied, not averaged
    assert calls0 == ["all_gather", "all_gather"] and calls1 == calls0  # ONE collective per sharded call, no shape-agreement round
    kw = dict(_real_rollout_kwargs(), noise_level=0)
    assert latent_item_shape(dict(kw, pipeline=kw["pipeline"])) == (84, 4, 15, 30)
    s1 = roll_out_serial(pred_timestamp=[datetime(2018, 1, 1, 0)], ensemble_size=1, **kw)
    s3 = roll_out_serial(pred_timestamp=[datetime(2018, 1, 1, 0)], ensemble_size=3, **kw)
    assert torch.equal(one0, s1) and torch.equal(one1, s1) and torch.equal(three0, s3) and torch.equal(three1, s3)
    with __import__("pytest").raises(ValueError):
        roll_out_sharded(_fake_decoded_rollout, ensemble_size=2, pred_timestamp=[1], return_ensemble_mean=True, return_latent=True)


def _worker_bad_shape(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    # one item, two ranks: rank 0 has no work, rank 1 owns the item - with a shape other than the announced one
    local = None if rank == 0 else torch.ones(1, 2, 3)
    try:
        gather_work(local, 1, 1, item_shape=(2, 4))
        msg = "no error"
    except ValueError as e:
        msg = str(e)
    torch.save(msg, f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_announced_shape_mismatch_raises_on_every_rank_instead_of_hanging(tmp_path):
    """a rank whose items do not have the announced shape used to raise before the all_gather the other ranks were
    already waiting in (a hang until the collective's timeout).  Now the error travels with the collective and every rank raises."""
    path = str(tmp_path / "msg")
    mp.spawn(_worker_bad_shape, args=(2, _free_port(), path), nprocs=2, join=True)
    m0, m1 = torch.load(path + ".0"), torch.load(path + ".1")
    assert "rank(s) [1]" in m0 and "announced item shape (2, 4)" in m0
    assert "rank(s) [1]" in m1 and "(here: (2, 3))" in m1
