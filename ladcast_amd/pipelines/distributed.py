"""Multi-GPU sharding of the rollout: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The path shards over (initial time x ensemble MEMBER) work items, never over lead time: members
are independent given the shared IC latent, member k is always seeded with k
(pipelines/utils.py:703-706), and lead-time chunks of one member are a sequential chain (:563).
So there is no data-path collective; the only exchange is the reference's own
``accelerator.gather(result_tensor)`` at the end (evaluate/pred_rollout.py:398-400), done here
as ONE all_gather of the per-rank result block.  On an 8-GPU xGMI mesh that is a 12 MB-per-rank
message over direct links -- never on the critical path, no ring/tree tuning needed.

Work split.  The reference deals whole initial times to ranks
(``accelerator.split_between_processes``, evaluate/pred_rollout.py:349-358): with fewer initial
times than ranks GPUs idle, and a 20-member forecast never uses more than one GPU.  Here the
``n_init * ensemble_size`` items of a batch of initial times, in time-major order, are cut into
``world`` contiguous blocks whose sizes differ by at most one (``shard_work``): 1 initial time x 20
members on 8 ranks is 2/3/2/3/2/3/2/3 (nothing better exists), 2 initial times x 20 members is 5 items
everywhere, and ``ensemble_size < world`` leaves ``world - n_items`` ranks without work - they launch
nothing and only join the gather.  Results do not depend on the partition - to fp32 rounding, and bit for bit whenever a member is
computed in a batch of the same size (the kernels pick their schedule - stream-K cut of the GEMMs, key split of the attention - from the
whole launch, batch included, and different schedules add the same terms in a different order: ~1e-7 on the fp32 path,
tests/test_gpu_bench_launch.py) - : member k is seeded with k, and the optional IC
perturbation (``noise_level > 0``: ONE draw per initial time shared by its members, pipelines/utils.py:518-528) is drawn from a
generator seeded by the initial time (``ic_noise_seed``, set here when the caller did not), not from a per-process RNG stream.

The helpers are device-agnostic (they only move tensors), so the N>1 logic is exercised on CPU
with the gloo backend in ``tests/test_distributed_cpu.py``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# -- 1-D: members of one initial time (bench.py's weak-scaling layout) ------------------------------
def shard_members(ensemble_size: int, rank: int, world_size: int) -> List[int]:
    """Round-robin member ownership: rank r computes members {k : k mod world_size == r}."""
    return list(range(rank, ensemble_size, world_size))


def members_per_rank(ensemble_size: int, world_size: int) -> List[int]:
    return [len(range(r, ensemble_size, world_size)) for r in range(world_size)]


def _active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized()


def gather_members(local: torch.Tensor, ensemble_size: int, member_dim: int = 0, group=None) -> torch.Tensor:
    """All-gather the per-rank member blocks and restore global member order.

    ``local``: this rank's members (in ``shard_members`` order) along ``member_dim``.  Ranks may own
    different member counts (ensemble_size not divisible by world size, or fewer members than ranks):
    blocks are padded to the largest count for the collective and trimmed afterwards.  Returns the full
    ensemble on every rank.
    """
    if not _active(group):
        assert local.shape[member_dim] == ensemble_size
        return local
    world = dist.get_world_size(group)
    counts = members_per_rank(ensemble_size, world)
    cmax = max(counts)
    x = local.movedim(member_dim, 0).contiguous()
    if x.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError("local block does not hold this rank's members")
    if x.shape[0] < cmax:
        pad = torch.zeros((cmax - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        x = torch.cat([x, pad], dim=0)
    bufs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(bufs, x, group=group)
    out = torch.empty((ensemble_size,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    for r in range(world):
        ids = shard_members(ensemble_size, r, world)
        if ids:
            out[ids] = bufs[r][: len(ids)]
    return out.movedim(0, member_dim)


# -- 2-D: (initial time x member) items ---------------------------------------------------------------
def work_bounds(n_items: int, world_size: int) -> List[int]:
    """block boundaries: rank r owns items [b[r], b[r+1]); sizes differ by at most one, larger blocks spread evenly"""
    return [(r * n_items) // world_size for r in range(world_size + 1)]


def shard_work(n_init: int, ensemble_size: int, rank: int, world_size: int) -> List[Tuple[int, List[int]]]:
    """This rank's items as ``[(init index, [member ids]), ...]`` (time-major item order, contiguous block)."""
    b = work_bounds(n_init * ensemble_size, world_size)
    groups: List[Tuple[int, List[int]]] = []
    for i in range(b[rank], b[rank + 1]):
        t, k = divmod(i, ensemble_size)
        if groups and groups[-1][0] == t:
            groups[-1][1].append(k)
        else:
            groups.append((t, [k]))
    return groups


GATHER_CHUNK_BYTES = 256 << 20  # per-rank bytes of one all_gather: bounds the device memory the collective itself needs


def gather_work(local: Optional[torch.Tensor], n_init: int, ensemble_size: int, group=None, device=None, out_device=None,
                max_bytes: int = GATHER_CHUNK_BYTES, item_shape: Optional[Sequence[int]] = None) -> torch.Tensor:
    """``local``: this rank's items ``(n_local, *item_shape)`` in ``shard_work`` order (``None`` / 0 rows for a rank without
    work) -> ``(n_init, ensemble_size, *item_shape)`` on every rank.  ``item_shape`` given (every rank can work it out - latent
    mode: ``(C, 1 + steps, h, w)`` is known without encoding): the exchange is ONE all_gather of the padded per-rank blocks, the
    single collective north_star describes (evaluate/pred_rollout.py:398-400), whenever a block fits ``max_bytes`` (latents: 12 MB
    per member at 240 h).  Without it one small all_reduce first tells ranks without work the item shape.  Blocks larger than
    ``max_bytes`` per rank (decoded fields of a long rollout - 0.8 GB per member at 240 h) move in several all_gathers, so the
    collective needs ``(world + 1) x max_bytes`` of scratch on ``device`` however large the batch is.  ``out_device``: where the
    assembled result lives (default: ``device``); pass "cpu" for decoded-field batches that should not be resident on every GPU."""
    if not _active(group):
        assert local is not None and local.shape[0] == n_init * ensemble_size
        out = local.reshape(n_init, ensemble_size, *local.shape[1:])
        return out if out_device is None else out.to(out_device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = work_bounds(n_init * ensemble_size, world)
    mine = b[rank + 1] - b[rank]
    if (0 if local is None else local.shape[0]) != mine:
        raise ValueError("local block does not hold this rank's work items")
    if device is None:
        device = local.device if local is not None else torch.device("cpu")
    bad_local = False
    if item_shape is not None:
        # a rank whose items do not have the announced shape must not raise HERE: the ranks without work (or with the right shape) would
        # already be inside the all_gather and wait for it until the collective times out.  It joins every collective with a zero block
        # and a status word behind the payload; every rank reads all status words and every rank raises the same error afterwards.
        item = tuple(int(v) for v in item_shape)
        bad_local = bool(mine) and tuple(local.shape[1:]) != item
    else:
        meta = torch.zeros(8, dtype=torch.int64, device=device)  # item ndim + dims (<= 7), agreed by MAX over ranks
        if mine:
            shp = tuple(local.shape[1:])
            meta[0] = len(shp)
            meta[1 : 1 + len(shp)] = torch.tensor(shp, dtype=torch.int64)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
        meta = meta.tolist()
        item = tuple(int(v) for v in meta[1 : 1 + int(meta[0])])
    cmax = max(b[r + 1] - b[r] for r in range(world))
    if out_device is None:
        out_device = device
    n_el = 1
    for v in item:
        n_el *= v
    rows = max(1, int(max_bytes // max(4 * n_el, 1)))  # rows of the padded block per collective
    out = torch.empty((n_init * ensemble_size,) + item, dtype=torch.float32, device=out_device)
    bad_ranks: List[int] = []
    for r0 in range(0, cmax, rows):
        n = min(rows, cmax - r0)
        x = torch.zeros(n * n_el + 1, dtype=torch.float32, device=device)  # [n padded items | status word: 1 = "my items have another shape"]
        have = min(max(mine - r0, 0), n)
        if bad_local:
            x[-1] = 1.0
        elif have:
            x[: have * n_el] = local[r0 : r0 + have].to(device).reshape(-1)
        bufs = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(bufs, x, group=group)
        if r0 == 0:
            bad_ranks = [r for r, v in enumerate(torch.stack([bf[-1] for bf in bufs]).tolist()) if v != 0.0]  # one read-back
        if bad_ranks:
            continue  # stay in step with the other ranks' collectives, copy nothing
        for r in range(world):
            cnt = min(max(b[r + 1] - b[r] - r0, 0), n)
            if cnt:
                out[b[r] + r0 : b[r] + r0 + cnt] = bufs[r][: cnt * n_el].reshape((cnt,) + item).to(out_device)
    if bad_ranks:
        mine_txt = f" (here: {tuple(local.shape[1:])})" if bad_local else ""
        raise ValueError(f"items on rank(s) {bad_ranks} do not have the announced item shape {item}{mine_txt}")
    return out.reshape(n_init, ensemble_size, *item)


def latent_item_shape(kwargs) -> Optional[Tuple[int, ...]]:
    """``(C, 1 + steps, h, w)`` of one (initial time, member) item of a LATENT-mode rollout, from the call's own arguments (no
    encoding, no field read): channels from the AR model's config, the latent grid from ``known_latents_override`` or the
    ``latent_hw`` hint.  None when it cannot be known up front (decoded mode; an encoder-fed rollout without the hint)."""
    if not kwargs.get("return_latent"):
        return None
    hw = None
    if kwargs.get("known_latents_override") is not None:
        hw = tuple(kwargs["known_latents_override"].shape[-2:])
    elif kwargs.get("latent_hw") is not None:
        hw = tuple(int(v) for v in kwargs["latent_hw"])
    pipe = kwargs.get("pipeline")
    if hw is None or pipe is None:
        return None
    cfg = pipe.ar_model.config
    total = int(kwargs.get("total_lead_time_hour", 240) / kwargs.get("step_size_hour", 6))
    return (int(getattr(cfg, "out_channels", None) or cfg.in_channels), total + 1) + hw


def roll_out_sharded(roll_out_fn, ensemble_size: int, pred_timestamp: Sequence, group=None, device: Optional[torch.device] = None,
                     out_device=None, timings: Optional[dict] = None, **kwargs) -> torch.Tensor:
    """Run ``roll_out_fn`` (``roll_out_serial``-compatible) on this rank's (initial time, member) items and gather.

    One call per initial time this rank touches, with ``ensemble_size = len(ids)`` and ``member_ids = ids``; its output is
    ``(1, len(ids), C, 1+steps, h, w)``.  Returns ``(n_init, ensemble_size, C, 1+steps, h, w)`` on every rank, identical to the
    single-process output.  A rank without items launches nothing.  ``device``: where the blocks live for the collective
    (this rank's GPU with the nccl = RCCL backend; the host with gloo).  In latent mode the item shape is known on every rank
    (``latent_item_shape``; pass ``latent_hw=(h, w)`` when the IC is encoded rather than given) and the exchange is a single
    all_gather.  ``return_ensemble_mean`` (pipelines/utils.py:296-300, decoded mode only): every member is computed where it lives,
    gathered, and averaged over the member axis afterwards - ``(n_init, 1, C, 1+steps, H, W)`` as the single-process call returns
    (same fp32 mean of the same members; slot 0, the IC field, is the same for every member and is copied, not averaged).
    ``ic_noise_seed``: with ``noise_level > 0`` the IC perturbation of an initial time is drawn from a generator seeded with
    ``ic_noise_seed + YYYYMMDDHH``; the default here is 0 (NOT the process RNG stream as in ``roll_out_serial``), so that every rank
    / piece perturbs an initial time identically - two forecasts of the same initial time therefore share their IC perturbation
    unless the caller varies the seed.  ``timings`` (optional dict): filled with ``rollout_s`` / ``gather_s`` host seconds of this
    call (a device synchronise separates the two phases when given - diagnostics, not for timed production loops)."""
    import time

    want_mean = bool(kwargs.pop("return_ensemble_mean", False))
    if want_mean and kwargs.get("return_latent"):
        raise ValueError("return_ensemble_mean must be False when return_latent is True.")  # pipelines/utils.py:296-300
    if kwargs.get("noise_level") and kwargs.get("ic_noise_seed") is None:
        kwargs["ic_noise_seed"] = 0  # the IC perturbation of an initial time must not depend on which rank / piece draws it
    item = latent_item_shape(kwargs)
    kwargs.pop("latent_hw", None)
    rank = dist.get_rank(group) if _active(group) else 0
    world = dist.get_world_size(group) if _active(group) else 1
    n_init = len(pred_timestamp)
    # the per-rank block stays where the collective wants it: with a device given (nccl: this rank's GPU) the rollout returns its
    # result there, un-synchronised, instead of a host copy that would be moved straight back
    on_dev = device is not None and torch.device(device).type == "cuda" and out_device is None and kwargs.get("output_device") is None
    if on_dev:
        kwargs["output_device"] = device
    t0 = time.perf_counter()
    blocks = []
    for t, ids in shard_work(n_init, ensemble_size, rank, world):
        o = roll_out_fn(ensemble_size=len(ids), member_ids=ids, pred_timestamp=[pred_timestamp[t]], **kwargs)
        blocks.append(o[0])
    local = torch.cat(blocks, dim=0) if blocks else None
    if local is not None and device is not None and out_device is None:
        local = local.to(device)  # (with a separate out_device the block is moved piece by piece inside gather_work)
    if timings is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        timings["rollout_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
    full = gather_work(local, n_init, ensemble_size, group=group, device=device, out_device=out_device, item_shape=item)
    if timings is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        timings["gather_s"] = time.perf_counter() - t0
    if want_mean:
        mean = full.mean(dim=1, keepdim=True)
        mean[:, 0, :, 0] = full[:, 0, :, 0]  # slot 0 = the IC field, identical for every member (written once by the reference, :462-468)
        return mean
    return full
