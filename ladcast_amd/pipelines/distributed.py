"""Multi-GPU sharding of the rollout: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The path shards over ensemble MEMBERS (and initial times), never over lead time: members are
independent given the shared IC latent, member k is always seeded with k
(pipelines/utils.py:703-706), and lead-time chunks of one member are a sequential chain
(:563).  So there is no data-path collective; the only exchange is the reference's own
``accelerator.gather(result_tensor)`` at the end (evaluate/pred_rollout.py:398-400), done here
as ONE all_gather of the per-rank result block.  On an 8-GPU xGMI mesh that is a 12 MB-per-rank
message over direct links -- never on the critical path, no ring/tree tuning needed.

The helpers are device-agnostic (they only move tensors), so the N>1 logic is exercised on CPU
with the gloo backend in ``tests/test_distributed_cpu.py``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_members(ensemble_size: int, rank: int, world_size: int) -> List[int]:
    """Round-robin member ownership: rank r computes members {k : k mod world_size == r}."""
    return list(range(rank, ensemble_size, world_size))


def members_per_rank(ensemble_size: int, world_size: int) -> List[int]:
    return [len(range(r, ensemble_size, world_size)) for r in range(world_size)]


def gather_members(local: torch.Tensor, ensemble_size: int, member_dim: int = 0, group=None) -> torch.Tensor:
    """All-gather the per-rank member blocks and restore global member order.

    ``local``: this rank's members (in ``shard_members`` order) along ``member_dim``.  Ranks may own
    different member counts (ensemble_size not divisible by world size): blocks are padded to the
    largest count for the collective and trimmed afterwards.  Returns the full ensemble on every rank.
    """
    if not (dist.is_available() and dist.is_initialized()):
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


def roll_out_sharded(roll_out_fn, ensemble_size: int, group=None, device: Optional[torch.device] = None, **kwargs) -> torch.Tensor:
    """Run ``roll_out_fn`` (``roll_out_serial``-compatible) on this rank's members and gather.

    The rollout output is ``(n_init, ens_local, C, 1+steps, h, w)``; the gathered result is
    ``(n_init, ensemble_size, ...)`` on every rank, identical to the single-process output.
    """
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ids = shard_members(ensemble_size, rank, world)
    out = roll_out_fn(ensemble_size=len(ids), member_ids=ids, **kwargs)
    if device is not None:
        out = out.to(device)
    full = gather_members(out, ensemble_size, member_dim=1, group=group)
    return full
