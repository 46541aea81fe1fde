"""``randn_tensor`` with diffusers semantics (pipelines/edm_sampler.py:53-55,
pipelines/pipeline_AR.py:77-82): the draw happens on the GENERATOR's device (the reference
always passes CPU generators, pipelines/utils.py:703-706) and is then uploaded; a list of
generators draws one ``(1, ...)`` sample per generator.  This is torch's CPU Philox/MT stream by
definition of the reference's seeding contract -- it is input generation, not the hot path."""
from __future__ import annotations

import torch

from .. import hip


def randn_tensor(shape, generator=None, device=None, dtype=None):
    with few_host_threads():
        return _randn_tensor(shape, generator, device, dtype)


_upload = hip.upload_nonblocking


def _randn_tensor(shape, generator=None, device=None, dtype=None):
    device = torch.device(device) if device is not None else torch.device("cpu")
    batch = shape[0]
    pin = device.type == "cuda"
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        if all(g.device.type == "cpu" for g in generator):  # one (1, ...) draw per generator, written into one (pinned) host buffer
            buf = torch.empty(tuple(shape), dtype=dtype, pin_memory=pin)
            for i in range(batch):
                torch.randn((1,) + tuple(shape[1:]), generator=generator[i], dtype=dtype, out=buf[i : i + 1])
            return _upload(buf, device)
        one = (1,) + tuple(shape[1:])
        parts = [torch.randn(one, generator=generator[i], device=generator[i].device, dtype=dtype) for i in range(batch)]
        return torch.cat(parts, dim=0).to(device)
    rand_device = generator.device if generator is not None else device
    if rand_device.type == "cpu":
        return _upload(torch.randn(tuple(shape), generator=generator, dtype=dtype, pin_memory=pin), device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)


class few_host_threads:
    """Context manager for the host-side glue of the rollout (noise draws, small CPU tensor writes): run it on at most `limit` intra-op
    threads.  Those tensors hold ~10^5 elements; torch would still wake its whole OpenMP pool (128 threads on a 2 x 64-core host) for each of
    them, and on a busy host - or with one process per GPU, 8 pools on one box - that wake-up costs 50-100 ms at random (measured: chunk
    times of 144 / 245 / 144 / 203 ms with the default pool, 144.5 +- 0.5 ms with 4 threads).  Values do not depend on the thread count."""

    def __init__(self, limit: int = 4):
        self.limit = limit
        self.saved = None

    def __enter__(self):
        self.saved = torch.get_num_threads()
        if self.saved > self.limit:
            torch.set_num_threads(self.limit)
        return self

    def __exit__(self, *exc):
        if self.saved is not None and torch.get_num_threads() != self.saved:
            torch.set_num_threads(self.saved)
        return False
