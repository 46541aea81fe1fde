"""``randn_tensor`` with diffusers semantics (pipelines/edm_sampler.py:53-55,
pipelines/pipeline_AR.py:77-82): the draw happens on the GENERATOR's device (the reference
always passes CPU generators, pipelines/utils.py:703-706) and is then uploaded; a list of
generators draws one ``(1, ...)`` sample per generator.  This is torch's CPU Philox/MT stream by
definition of the reference's seeding contract -- it is input generation, not the hot path."""
from __future__ import annotations

import torch


def randn_tensor(shape, generator=None, device=None, dtype=None):
    device = torch.device(device) if device is not None else torch.device("cpu")
    batch = shape[0]
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        one = (1,) + tuple(shape[1:])
        parts = [torch.randn(one, generator=generator[i], device=generator[i].device, dtype=dtype) for i in range(batch)]
        return torch.cat(parts, dim=0).to(device)
    rand_device = generator.device if generator is not None else device
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)
