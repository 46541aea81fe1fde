"""``AutoRegressive2DPipeline`` (pipelines/pipeline_AR.py:9-107): the diffusers-style
``scale_model_input -> model -> scheduler.step`` loop.  Keeps the reference's quirks: the
initial noise is NOT multiplied by ``init_noise_sigma`` (:77-82) and ``do_edm_style=False``
raises ``NotImplementedError`` (:97)."""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch

from .torch_utils import randn_tensor
from .utils import Fields2DPipelineOutput


class AutoRegressive2DPipeline:
    model_cpu_offload_seq = "unet"

    def __init__(self, ar_model, scheduler, scheduler_step_kwargs: Optional[dict] = None):
        self.register_modules(ar_model=ar_model, scheduler=scheduler)
        self.scheduler_step_kwargs = scheduler_step_kwargs or {}

    # the slice of diffusers.DiffusionPipeline the rollout driver touches
    def register_modules(self, **mods):
        for k, v in mods.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return self.ar_model.device

    @property
    def device(self):
        return self.ar_model.device

    def to(self, *a, **k):
        self.ar_model.to(*a, **k)
        return self

    def return_trajectory(self, *a, **k):
        raise NotImplementedError("This function is not implemented yet.")

    @torch.no_grad()
    def __call__(
        self,
        batch_size: int = 1,
        return_seq_len: int = 1,
        known_latents: torch.Tensor = None,
        timestamps: Optional[torch.LongTensor] = None,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        num_inference_steps: int = 50,
        return_dict: bool = True,
        do_edm_style: bool = True,
    ) -> Union[Fields2DPipelineOutput, Tuple]:
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators."
            )
        assert known_latents is not None, "known_latents must be provided"
        dev = self._execution_device
        shape = (batch_size, self.ar_model.config.out_channels, return_seq_len, *known_latents.shape[-2:])
        image = randn_tensor(shape, generator=generator, device=dev, dtype=self.ar_model.dtype).contiguous()
        known_latents = known_latents.to(dev)  # the reference discards this result (Q13); we keep it
        self.scheduler.set_timesteps(num_inference_steps)
        for t in self.scheduler.timesteps:
            if not do_edm_style:
                raise NotImplementedError("Only EDM style is supported for now")
            x_in = self.scheduler.scale_model_input(image, t)
            t = t.expand(batch_size).to(dev)
            model_output = self.ar_model(x_in, t, known_latents, time_elapsed=timestamps, return_dict=False)[0]
            image = self.scheduler.step(model_output, t, image, **self.scheduler_step_kwargs, return_dict=False)[0]
        if not return_dict:
            return (image,)
        return Fields2DPipelineOutput(fields=image)
