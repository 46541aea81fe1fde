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
        # the initial noise is drawn in fp32 whatever dtype the caller cast the model to (pipeline_AR.py:77-82 asks for ar_model.dtype): the
        # kernels read fp32 samples, the parameters of a bf16 / fp16 model are up-cast when its plan is built (`_upcast_to_fp32`), and an fp32
        # draw is the stream the fp32 run of the same seeds sees
        image = randn_tensor(shape, generator=generator, device=dev, dtype=torch.float32).contiguous()
        known_latents = known_latents.to(device=dev, dtype=torch.float32)  # the reference discards this result (Q13); we keep it (fp32: what the kernels read)
        if not do_edm_style:
            raise NotImplementedError("Only EDM style is supported for now")
        net, sch = self.ar_model, self.scheduler
        # a scheduler whose `step` is host scalars + kernel launches (`launch_only`) lets the WHOLE loop be captured; step kwargs that make it
        # draw noise (DDIM's eta > 0, any generator) put the loop back on eager launches
        kw_ = self.scheduler_step_kwargs
        graphable = getattr(sch, "launch_only", False) and hasattr(sch, "graph_signature") and not kw_.get("eta") and kw_.get("generator") is None and kw_.get("variance_noise") is None
        if getattr(net, "use_hip_graph", False) and hasattr(net, "forward_launch_only") and hasattr(net, "_graphs") and graphable:
            image = self._graph_loop(net, sch, image, known_latents.contiguous(), timestamps, num_inference_steps, batch_size, dev)
            if not return_dict:
                return (image,)
            return Fields2DPipelineOutput(fields=image)
        self.scheduler.set_timesteps(num_inference_steps)
        # the sample-independent part of the network for all timesteps of the loop in one batch (ladcast_amd models only; see
        # LaDCastTransformer3DModel.prepare_conditioning and pipelines/edm_sampler.py)
        pack = None
        if getattr(net, "batch_conditioning", False) and hasattr(net, "prepare_conditioning"):
            kn = known_latents if known_latents.shape[0] == batch_size else known_latents.expand(batch_size, *known_latents.shape[1:])
            pack = net.prepare_conditioning(self.scheduler.timesteps, kn.contiguous(), net.time_elapsed_embedding(timestamps))
        for i, t in enumerate(self.scheduler.timesteps):
            if not do_edm_style:
                raise NotImplementedError("Only EDM style is supported for now")
            x_in = self.scheduler.scale_model_input(image, t)
            t_host = t
            t = t.expand(batch_size).to(dev)
            t.host_value = t_host  # schedulers that index by timestep (DDIM / DDPM) read it here instead of stalling on a device read-back
            ckw = {} if pack is None else {"conditioning": (pack, i)}
            model_output = self.ar_model(x_in, t, known_latents, time_elapsed=timestamps, return_dict=False, **ckw)[0]
            image = self.scheduler.step(model_output, t, image, **self.scheduler_step_kwargs, return_dict=False)[0]
        if not return_dict:
            return (image,)
        return Fields2DPipelineOutput(fields=image)

    def _graph_loop(self, net, sch, image, known, timestamps, num_inference_steps, batch_size, dev):
        """hipGraph of the whole scheduler loop (N forwards + N scheduler steps), as edm_AR_sampler does for its chunk: with this
        build's scheduler every coefficient is a host scalar known before the first launch, so the loop is launches only.  Same
        kernels and arguments as the eager loop (bit-identical samples); afterwards the scheduler object is left in the state
        the eager loop leaves it in (`graph_state` / `set_graph_state`: EDM - step index N and the last two x0 predictions; DDIM - none).
        The scheduler describes itself to the graph cache through `graph_signature()` (its config and schedule)."""
        sch.set_timesteps(num_inference_steps)
        te = net.time_elapsed_embedding(timestamps)
        key = (tuple(image.shape), tuple(known.shape), num_inference_steps, sch.graph_signature(),
               tuple(sorted((k, str(v)) for k, v in self.scheduler_step_kwargs.items())), None if te is None else (te.data_ptr(), tuple(te.shape)), str(dev), net.plan_identity(),
               bool(getattr(net, "batch_conditioning", False)))
        key = ("pipeline_loop",) + key
        cache = net._graphs  # the model's graph store: dropped with the packed weights (load_state_dict, .to(), precision switch)
        ent = cache.get(key)  # (one instance: the two-instance scheme of edm_sampler.py is not applied to this secondary loop)

        # device copy of the timesteps for `prepare_conditioning`, made outside the capture (a host-to-device copy) and kept alive with
        # the graph that reads it
        ts_dev = sch.timesteps.to(device=dev, dtype=torch.float32).contiguous() if ent is None else None

        def loop(img, kn, tsteps):
            kb = kn if kn.shape[0] == batch_size else kn.expand(batch_size, *kn.shape[1:]).contiguous()
            pack = net.prepare_conditioning(ts_dev, kb, te) if getattr(net, "batch_conditioning", False) else None
            for i, (t, t_dev) in enumerate(zip(sch.timesteps, tsteps)):
                x_in = sch.scale_model_input(img, t)
                out = net.forward_launch_only(x_in, t_dev, kb, te, None if pack is None else (pack, i))
                img = sch.step(out, t, img, **self.scheduler_step_kwargs, return_dict=False)[0]
            return img

        if ent is None:
            st_img, st_known = torch.empty_like(image), torch.empty_like(known)
            st_img.copy_(image)
            st_known.copy_(known)
            # what the eager loop feeds the model, in the fp32 its forward() casts to (DDIM / DDPM timesteps are int64; the launch-only entry reads raw fp32)
            tsteps = [t.to(device=dev, dtype=torch.float32).expand(batch_size).contiguous() for t in sch.timesteps]
            side = net.capture_stream() if hasattr(net, "capture_stream") else torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up on the capture stream (per-stream workspaces)
                loop(st_img, st_known, tsteps)
            torch.cuda.synchronize()
            sch.set_timesteps(num_inference_steps)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                st_out = loop(st_img, st_known, tsteps)
            ent = (graph, st_img, st_known, st_out, tsteps, side, sch.graph_state(), ts_dev)  # ts_dev: read by the graph
            cache[key] = ent
        graph, st_img, st_known, st_out = ent[:4]
        st_img.copy_(image)
        st_known.copy_(known)
        graph.replay()
        sch.set_graph_state(ent[6])
        return st_out.clone()
