"""``edm_AR_sampler`` -- EDM Heun sampler (pipelines/edm_sampler.py:10-120; deterministic, and the stochastic-churn branch) with
the fp64 state kept on the device and every update fused into one HIP kernel per half-step.

Per solver step the reference issues ~15 separate fp64 torch ops (clone, scale, cast, c_skip /
c_out axpy, subtract, divide, axpy ...) around each model call; here a step is
``ldc_edm_scale_f64_to_f32 -> net -> ldc_edm_euler`` and, for the 2nd-order correction,
``ldc_edm_scale_f64_to_f32 -> net -> ldc_edm_heun``.  The scalar coefficients are computed on
the host in fp32 with the scheduler's own expressions and widened to double, so given the
same network output the state is bit-identical to the reference's.
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import torch

from .. import hip
from .torch_utils import randn_tensor


@torch.no_grad()
def edm_AR_sampler(
    net,
    noise_scheduler,
    batch_size=1,
    return_seq_len=1,
    randn_like=torch.randn_like,
    num_inference_steps=18,
    S_churn=0,
    S_min=0,
    S_max=float("inf"),
    S_noise=0,
    deterministic=True,
    known_latents=None,
    timestamps: Optional[torch.LongTensor] = None,
    generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
    device="cpu",
):
    if isinstance(generator, list) and len(generator) != batch_size:
        raise ValueError(
            f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
            f" size of {batch_size}. Make sure the batch size matches the length of the generators."
        )
    assert known_latents is not None, "known_latents must be provided"
    if isinstance(device, str):
        device = torch.device(device)

    shape = (batch_size, net.config.out_channels, return_seq_len, *known_latents.shape[-2:])
    latents = randn_tensor(shape, generator=generator, device=device, dtype=torch.float32).contiguous()  # fp32 whatever net.dtype says: see pipeline_AR.py
    noise_scheduler.set_timesteps(num_inference_steps, device=device)
    t_steps = noise_scheduler.sigmas  # (N+1,) fp32 on the host
    # (N,) fed to the model as (1,) views; a captured chunk holds its own device copy, so the upload happens on the paths that need it
    c_noise_host = noise_scheduler.precondition_noise(t_steps[:-1])
    known = known_latents.to(device=device, dtype=torch.float32)  # the model's kernels read fp32 (its forward() casts; the launch-only entry does not)
    if known.shape[0] != batch_size:
        known = known.expand(batch_size, *known.shape[1:])
    known = known.contiguous()

    # hipGraph of the WHOLE chunk (every forward and every state update of the 2N-1 evaluations): all scalars of the
    # solver are known on the host before the first launch, so one replay per chunk replaces ~3000 launches and the
    # copies around 39 per-forward graphs (2.5 % of the chunk), and the host's speed stops mattering (8 ranks on one box).
    # Used when the model runs in hipGraph mode and exposes its launch-only forward; same kernels, same numbers.
    chunk_graph = bool(getattr(net, "use_hip_graph", False)) and hasattr(net, "forward_launch_only") and hasattr(net, "_graphs")
    # The part of the network that does not see the sample (context refiner, conditioning embedding, AdaLN modulation vectors) depends
    # on (noise level, conditioning, timestamp) only: a model that offers `prepare_conditioning` evaluates it for the chunk's N noise
    # levels as ONE batch at the start of the chunk, every chunk; each of the 2N - 1 network evaluations then runs the sample-dependent
    # part only (same arithmetic per entry; the Heun correction at t_next and the next Euler step at t_cur = t_next share an entry).
    batched = bool(getattr(net, "batch_conditioning", False)) and hasattr(net, "prepare_conditioning")
    if not deterministic:
        # stochastic churn (:67-76; never enabled by the reference's own rollout driver, pipelines/utils.py:716-727): the noise comes
        # from the caller's `randn_like` at every step, so the chunk is launched eagerly - nothing to capture once and replay
        out = torch.empty(shape, device=device, dtype=torch.float32)
        fwd = lambda x, t, k, te_: net(x, t, k, time_elapsed=timestamps).sample  # noqa: E731
        _heun_chunk_stochastic(fwd, noise_scheduler, t_steps, latents, known, out, shape, device, num_inference_steps, randn_like, S_churn, S_min,
                               S_max, S_noise)
        return out
    if chunk_graph:
        te = net.time_elapsed_embedding(timestamps)  # eager, cached per chunk; the graph reads its persistent buffer
        plan_id = net.plan_identity()  # a re-packed / re-loaded model gets new graphs
        key = (tuple(shape), tuple(known.shape), num_inference_steps, tuple(float(v) for v in t_steps.tolist()),
               None if te is None else (te.data_ptr(), tuple(te.shape)), str(device), plan_id, batched)
        prepare = net.prepare_conditioning if batched else None
        key = ("edm_chunk",) + key
        cache = net._graphs  # the model's graph store: dropped with the packed weights (load_state_dict, .to(), precision switch)
        # TWO instances of the captured chunk, used alternately: launching an executable graph that is still running blocks the host
        # until it has finished, and the GPU then idles through the ~0.4 ms the launch call takes to build its packets; the other
        # instance is queued behind the running one instead (same stream: they never overlap on the GPU, and share the workspaces)
        # Both are captured at the first use of a chunk shape: whoever warms up once has warmed up.
        turn = cache.get(key + ("turn",), 0)
        cache[key + ("turn",)] = turn ^ 1

        def capture():
            st_lat, st_known, st_out = torch.empty_like(latents), torch.empty_like(known), torch.empty(shape, device=device, dtype=torch.float32)
            st_lat.copy_(latents)
            st_known.copy_(known)
            cn = c_noise_host.to(device)
            side = net.capture_stream() if hasattr(net, "capture_stream") else torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):  # warm-up on the capture stream: per-stream workspaces are created here
                _heun_chunk(net.forward_launch_only, noise_scheduler, t_steps, cn, st_lat, st_known, te, st_out, shape, device, num_inference_steps, prepare)
            torch.cuda.synchronize()
            hip.rearm_attention_workspaces(device)  # ticket counters of the balanced fp32 attention: zeroed before a (re)capture, with nothing in flight (hip.py)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):  # other threads (the RCCL watchdog) may touch the runtime
                _heun_chunk(net.forward_launch_only, noise_scheduler, t_steps, cn, st_lat, st_known, te, st_out, shape, device, num_inference_steps, prepare)
            return (graph, st_lat, st_known, st_out, cn, side)

        for inst in (turn, turn ^ 1):
            if cache.get(key + (inst,)) is None:
                cache[key + (inst,)] = capture()
        ent = cache[key + (turn,)]
        graph, st_lat, st_known, st_out = ent[:4]
        st_lat.copy_(latents)
        st_known.copy_(known)
        graph.replay()
        return st_out.clone()

    out = torch.empty(shape, device=device, dtype=torch.float32)
    c_noise = hip.upload_nonblocking(c_noise_host, device)
    fwd = lambda x, t, k, te_, ci=None: net(x, t, k, time_elapsed=timestamps, **({} if ci is None else {"conditioning": ci})).sample  # noqa: E731
    prepare = None
    if batched:
        te = net.time_elapsed_embedding(timestamps)
        prepare = lambda cn, kn, te_: net.prepare_conditioning(cn, kn, te)  # noqa: E731
    _heun_chunk(fwd, noise_scheduler, t_steps, c_noise, latents, known, None, out, shape, device, num_inference_steps, prepare)
    return out


def _heun_chunk(fwd, noise_scheduler, t_steps, c_noise, latents, known, te, out, shape, device, num_inference_steps, prepare=None):
    """the 2N-1 evaluations of pipelines/edm_sampler.py:60-113 as launches only: fwd(x, timestep, known, te, conditioning) -> F;
    prepare(c_noise, known, te) -> the model's conditioning pack for the N noise levels (or None: every evaluation computes its own)"""
    cond = prepare(c_noise, known, te) if prepare is not None else None
    ci = (lambda i: None) if cond is None else (lambda i: (cond, i))  # noqa: E731
    x_hat = torch.empty(shape, device=device, dtype=torch.float64)
    x_next = torch.empty_like(x_hat)
    d_cur = torch.empty_like(x_hat)
    x_in = torch.empty(shape, device=device, dtype=torch.float32)
    hip.edm_init_state(latents, float(t_steps[0]), x_next)

    for i in range(num_inference_steps):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_hat, x_next = x_next, x_hat  # x_hat = previous x_next
        c_skip, c_out = noise_scheduler._c_skip_out(t_cur)
        hip.edm_scale_f64_to_f32(x_hat, float(noise_scheduler._c_in(t_cur)), x_in)
        F = fwd(x_in, c_noise[i : i + 1], known, te, ci(i))
        hip.edm_euler(x_hat, F, float(c_skip), float(c_out), float(t_cur), float(t_next - t_cur), x_next, d_cur)
        if i < num_inference_steps - 1:
            c_skip, c_out = noise_scheduler._c_skip_out(t_next)
            hip.edm_scale_f64_to_f32(x_next, float(noise_scheduler._c_in(t_next)), x_in)
            F = fwd(x_in, c_noise[i + 1 : i + 2], known, te, ci(i + 1))
            hip.edm_heun(x_hat, x_next, F, d_cur, float(c_skip), float(c_out), float(t_next), float(t_next - t_cur))
    hip.f64_to_f32(x_next, out)


def _heun_chunk_stochastic(fwd, noise_scheduler, t_steps, latents, known, out, shape, device, num_inference_steps, randn_like, S_churn, S_min,
                           S_max, S_noise):
    """pipelines/edm_sampler.py:60-113 with `deterministic=False`: before every step the state is pushed back up to the noise level
    t_hat = t_cur + gamma t_cur with fresh noise from `randn_like` (called with the fp64 device state, every step - also when
    gamma = 0 - so the caller's random stream advances as in the reference), and the step runs from t_hat.  All scalars are
    formed on the host with the reference's own expressions (fp32 0-dim tensor arithmetic) and widened to double."""
    x_cur = torch.empty(shape, device=device, dtype=torch.float64)
    x_hat = torch.empty_like(x_cur)
    x_next = torch.empty_like(x_cur)
    d_cur = torch.empty_like(x_cur)
    x_in = torch.empty(shape, device=device, dtype=torch.float32)
    hip.edm_init_state(latents, float(t_steps[0]), x_next)
    for i in range(num_inference_steps):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_cur, x_next = x_next, x_cur
        gamma = min(S_churn / num_inference_steps, np.sqrt(2) - 1) if S_min <= t_cur <= S_max else 0
        t_hat = torch.as_tensor(t_cur + gamma * t_cur)
        noise = randn_like(x_cur)
        if noise.dtype != torch.float64 or noise.device != x_cur.device:
            noise = noise.to(device=x_cur.device, dtype=torch.float64)
        hip.edm_churn(x_cur, noise.contiguous(), float((t_hat**2 - t_cur**2).sqrt() * S_noise), x_hat)
        c_skip, c_out = noise_scheduler._c_skip_out(t_hat)
        hip.edm_scale_f64_to_f32(x_hat, float(noise_scheduler._c_in(t_hat)), x_in)
        F = fwd(x_in, noise_scheduler.precondition_noise(t_hat.reshape(1)).to(device), known, None)
        hip.edm_euler(x_hat, F, float(c_skip), float(c_out), float(t_hat), float(t_next - t_hat), x_next, d_cur)
        if i < num_inference_steps - 1:
            c_skip, c_out = noise_scheduler._c_skip_out(t_next)
            hip.edm_scale_f64_to_f32(x_next, float(noise_scheduler._c_in(t_next)), x_in)
            F = fwd(x_in, noise_scheduler.precondition_noise(t_next.reshape(1)).to(device), known, None)
            hip.edm_heun(x_hat, x_next, F, d_cur, float(c_skip), float(c_out), float(t_next), float(t_next - t_hat))
    hip.f64_to_f32(x_next, out)
