"""ladcast_amd -- MI355X-native (gfx950) implementation of LaDCast's autoregressive
latent-diffusion rollout behind the reference's own pipeline / model / scheduler API.

Layout mirrors the reference package for the path it replaces:
``ladcast_amd.pipelines`` <-> ``ladcast.pipelines``, ``ladcast_amd.models`` <->
``ladcast.models``, ``ladcast_amd.schedulers`` <-> ``diffusers.schedulers``.  All tensor
arithmetic runs in hand-written HIP kernels from ``libladcast_hip.so`` (``csrc/``) reached
through the C ABI in ``include/ladcast_hip.h``; there is no CPU / PyTorch fallback.
"""
__version__ = "0.1.0"
