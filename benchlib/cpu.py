"""`cpu_baseline` legs: the CPU oracle (kind "port": the reference's own path needs diffusers, absent here and on the GPU box) timed on a
bounded sample of the same workload on the box's host cores.  The ONLY place of the benchmark that touches oracle/."""
import statistics
import time

import torch

from .configs import CONFIG_DCAE_84, CONFIGS
from .host import host_description


def ar_forward_seconds(cfg_name, R, n_forwards):
    """`n_forwards` forwards (after one warm-up forward) of the oracle AR transformer, 1 member, return_seq_len R"""
    from oracle.ar_model import LaDCastTransformer3DModel as OracleModel

    torch.manual_seed(1234)
    m = OracleModel.from_config(CONFIGS[cfg_name]).eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 84, R, 15, 30, generator=g)
    known = 0.5 * torch.randn(1, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))
    ts = torch.tensor([2018010100])
    times = []
    with torch.no_grad():
        m(x, torch.tensor([0.5]), known, time_elapsed=ts)  # warm-up (page-in, thread pool)
        for i in range(n_forwards):
            t0 = time.perf_counter()
            m(x, torch.tensor([0.5 - 0.05 * i]), known, time_elapsed=ts)
            times.append(time.perf_counter() - t0)
    return times


def ar_baseline(cfg_name, R, n_forwards, forwards_per_step, lead_per_step):
    """cfg 2's cpu_baseline: median forward time scaled to the forwards one step (a sampler chunk) needs"""
    cores = torch.get_num_threads()
    times = ar_forward_seconds(cfg_name, R, n_forwards)
    dt = statistics.median(times)
    return dict(value=round(lead_per_step / (forwards_per_step * dt), 5), unit="member-steps/s", cores=cores, kind="port", host=host_description(),
                forward_seconds=[round(t, 3) for t in times],
                sample=f"{n_forwards} forwards (after one warm-up forward) of the same {cfg_name} model (1 member, R={R}) by the PyTorch CPU oracle on {cores} torch "
                       f"threads, median {dt:.2f} s per forward, scaled to {forwards_per_step} forwards per step")


def dcae_baseline(n=3):
    """cfg 1's cpu_baseline (BASELINE configs[0]): one 84 x 120 x 240 frame (+ 5 static channels) through the oracle's
    DC-AE encode and decode, fp32, median of `n` after one warm-up pass"""
    from oracle.dcae import AutoencoderDC as OracleAE

    torch.manual_seed(1234)
    o = OracleAE.from_config(CONFIG_DCAE_84).eval()
    g = torch.Generator().manual_seed(3)
    x, st = torch.randn(1, 84, 120, 240, generator=g), torch.randn(1, 5, 120, 240, generator=g)
    enc, dec = [], []
    with torch.no_grad():
        z = o.encode(x, static_conditioning_tensor=st).latent
        o.decode(z)
        for _ in range(n):
            t0 = time.perf_counter()
            z = o.encode(x, static_conditioning_tensor=st).latent
            t1 = time.perf_counter()
            o.decode(z)
            t2 = time.perf_counter()
            enc.append(t1 - t0)
            dec.append(t2 - t1)
    te, td, cores = statistics.median(enc), statistics.median(dec), torch.get_num_threads()
    return dict(value=round(1.0 / (te + td), 4), unit="frames/s (encode + decode)", encode_ms=round(1e3 * te, 1), decode_ms=round(1e3 * td, 1), cores=cores, kind="port",
                host=host_description(), encode_seconds=[round(t, 3) for t in enc], decode_seconds=[round(t, 3) for t in dec],
                sample=f"{n} encode + decode passes (after one warm-up pass) of one 84 x 120 x 240 frame by the PyTorch CPU oracle's DC_AE_84 on {cores} torch threads, median")
