import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- session-wide oracle runs shared by the full-size GPU tests (the CPU oracle is most of the GPU suite's wall time) ----
@pytest.fixture(scope="session")
def full_dcae_oracle():
    """BASELINE configs[0] on the oracle, once per session: the full-size DCAE, one synthetic 84 x 120 x 240 frame + static
    fields, its latent and the decoded frame.  Returns a namespace (model, f, st, z, y)."""
    from types import SimpleNamespace

    import torch

    from oracle.dcae import CONFIG_DCAE_84
    from tests.synth import make_dcae, synth_field

    o = make_dcae(dict(CONFIG_DCAE_84))
    f, st = synth_field(1, 84, 120, 240), synth_field(1, 5, 120, 240, seed=1)
    gold = os.path.join(ROOT, "tests", "golden", "fullsize_dcae.npz")
    if os.path.exists(gold):  # the oracle's outputs as committed by tests/golden/make_fullsize_golden.py (y: every 7th value, tests/synth.py Sub)
        from tests.synth import load_fullsize_golden

        g = load_fullsize_golden(gold)
        z, y = g["z"], g["y"]
    else:
        with torch.no_grad():
            z = o.encode(f, static_conditioning_tensor=st).latent
            y = o.decode(z).sample
    return SimpleNamespace(model=o, cfg=dict(CONFIG_DCAE_84), f=f, st=st, z=z, y=y)


@pytest.fixture(scope="session")
def oracle_375m():
    """the oracle's 375M transformer with the seeded synthetic weights (1.5 GB), built once per session"""
    from oracle.ar_model import CONFIG_375M
    from tests.synth import make_ar

    return make_ar(dict(CONFIG_375M))


class RecordingNet:
    """records the input state and the output of every network evaluation of a sampler run (eager launches only: it hides
    the model's hipGraph surface on purpose)"""

    _HIDE = ("use_hip_graph", "forward_launch_only", "_graphs", "capture_stream")

    def __init__(self, net):
        self._net, self.ins, self.outs = net, [], []

    def __getattr__(self, k):
        if k in RecordingNet._HIDE:
            raise AttributeError(k)
        return getattr(self._net, k)

    def __call__(self, x, *a, **kw):
        o = self._net(x, *a, **kw)
        y = o[0] if isinstance(o, tuple) else o.sample
        self.ins.append(x.detach().float().cpu().clone())
        self.outs.append(y.detach().float().cpu().clone())
        return o


@pytest.fixture(scope="session")
def fullsize_chunk_oracle(full_dcae_oracle, oracle_375m):
    """BASELINE configs[1] (and the single-GPU share of configs[4]) on the oracle, once per session: the synthetic frame's latent,
    normalised with its own per-channel statistics (target std 0.5), is the known latent of ONE 20-step Heun chunk of the 375M
    model (39 recorded forwards, 1 member, R = 4); the 4 predicted latent frames are de-normalised and decoded by the oracle DCAE.
    Namespace: ar (oracle model), targs, known (1, 84, 1, 15, 30), ts, want (1, 84, 4, 15, 30) normalised samples, ins / outs per
    evaluation, decoded (1, 84, 4, 120, 240), seconds."""
    import time
    from types import SimpleNamespace

    import torch

    from oracle import pipelines as OP
    from oracle.ar_model import CONFIG_375M
    from oracle.scheduler import EDMDPMSolverMultistepScheduler as OracleScheduler
    d = full_dcae_oracle
    gold = os.path.join(ROOT, "tests", "golden", "fullsize_375m_chunk.npz")
    if os.path.exists(gold):
        # the oracle's side of the chunk as committed by tests/golden/make_fullsize_golden.py (made in the build container from the same
        # seeds): per-evaluation tensors as every 32nd value, the decoded frames as every 16th (tests/synth.py Sub); ~4 minutes of CPU
        # oracle per GPU session less, and the 20-forward `pipeline` chunk comes with it
        from tests.synth import load_fullsize_golden

        g = load_fullsize_golden(gold)
        targs = {"mean": g["mean"].tolist(), "std": g["std"].tolist(), "target_std": 0.5}
        return SimpleNamespace(ar=oracle_375m, cfg=dict(CONFIG_375M), targs=targs, known=g["known"], ts=torch.tensor([2018010100]), want=g["want_edm"],
                               ins=g["ins_edm"], outs=g["outs_edm"], latents=g["latents"], decoded=g["decoded"], in20=g["in20"], fwd_in20=g["fwd_in20_t0p3"],
                               seconds=float(g["seconds_edm"][0]), pipeline=SimpleNamespace(want=g["want_pipeline"], ins=g["ins_pipeline"], outs=g["outs_pipeline"],
                                                                                             seconds=float(g["seconds_pipeline"][0])))
    z = d.z[0]  # (84, 15, 30)
    mu, sd = z.mean(dim=(1, 2)), z.std(dim=(1, 2))
    targs = {"mean": mu.tolist(), "std": sd.tolist(), "target_std": 0.5}
    known = OP.get_transform_3D("normalize", targs)(z[:, None].clone())[None]  # (1, 84, 1, 15, 30)
    ts = torch.tensor([2018010100])
    ar = oracle_375m
    rec = RecordingNet(ar)
    t0 = time.perf_counter()
    want = OP.ensemble_AR_sampler(OP.AutoRegressive2DPipeline(rec, OracleScheduler()), 1, 4, 20, known_latents=known, timestamps=ts, sampler_type="edm")
    lat = OP.get_inv_transform_3D("normalize", targs)(want.permute(1, 0, 2, 3, 4).reshape(84, 4, 15, 30)).reshape(84, 1, 4, 15, 30).permute(1, 0, 2, 3, 4)
    decoded = OP.decode_latent_ens(d.model, lat)
    with torch.no_grad():
        fwd = ar(rec.ins[20], torch.tensor([0.3]), known, time_elapsed=ts).sample
    return SimpleNamespace(ar=ar, cfg=dict(CONFIG_375M), targs=targs, known=known, ts=ts, want=want, ins=rec.ins, outs=rec.outs, latents=lat,
                           decoded=decoded, in20=rec.ins[20], fwd_in20=fwd, seconds=time.perf_counter() - t0, pipeline=None)


@pytest.fixture(autouse=True)
def _few_threads_unless_full_size(request):
    """Everything but the full-size parity tests runs its CPU oracle on 16 threads: the ops of the tiny-width models are small, and on a
    busy 128-core host waking the whole OpenMP pool for each of them costs 50-100 ms at random (a 20-step chunk of the tiny model: 1.8 s
    on 16 threads, 15 s with the default pool, 73 s on one noisy box).  The full-size oracle runs (375M, 1.6B, the full DCAE, the
    1.09 G-parameter autoencoder) keep every core.  Values do not depend on the thread count beyond ~1e-7."""
    name = request.node.name.lower()
    full = any(k in name for k in ("full", "375m", "1_6b", "1p6b", "ray_1024")) or any(
        f in request.fixturenames for f in ("oracle_375m", "full_dcae_oracle", "fullsize_chunk_oracle"))
    if full:
        yield
        return
    import torch

    old = torch.get_num_threads()
    torch.set_num_threads(min(16, old))
    try:
        yield
    finally:
        torch.set_num_threads(old)
