"""End-to-end parity of the HIP product path against the CPU oracle on identical weights and inputs.
Tolerance (BASELINE.json north_star): rel-L2 <= 1e-4 in fp32; scheduler indexing bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipelines as OP  # noqa: E402
from oracle.ar_model import CONFIG_375M  # noqa: E402
from oracle.scheduler import EDMDPMSolverMultistepScheduler as OracleScheduler  # noqa: E402
from tests.synth import make_ar, rel_l2, synth_known, tiny_ar_config  # noqa: E402

TOL = 1e-4


def to_hip(oracle_model, cfg):
    from ladcast_amd.models import LaDCastTransformer3DModel

    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(oracle_model.state_dict(), strict=True)
    return m.to("cuda").eval()


@pytest.fixture(scope="module")
def tiny_pair():
    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    o = make_ar(cfg)
    return o, to_hip(o, cfg)


@pytest.mark.parametrize("B,R,Bt", [(2, 4, 1), (1, 1, 1), (3, 2, 3)])
def test_tiny_forward_matches_oracle(tiny_pair, B, R, Bt):
    o, g = tiny_pair
    x = torch.randn(B, 84, R, 15, 30, generator=torch.Generator().manual_seed(3))
    known = synth_known(B)
    t = torch.linspace(-1.2, 1.0, Bt)
    ts = torch.tensor([2018010100])
    with torch.no_grad():
        want = o(x, t, known, time_elapsed=ts).sample
    got = g(x.cuda(), t.cuda(), known.cuda(), time_elapsed=ts.cuda()).sample
    assert got.shape == want.shape
    assert rel_l2(got.cpu(), want) < 2e-5
    got2 = g(x.cuda(), t.cuda(), known.cuda(), time_elapsed=ts.cuda(), return_dict=False)[0]
    assert torch.equal(got, got2)  # deterministic, workspace reuse is clean


def test_hip_graph_replay_is_bitwise_equal_to_eager(tiny_pair):
    """hipGraph mode: one captured graph per input shape, replayed with new inputs (timestep, state, timestamp)."""
    _, g = tiny_pair
    known = synth_known(2).cuda()
    outs = {}
    for mode in (False, True):
        g.enable_hip_graph(mode)
        res = []
        for i, (t, stamp) in enumerate([(0.3, 2018010100), (-0.7, 2018010100), (1.05, 2019063012)]):
            x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(10 + i)).cuda()
            res.append(g(x, torch.tensor([t]).cuda(), known, time_elapsed=torch.tensor([stamp]).cuda()).sample.clone())
        outs[mode] = res
    g.enable_hip_graph(False)
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    assert not torch.equal(outs[True][0], outs[True][1])


def test_edm_chunk_graph_is_bitwise_equal_to_eager(tiny_pair):
    """hipGraph mode captures the WHOLE Heun chunk (every forward + state update) in one graph: same launches, so the
    samples are bit-identical to eager launching, for new noise / new conditioning / new timestamps replayed through
    the same graph, and for a second chunk shape."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    _, g = tiny_pair
    pipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
    cases = [(2, synth_known(1), 2018010100, None), (2, synth_known(1) * 0.5 + 0.1, 2019063012, None), (1, synth_known(1), 2018010100, [1])]
    outs = {}
    for mode in (False, True, True):
        g.enable_hip_graph(mode)
        res = []
        for (n, known, stamp, ids) in cases:
            kw = dict(member_ids=ids) if ids else {}
            res.append(ensemble_AR_sampler(pipe, n, 4, 3, known_latents=known.cuda(), timestamps=torch.tensor([stamp]).cuda(), sampler_type="edm",
                                           device="cuda", **kw).clone())
        outs.setdefault(mode, []).append(res)
    # one captured chunk per chunk shape (new timestamps / conditioning / noise replay it), kept as two alternating instances
    assert len({k[:-1] for k in g._graphs if k[0] == "edm_chunk"}) == 2
    assert sum(1 for k, v in g._graphs.items() if k[0] == "edm_chunk" and isinstance(v, tuple)) == 4
    g.enable_hip_graph(False)
    assert g._graphs == {}
    eager, first, second = outs[False][0], outs[True][0], outs[True][1]
    for a, b, c in zip(eager, first, second):
        assert torch.equal(a, b) and torch.equal(a, c)  # capture pass and pure replays
    assert not torch.equal(eager[0], eager[1])


def test_pipeline_loop_graph_is_bitwise_equal_to_eager(tiny_pair):
    """the diffusers-style loop (scale_model_input -> model -> scheduler.step, N times) captured as one hipGraph: bit-identical
    samples for new noise / conditioning / timestamps, and the scheduler object ends in the state the eager loop leaves"""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    _, g = tiny_pair
    outs, states = {}, {}
    for mode in (False, True, True):
        g.enable_hip_graph(mode)
        pipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
        res = []
        for seed, stamp, scale in ((0, 2018010100, 1.0), (1, 2019063012, 0.5)):
            gen = [torch.Generator().manual_seed(seed + k) for k in range(2)]
            res.append(pipe(batch_size=2, return_seq_len=4, known_latents=(synth_known(2) * scale).cuda(), timestamps=torch.tensor([stamp]).cuda(),
                            generator=gen, num_inference_steps=5, return_dict=False)[0].clone())
        outs.setdefault(mode, []).append(res)
        sch = pipe.scheduler
        states.setdefault(mode, []).append((sch._step_index, sch.lower_order_nums, [m.clone() for m in sch.model_outputs]))
    g.enable_hip_graph(False)
    for a, b, c in zip(outs[False][0], outs[True][0], outs[True][1]):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert not torch.equal(outs[False][0][0], outs[False][0][1])
    (i0, l0, m0), (i1, l1, m1) = states[False][0], states[True][1]
    assert (i0, l0) == (i1, l1) == (5, 2) and all(torch.equal(x, y) for x, y in zip(m0, m1))
    g.enable_hip_graph(True)
    AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())(batch_size=2, return_seq_len=4, known_latents=synth_known(2).cuda(), timestamps=torch.tensor([2018010100]).cuda(),
                                                                  num_inference_steps=5)
    assert sum(1 for k in g._graphs if k[0] == "pipeline_loop") == 1
    g.enable_hip_graph(False)
    assert g._graphs == {}


def test_checkpoint_folder_loads_into_the_hip_path(tiny_pair, tmp_path):
    """a diffusers-layout folder (sharded here) written from the oracle's weights loads through
    ``from_pretrained`` and runs on the HIP path with the same result, bit for bit, as the model it was written from; the AR
    model is found the way training leaves it (``checkpoint-<step>/ar_model``, train_AR.py:561-570)."""
    from ladcast_amd.models import LaDCastTransformer3DModel
    from ladcast_amd.models.modeling_utils import checkpoint_model_folder

    o, g = tiny_pair
    g.save_pretrained(tmp_path / "checkpoint-200" / "ar_model", max_shard_size=150_000)
    g.save_pretrained(tmp_path / "checkpoint-1000" / "ar_model", max_shard_size=150_000)
    loaded = LaDCastTransformer3DModel.from_pretrained(checkpoint_model_folder(str(tmp_path), "latest")).to("cuda")
    assert str(tmp_path / "checkpoint-1000") in loaded._name_or_path
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    args = (x, torch.tensor([0.4]).cuda(), synth_known(2).cuda())
    ts = torch.tensor([2018010100]).cuda()
    assert torch.equal(loaded(*args, time_elapsed=ts).sample, g(*args, time_elapsed=ts).sample)
    with torch.no_grad():
        want = o(*(a.cpu() for a in args), time_elapsed=ts.cpu()).sample
    assert rel_l2(loaded(*args, time_elapsed=ts).sample.cpu(), want) < 2e-5


def test_tiny_forward_matches_golden_pin(tiny_pair, golden_dir):
    """The committed pin (made by the oracle in the build container) must also be hit by the HIP path."""
    _, g = tiny_pair
    pin = np.load(os.path.join(golden_dir, "oracle_pins.npz"))
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
    y = g(x.cuda(), torch.tensor([0.3]).cuda(), synth_known(1).cuda(), time_elapsed=torch.tensor([2018010100])).sample
    got = y.cpu().double().flatten()[::13][:4096]
    want = torch.from_numpy(pin["tiny_ar_fwd"])
    assert ((got - want).norm() / want.norm()).item() < 2e-5


def test_full_375m_forward_all_modes_one_and_two_members(oracle_375m):
    """The 375M model at full size, one forward: B = 1 at two noise levels and B = 2 (BASELINE configs[2]'s per-GPU batch: 16 members
    on 8 GPUs) in the three arithmetic modes, each at its stated tolerance (ladcast_amd/precision.py); the single-term `bf16` mode
    additionally against the like-for-like comparator - the oracle under the reference's own mixed precision (oracle/autocast.py)."""
    from ladcast_amd.precision import tolerance
    from oracle import autocast as OA

    o = oracle_375m
    g = to_hip(o, dict(CONFIG_375M))
    known, ts = synth_known(1), torch.tensor([2018010100])
    cases = [(1, 1.0955067, 3), (1, -1.553652, 3), (2, 0.3, 5)]
    for i, (B, t, seed) in enumerate(cases):
        x = torch.randn(B, 84, 4, 15, 30, generator=torch.Generator().manual_seed(seed))
        kn = known.expand(B, -1, -1, -1, -1)
        with torch.no_grad():
            want = o(x, torch.tensor([t]), kn, time_elapsed=ts).sample
            e_auto = None
            if i == 0:
                with OA.reference_autocast("cuda"):
                    e_auto = rel_l2(o(x, torch.tensor([t]), kn, time_elapsed=ts).sample.float(), want)
        errs = {}
        for mode in ("fp32", "bf16x3", "bf16"):
            g.set_gemm_precision(mode)
            got = g(x.cuda(), torch.tensor([t]).cuda(), known.cuda(), time_elapsed=ts.cuda()).sample
            errs[mode] = rel_l2(got.cpu(), want)
            assert errs[mode] < tolerance(mode, "forward"), (B, t, mode, errs)
            if B == 2:  # members of one batch are independent: member 1 alone gives the same values as inside the batch (not the same
                # bits: the stream-K unit ranges of a two-member launch cut the k loops of its tiles elsewhere)
                alone = g(x[1:].cuda(), torch.tensor([t]).cuda(), known.cuda(), time_elapsed=ts.cuda()).sample
                assert rel_l2(alone[0], got[1]) < (1e-6 if mode == "fp32" else 2e-5 if mode == "bf16x3" else 2e-3), mode
        print(f"\n375M forward B = {B}, c_noise = {t}: rel-L2 vs fp32 oracle " + ", ".join(f"{m} {e:.2e}" for m, e in errs.items())
              + (f"; oracle under autocast {e_auto:.2e}" if e_auto is not None else ""))
        assert errs["bf16"] > 1e-5  # the single-term mode is really on
        if e_auto is not None:
            assert errs["bf16"] <= e_auto, (errs, e_auto)
    g.set_gemm_precision("fp32")


@pytest.mark.parametrize("sampler_type", ["edm", "pipeline"])
def test_tiny_sampler_chunk_matches_oracle(tiny_pair, sampler_type):
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o, g = tiny_pair
    known, ts = synth_known(1), torch.tensor([2018010100])
    opipe = OP.AutoRegressive2DPipeline(o, OracleScheduler())
    want = OP.ensemble_AR_sampler(opipe, 3, 4, 6, known_latents=known, timestamps=ts, sampler_type=sampler_type)
    gpipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
    got = ensemble_AR_sampler(gpipe, 3, 4, 6, known_latents=known.cuda(), timestamps=ts.cuda(), sampler_type=sampler_type, device="cuda")
    assert rel_l2(got.cpu(), want) < TOL
    # sharding property: member 2 alone == member 2 of the full ensemble (seed-by-member noise)
    part = ensemble_AR_sampler(gpipe, 1, 4, 6, known_latents=known.cuda(), timestamps=ts.cuda(), sampler_type=sampler_type, device="cuda", member_ids=[2])
    assert rel_l2(part.cpu(), got[2:3].cpu()) < 1e-5


def test_stochastic_churn_chunk_matches_oracle(tiny_pair):
    """edm_AR_sampler(deterministic=False, S_churn, S_min, S_max, S_noise, randn_like) (pipelines/edm_sampler.py:67-76) with the real
    (tiny) network: same noise stream on both sides (a seeded CPU generator behind the caller's randn_like), hipGraph mode on (the
    stochastic chunk is launched eagerly whatever the mode) - within the chunk budget of the oracle; and the noise is really used"""
    from ladcast_amd.pipelines import edm_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o, g = tiny_pair
    known, ts = synth_known(2), torch.tensor([2018010100])
    kw = dict(batch_size=2, return_seq_len=4, num_inference_steps=6, timestamps=None, deterministic=False, S_churn=3.0, S_min=0.01, S_max=60.0,
              S_noise=1.007)

    def noise_fn(seed):
        gc = torch.Generator("cpu").manual_seed(seed)
        return lambda x: torch.randn(x.shape, generator=gc, dtype=x.dtype).to(x.device)

    gens = lambda: [torch.Generator("cpu").manual_seed(k) for k in range(2)]  # noqa: E731
    want = OP.edm_AR_sampler(o, OracleScheduler(), known_latents=known, generator=gens(), randn_like=noise_fn(5), **kw)
    g.enable_hip_graph(True)
    try:
        got = edm_AR_sampler(g, EDMDPMSolverMultistepScheduler(), known_latents=known.cuda(), generator=gens(), randn_like=noise_fn(5), device="cuda", **kw)
        other = edm_AR_sampler(g, EDMDPMSolverMultistepScheduler(), known_latents=known.cuda(), generator=gens(), randn_like=noise_fn(6), device="cuda", **kw)
    finally:
        g.enable_hip_graph(False)
    assert rel_l2(got.cpu(), want) < TOL
    assert rel_l2(other.cpu(), want) > 1e-2  # another noise stream, another sample
    del ts


def test_bf16x3_mode_stays_inside_the_parity_budget(tiny_pair):
    """split-bf16 GEMM mode: per-forward and per-chunk rel-L2 vs the fp32 CPU oracle.  Budget: 1e-4 (north star);
    measured on CPU emulation: 4e-6 / 2e-6 (tests/split_precision_study.py)."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o, g = tiny_pair
    g.set_gemm_precision("bf16x3")
    try:
        x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
        known, ts = synth_known(1), torch.tensor([2018010100])
        with torch.no_grad():
            want = o(x, torch.tensor([0.3]), known.expand(2, -1, -1, -1, -1), time_elapsed=ts).sample
        got = g(x.cuda(), torch.tensor([0.3]).cuda(), known.cuda(), time_elapsed=ts.cuda()).sample
        e_fwd = rel_l2(got.cpu(), want)
        assert e_fwd < 3e-5, e_fwd
        opipe = OP.AutoRegressive2DPipeline(o, OracleScheduler())
        want = OP.ensemble_AR_sampler(opipe, 2, 4, 20, known_latents=known, timestamps=ts, sampler_type="edm")
        gpipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
        got = ensemble_AR_sampler(gpipe, 2, 4, 20, known_latents=known.cuda(), timestamps=ts.cuda(), sampler_type="edm", device="cuda")
        e_chunk = rel_l2(got.cpu(), want)
        print(f"bf16x3: forward rel-L2 {e_fwd:.2e}, 20-step Heun chunk rel-L2 {e_chunk:.2e}")
        assert e_chunk < TOL, e_chunk
    finally:
        g.set_gemm_precision("fp32")


def test_full_1_6b_forward_matches_oracle_both_modes():
    """BASELINE configs[3] model (1.6B: D = 2048, 16 heads, 3 + 5 + 10 blocks), one forward, both arithmetic modes."""
    from oracle.ar_model import CONFIG_1_6B

    o = make_ar(dict(CONFIG_1_6B))
    g = to_hip(o, dict(CONFIG_1_6B))
    x = torch.randn(1, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
    known, ts = synth_known(1), torch.tensor([2018010100])
    with torch.no_grad():
        want = o(x, torch.tensor([0.3]), known, time_elapsed=ts).sample
    del o
    for mode, tol in (("fp32", 2e-5), ("bf16x3", 3e-5)):
        g.set_gemm_precision(mode)
        got = g(x.cuda(), torch.tensor([0.3]).cuda(), known.cuda(), time_elapsed=ts.cuda()).sample
        e = rel_l2(got.cpu(), want)
        print(f"1.6B {mode} forward rel-L2 {e:.2e}")
        assert e < tol, (mode, e)


def test_scheduler_indexing_is_bit_exact():
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    a, b = EDMDPMSolverMultistepScheduler(), OracleScheduler()
    a.set_timesteps(20, device="cuda")
    b.set_timesteps(20)
    assert torch.equal(a.sigmas, b.sigmas) and torch.equal(a.timesteps.cpu(), b.timesteps)
    x = torch.randn(2, 84, 1, 15, 30, generator=torch.Generator().manual_seed(0))
    xa, xb = x.cuda(), x.clone()
    for t in b.timesteps:
        ia, ib = a.scale_model_input(xa, t), b.scale_model_input(xb, t)
        assert a.step_index == b.step_index
        assert torch.equal(ia.cpu(), ib)
        f = torch.tanh(ib)
        xa = a.step(f.cuda(), t, xa, return_dict=False)[0]
        xb = b.step(f, t, xb, return_dict=False)[0]
        assert torch.equal(xa.cpu(), xb)  # fp32 update reproduces torch's rounding


def test_tiny_rollout_latent_mode_matches_oracle(tiny_pair):
    """roll_out_serial tensor contract in latent mode with a synthetic IC latent:
    slot 0 = un-normalised IC, chunks chained through known = samples[:, :, -T_in:], last chunk truncated."""
    from datetime import datetime

    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o, g = tiny_pair
    g_ = torch.Generator().manual_seed(9)
    mu, sd = torch.randn(84, generator=g_) * 0.2, torch.rand(84, generator=g_) + 0.5
    targs = {"mean": mu.tolist(), "std": sd.tolist(), "target_std": 0.5}
    ic = synth_known(1)[0] * 2 * sd[:, None, None, None] + mu[:, None, None, None]  # un-normalised latent (C,1,15,30)

    class FakeAE:  # oracle-side stand-in: "encode" returns the given latent
        device = torch.device("cpu")
        config = type("c", (), dict(latent_channels=84, out_channels=89, static_channels=5))

        def encode(self, x, static_conditioning_tensor=None):
            return type("o", (), dict(latent=ic.permute(1, 0, 2, 3)))

    t0 = [datetime(2018, 1, 1, 0)]
    want = OP.roll_out_serial(
        lambda t: torch.zeros(84, 1, 120, 240), t0, OP.AutoRegressive2DPipeline(o, OracleScheduler()), ensemble_size=2, num_inference_steps=3,
        return_seq_len=2, encdec_model=FakeAE(), static_tensor4encdec=torch.zeros(5, 120, 240), latent_transform_args=targs,
        total_lead_time_hour=18, sampler_type="edm", return_latent=True,
    )
    got = roll_out_serial(
        None, t0, AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), ensemble_size=2, num_inference_steps=3, return_seq_len=2,
        latent_transform_args=targs, total_lead_time_hour=18, sampler_type="edm", return_latent=True, known_latents_override=ic,
        log_pred_interval_hour=6,
    )
    assert got.shape == want.shape == (1, 2, 84, 4, 15, 30)
    assert not torch.isnan(got).any()
    assert torch.equal(got[:, :, :, 0], want[:, :, :, 0])
    assert rel_l2(got, want) < TOL


def test_rollouts_queued_without_host_sync_equal_the_synchronised_ones(tiny_pair):
    """round 3: `output_device=<gpu>` returns a rollout's tensor without synchronising (results collected in HBM, pinned + non-blocking
    uploads of the noise draw / timestamps / calendar embedding, no device read-back), so the host prepares call k + 1 while the GPU runs
    call k - what bench.py times.  Three calls with different members and ICs queued back to back, eager and graph-replayed, must equal
    the same calls made one at a time with host outputs, bit for bit; the default (host) return is unchanged."""
    from datetime import datetime

    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    _, g = tiny_pair
    pipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
    targs = {"mean": [0.1] * 84, "std": [1.3] * 84, "target_std": 0.5}
    ics = [synth_known(1)[0] * s for s in (1.0, 2.0, 0.5)]
    t0s = [[datetime(2018, 1, 1, 0)], [datetime(2019, 6, 30, 18)], [datetime(2018, 1, 1, 0)]]
    kw = dict(ensemble_size=2, num_inference_steps=3, return_seq_len=2, latent_transform_args=targs, total_lead_time_hour=18, sampler_type="edm",
              return_latent=True)
    for graph in (False, True):
        g.enable_hip_graph(graph)
        want = [roll_out_serial(None, t0s[i], pipe, known_latents_override=ics[i], member_ids=[i, i + 5], **kw) for i in range(3)]
        assert all(w.device.type == "cpu" and not torch.isnan(w).any() for w in want)
        got = [roll_out_serial(None, t0s[i], pipe, known_latents_override=ics[i].cuda(), member_ids=[i, i + 5], output_device="cuda", **kw)
               for i in range(3)]  # no synchronisation in between
        assert all(x.device.type == "cuda" for x in got)
        torch.cuda.synchronize()
        for x, w in zip(got, want):
            assert torch.equal(x.cpu(), w)
        assert not torch.equal(want[0], want[2])  # different members: the queued calls did not overwrite each other's inputs
    g.enable_hip_graph(False)


def test_end_to_end_rollout_with_dcae_matches_oracle():
    """encode IC -> AR chunks -> decode (tiny DCAE + tiny AR), the decoded-field mode of roll_out_serial:
    the whole product path on HIP vs the whole oracle path on CPU, same weights, same seeds."""
    from datetime import datetime

    from ladcast_amd.models import AutoencoderDC, LaDCastTransformer3DModel
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from oracle.dcae import CONFIG_DCAE_84
    from tests.synth import make_dcae, synth_field

    # latent grid must be 15x30 with 84 channels for the AR model: field 120x240, DCAE with small widths
    # (widths must keep the reference's shortcut arithmetic integral: 4*c_i % c_{i+1} == 0 and c_last % 84 == 0)
    ae_cfg = dict(CONFIG_DCAE_84, encoder_block_out_channels=(84, 84, 84, 168), decoder_block_out_channels=(84, 84, 84, 168),
                  encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(1, 1, 1, 1))
    oae = make_dcae(ae_cfg)
    gae = AutoencoderDC.from_config(ae_cfg)
    gae.load_state_dict(oae.state_dict(), strict=True)
    gae = gae.cuda().eval()
    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    oar = make_ar(cfg)
    gar = to_hip(oar, cfg)
    field = synth_field(84, 1, 120, 240)  # (C, T_in, H, W), already "normalised"
    static = synth_field(1, 5, 120, 240, seed=1)[0]
    g_ = torch.Generator().manual_seed(9)
    mu, sd = torch.randn(84, generator=g_) * 0.2, torch.rand(84, generator=g_) + 0.5
    targs = {"mean": mu.tolist(), "std": sd.tolist(), "target_std": 0.5}
    fmu, fsd = torch.randn(84, generator=g_), torch.rand(84, generator=g_) + 0.5
    t0 = [datetime(2018, 1, 1, 0)]
    kw = dict(ensemble_size=2, num_inference_steps=2, return_seq_len=2, static_tensor4encdec=static, latent_transform_args=targs,
              total_lead_time_hour=18, sampler_type="edm", return_latent=False, encdec_model_type="ae")
    raw = field[:, -1] * fsd[:, None, None] + fmu[:, None, None]  # the un-normalised IC field the reference reads from its dataset
    want = OP.roll_out_serial(lambda t: field, t0, OP.AutoRegressive2DPipeline(oar, OracleScheduler()), mean_tensor=fmu, std_tensor=fsd,
                              encdec_model=oae, raw_input_fields=lambda t: raw, **kw)
    got = roll_out_serial(lambda t: field, t0, AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler()),
                          normalization_param_dict={"mean": fmu, "std": fsd}, encdec_model=gae, **kw)
    assert got.shape == want.shape == (1, 2, 84, 4, 120, 240)
    assert torch.equal(got[:, :, :, 0], want[:, :, :, 0])  # slot 0 = raw IC field (pipelines/utils.py:462-468), here de-normalised from the input
    got_raw = roll_out_serial(lambda t: field, t0, AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler()), encdec_model=gae,
                              raw_input_fields=lambda t: raw + 1.0, **dict(kw, total_lead_time_hour=6))
    assert torch.equal(got_raw[0, 1, :, 0], raw + 1.0)  # an explicit raw-field callable wins; without statistics decoded fields stay normalised
    with pytest.raises(ValueError):  # input_seq_len frames are the caller's job and are checked
        roll_out_serial(lambda t: field, t0, AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler()), encdec_model=gae, input_seq_len=2, **kw)
    assert rel_l2(got[:, :, :, 1:], want[:, :, :, 1:]) < TOL
    # everything on the bf16 matrix cores: AR GEMMs / attention and the DCAE convs in split-bf16 mode, same 1e-4 budget
    gae.set_gemm_precision("bf16x3")
    gar.set_gemm_precision("bf16x3")
    got3 = roll_out_serial(lambda t: field, t0, AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler()),
                           normalization_param_dict={"mean": fmu, "std": fsd}, encdec_model=gae, **kw)
    assert rel_l2(got3[:, :, :, 1:], want[:, :, :, 1:]) < TOL


def test_driver_counterpart_writes_the_reference_file_layout(tmp_path):
    """`ladcast_amd.evaluate.pred_rollout.run_rollout` (the tensor-level counterpart of evaluate/pred_rollout.py) - prepared
    inputs -> encode -> rollout -> `latent_YYYYMMDDHH.npy` per initial time holding (ens, 84, 1 + steps, 15, 30) with slot 0 = the un-normalised
    IC latent; values vs the oracle's roll_out_serial on the same prepared inputs."""
    from datetime import datetime

    from ladcast_amd.evaluate.pred_rollout import build_static_conditioning, fill_sst_nan, load_latent_transform_args, run_rollout
    from ladcast_amd.models import AutoencoderDC
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, list_latent_files, load_latent_npy
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from oracle.dcae import CONFIG_DCAE_84
    from tests.synth import make_dcae, synth_field

    ae_cfg = dict(CONFIG_DCAE_84, encoder_block_out_channels=(84, 84, 84, 168), decoder_block_out_channels=(84, 84, 84, 168),
                  encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(1, 1, 1, 1))
    oae = make_dcae(ae_cfg)
    gae = AutoencoderDC.from_config(ae_cfg)
    gae.load_state_dict(oae.state_dict(), strict=True)
    gae = gae.cuda().eval()
    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    oar = make_ar(cfg)
    gar = to_hip(oar, cfg)
    g_ = torch.Generator().manual_seed(12)
    static = build_static_conditioning(torch.rand(121, 240, generator=g_), torch.randn(4, 121, 240, generator=g_) * 50 + 100)
    assert static.shape == (5, 120, 240)
    fields = {}
    for k, t in enumerate([datetime(2018, 1, 1, 0), datetime(2018, 1, 3, 12)]):
        f = synth_field(84, 1, 120, 240, seed=20 + k)
        f[82, :, ::7, ::5] = float("nan")  # land points of the SST channel
        fields[t] = fill_sst_nan(f)
        assert not torch.isnan(fields[t]).any() and (fields[t][82, 0, ::7, ::5] == -2).all()
    targs = load_latent_transform_args({"mean": (torch.randn(84, generator=g_) * 0.2).tolist(), "std": (torch.rand(84, generator=g_) + 0.5).tolist()})
    assert targs["target_std"] == 0.5
    kw = dict(ensemble_size=3, num_inference_steps=2, return_seq_len=2, total_lead_time_hour=18, sampler_type="edm")
    got = run_rollout(lambda t: fields[t], list(fields), AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler()), gae, targs,
                      static_conditioning_tensor=static, output=str(tmp_path), **kw)
    files = list_latent_files(str(tmp_path))
    assert [n for n, _ in files] == ["2018010100", "2018010312"]
    for (name, path), t, mine in zip(files, fields, got):
        arr, stamp = load_latent_npy(path)
        assert stamp == int(name) and arr.shape == (3, 84, 4, 15, 30) and torch.equal(arr, mine.cpu())
        want = OP.roll_out_serial(lambda _t: fields[t], [t], OP.AutoRegressive2DPipeline(oar, OracleScheduler()), encdec_model=oae, encdec_model_type="ae",
                                  static_tensor4encdec=static, latent_transform_args=targs, return_latent=True, **kw)[0]
        assert rel_l2(arr, want) < TOL
        assert rel_l2(arr[:, :, 0], want[:, :, 0]) < 2e-5  # slot 0: the encoded, un-normalised initial condition


def test_bf16_single_term_mode(tiny_pair):
    """BASELINE configs[4] ("fp16/bf16 mixed"): `set_gemm_precision("bf16")` = one bf16 MFMA per product in the token-stream GEMMs and
    both attention contractions, everything the reference pins to fp32 under autocast left in fp32 (temb: LaDCast_3D_model.py:953;
    norms, softmax statistics, residual stream, fp64 sampler state).  Stated tolerances vs the fp32 oracle: ladcast_amd/precision.py (measured x 2: 7e-3 per forward, 4e-3 per 20-step Heun chunk; bf16 has
    8 significand bits: 2^-9 = 2e-3 per rounding); the 1e-4 budget does not apply to this mode."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.precision import tolerance
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o, g = tiny_pair
    g.set_gemm_precision("bf16")
    try:
        x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
        known, ts = synth_known(1), torch.tensor([2018010100])
        with torch.no_grad():
            want = o(x, torch.tensor([0.3]), known.expand(2, -1, -1, -1, -1), time_elapsed=ts).sample
        got = g(x.cuda(), torch.tensor([0.3]).cuda(), known.cuda(), time_elapsed=ts.cuda()).sample
        e_fwd = rel_l2(got.cpu(), want)
        opipe = OP.AutoRegressive2DPipeline(o, OracleScheduler())
        want = OP.ensemble_AR_sampler(opipe, 2, 4, 20, known_latents=known, timestamps=ts, sampler_type="edm")
        gpipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
        got = ensemble_AR_sampler(gpipe, 2, 4, 20, known_latents=known.cuda(), timestamps=ts.cuda(), sampler_type="edm", device="cuda")
        e_chunk = rel_l2(got.cpu(), want)
        g.enable_hip_graph(True)
        got_g = ensemble_AR_sampler(gpipe, 2, 4, 20, known_latents=known.cuda(), timestamps=ts.cuda(), sampler_type="edm", device="cuda")
        g.enable_hip_graph(False)
        print(f"\nbf16 (single-term): tiny forward rel-L2 {e_fwd:.2e}, 20-step Heun chunk rel-L2 {e_chunk:.2e}")
        assert 1e-5 < e_fwd < tolerance("bf16", "forward"), e_fwd  # lower bound: the mode is really on
        assert e_chunk < tolerance("bf16", "chunk_edm"), e_chunk
        assert torch.equal(got_g, got)
    finally:
        g.set_gemm_precision("fp32")


def test_foreign_attention_processor_is_called_not_ignored(tiny_pair):
    """`set_attn_processor` (models/LaDCast_3D_model.py:793-827) with a processor that is not the built-in one: the forward calls it
    with the reference's protocol.  (1) The oracle's restatement of the reference processor, installed as a FOREIGN processor on the
    HIP model, reproduces the oracle forward (fp32 tolerance) and the fused path; (2) a processor that drops the sample stream's
    attention output changes the result - it really ran; (3) a mixed dict; (4) the split modes and hipGraph capture refuse it."""
    from ladcast_amd.models import LaDCastAttnProcessor2_0
    from oracle.ar_model import LaDCastAttnProcessor as OracleProcessor

    o, g = tiny_pair
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
    known, ts, t = synth_known(2), torch.tensor([2018010100]), torch.tensor([0.3])
    with torch.no_grad():
        want = o(x, t, known, time_elapsed=ts).sample
    args = (x.cuda(), t.cuda(), known.cuda())
    fused = g(*args, time_elapsed=ts.cuda()).sample
    seen = []

    class Counting(OracleProcessor):
        def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None, cond_image_rotary_emb=None):
            seen.append((tuple(hidden_states.shape), None if encoder_hidden_states is None else tuple(encoder_hidden_states.shape),
                         image_rotary_emb is not None, cond_image_rotary_emb is not None))
            return super().__call__(attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb, cond_image_rotary_emb)

    try:
        g.set_attn_processor(Counting())
        got = g(*args, time_elapsed=ts.cuda()).sample
        assert seen == [((2, 450, 256), None, True, False), ((2, 1800, 256), (2, 450, 256), True, True), ((2, 1800, 256), (2, 450, 256), True, True)]
        assert rel_l2(got.cpu(), want) < 2e-5 and rel_l2(got, fused) < 2e-5

        class ZeroAttention(OracleProcessor):  # a processor with different arithmetic: the sample stream's attention output is dropped
            def __call__(self, attn, hidden_states, encoder_hidden_states=None, **kw):
                a, ca = super().__call__(attn, hidden_states, encoder_hidden_states, **kw)
                return torch.zeros_like(a), ca

        names = list(g.attn_processors)
        g.set_attn_processor({k: (ZeroAttention() if "single" in k else LaDCastAttnProcessor2_0()) for k in names})
        changed = g(*args, time_elapsed=ts.cuda()).sample
        assert rel_l2(changed, fused) > 1e-3
        g.set_gemm_precision("bf16x3")
        with pytest.raises(NotImplementedError):
            g(*args, time_elapsed=ts.cuda())
        g.set_gemm_precision("fp32")
        with pytest.raises(NotImplementedError):
            g.enable_hip_graph(True)
    finally:
        g.set_gemm_precision("fp32")
        g.set_attn_processor(LaDCastAttnProcessor2_0())
    assert torch.equal(g(*args, time_elapsed=ts.cuda()).sample, fused)  # back on the fused path, bit for bit


def test_duck_typed_ddim_scheduler_through_the_hip_model(tiny_pair):
    """north star: "the DDIM/DDPM scheduler loop ... diffusers-style scheduler surface" (pipelines/pipeline_AR.py:19-21,85-102).  A
    scheduler that is NOT this build's EDM class - DDIM-shaped, only set_timesteps / timesteps / scale_model_input / step, integer
    timesteps - goes through `AutoRegressive2DPipeline.__call__` with the HIP model (eager loop, also with graphs switched on: the
    whole-loop capture needs this build's scheduler and falls back to per-forward graphs) and the oracle pipeline drives the same
    scheduler class with the oracle model."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline
    from tests.synth import DuckDDIMScheduler

    o, g = tiny_pair
    known, ts = synth_known(2), torch.tensor([2018010106])
    so = DuckDDIMScheduler()
    want = OP.AutoRegressive2DPipeline(o, so)(batch_size=2, return_seq_len=4, known_latents=known, timestamps=ts,
                                             generator=[torch.Generator().manual_seed(k) for k in range(2)], num_inference_steps=6, return_dict=False)[0]
    outs = []
    for graphs in (False, True):
        g.enable_hip_graph(graphs)
        sg = DuckDDIMScheduler()
        outs.append(AutoRegressive2DPipeline(g, sg)(batch_size=2, return_seq_len=4, known_latents=known.cuda(), timestamps=ts.cuda(),
                                                    generator=[torch.Generator().manual_seed(k) for k in range(2)], num_inference_steps=6, return_dict=False)[0])
        assert sg.calls == so.calls and len(sg.calls) == 12
    g.enable_hip_graph(False)
    assert torch.equal(outs[0], outs[1])
    e = rel_l2(outs[0].cpu(), want)
    print(f"\nduck-typed DDIM scheduler, 6 steps, tiny AR model: rel-L2 vs the oracle pipeline with the same scheduler {e:.2e}")
    assert e < TOL


def test_batched_conditioning_equals_per_evaluation_conditioning(tiny_pair):
    """`prepare_conditioning`: the sample-independent part of the forward (context refiner, conditioning embedding, AdaLN modulation
    vectors) for all noise levels of a chunk in one batch == the same part computed inside each forward (same kernels on a (level,
    member) batch: equal to fp32 rounding - another stream-K cut of the same sums -, not bit for bit), also when the batch is cut
    into several passes; and a whole Heun chunk with the batching switched off gives the same sample."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o, g = tiny_pair
    known, stamp = synth_known(2).cuda(), torch.tensor([2019063012]).cuda()
    ts = torch.tensor([0.3, -0.7, 1.05, -1.4])
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    try:
        for mode, tol in (("fp32", 1e-6), ("bf16x3", 2e-5), ("bf16", 5e-3)):
            g.set_gemm_precision(mode)
            te = g.time_elapsed_embedding(stamp)
            plain = [g(x, t.reshape(1).cuda(), known, time_elapsed=stamp).sample.clone() for t in ts]
            for rows in (96, 4, 2):  # one pass; two passes of two levels; four passes
                g.COND_MAX_ROWS = rows
                pack = g.prepare_conditioning(ts.cuda(), known, te)
                assert pack.levels == 4 and pack.ctx.shape[0] == 8
                for i, t in enumerate(ts):
                    got = g(x, t.reshape(1).cuda(), known, time_elapsed=stamp, conditioning=(pack, i)).sample
                    assert rel_l2(got, plain[i]) < tol, (mode, rows, i)
            g.COND_MAX_ROWS = 96
            with pytest.raises(ValueError):
                g(x[:1], ts[:1].cuda(), known[:1], time_elapsed=stamp, conditioning=(pack, 0))  # prepared for two members
            pipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
            outs = {}
            for flag in (True, False):
                g.batch_conditioning = flag
                outs[flag] = ensemble_AR_sampler(pipe, 2, 4, 6, known_latents=known[:1], timestamps=stamp, sampler_type="edm", device="cuda")
                pl = ensemble_AR_sampler(pipe, 2, 4, 6, known_latents=known[:1], timestamps=stamp, sampler_type="pipeline", device="cuda")
                outs[("p", flag)] = pl
            assert rel_l2(outs[True], outs[False]) < 20 * tol and rel_l2(outs[("p", True)], outs[("p", False)]) < 20 * tol, mode
    finally:
        g.COND_MAX_ROWS = 96
        g.batch_conditioning = True
        g.set_gemm_precision("fp32")


def test_model_cast_to_bf16_is_upcast_with_a_warning(tiny_pair):
    """a model held in bf16 / fp16 (`.to(torch.bfloat16)`) used to raise at the first forward; now its parameters are
    up-cast to fp32 once, with a warning, and the forward equals the one of the same (rounded) weights held in fp32 - bit for bit."""
    import warnings

    from ladcast_amd.models import AutoencoderDC
    from tests.synth import make_dcae, tiny_dcae_config

    from ladcast_amd.models import LaDCastTransformer3DModel

    o, _ = tiny_pair
    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(o.state_dict(), strict=True)
    m = m.to("cuda").eval().to(torch.bfloat16)
    assert m.dtype == torch.bfloat16
    ref = LaDCastTransformer3DModel.from_config(cfg)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()}, strict=True)  # the rounded weights, held in fp32
    ref = ref.to("cuda").eval()
    x = torch.randn(1, 84, 2, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    known, ts, t = synth_known(1).cuda(), torch.tensor([2018010100]).cuda(), torch.tensor([0.3]).cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = m(x, t, known, time_elapsed=ts).sample
    assert any("up-casting" in str(i.message) for i in w) and m.dtype == torch.float32
    assert torch.equal(got, ref(x, t, known, time_elapsed=ts).sample)
    # the samplers drew their initial noise in `net.dtype` BEFORE the first forward had up-cast the model - the first call on a
    # bf16-cast model raised (or drew a bf16 noise stream).  Both samplers, graph and eager, on a freshly cast model each time.
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    want = {st: ensemble_AR_sampler(AutoRegressive2DPipeline(ref, EDMDPMSolverMultistepScheduler()), 2, 2, 3, known_latents=known, timestamps=ts,
                                    sampler_type=st, device="cuda") for st in ("edm", "pipeline")}
    for st in ("edm", "pipeline"):
        for graph in (True, False):
            mb = LaDCastTransformer3DModel.from_config(cfg)
            mb.load_state_dict(o.state_dict(), strict=True)
            mb = mb.to("cuda").eval().to(torch.bfloat16)
            mb.use_hip_graph = graph
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = ensemble_AR_sampler(AutoRegressive2DPipeline(mb, EDMDPMSolverMultistepScheduler()), 2, 2, 3, known_latents=known, timestamps=ts,
                                          sampler_type=st, device="cuda")
            assert got.dtype == torch.float32 and torch.equal(got, want[st]), (st, graph)
    cfg = tiny_dcae_config()
    od = make_dcae(cfg)
    ae = AutoencoderDC.from_config(cfg)
    ae.load_state_dict(od.state_dict(), strict=True)
    ae = ae.cuda().eval().half()
    f, st = torch.randn(1, 8, 48, 64).cuda(), torch.randn(1, 5, 48, 64).cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        z = ae.encode(f, static_conditioning_tensor=st).latent
    assert any("up-casting" in str(i.message) for i in w) and ae.dtype == torch.float32 and torch.isfinite(z).all()
