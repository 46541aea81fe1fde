"""Full-size CHAIN parity (VERDICT r01 "What's weak" 2): error growth over chained forwards is the hard part of this path
(sigma 80 -> 0.002, c_in down to 0.0125, 39 network evaluations per chunk, chunk output feeding the next chunk), and the
split-bf16 mode is exactly where it would show.  These tests run BASELINE's literal workloads against the CPU oracle on the
GPU box and PRINT the error curves (run with -s to see them; the driver's log keeps them).

Budget (north star): rel-L2 <= 1e-4 per sampler chunk / rollout for the fp32 and bf16x3 modes.
"""
import time
from datetime import datetime

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipelines as OP  # noqa: E402
from oracle.ar_model import CONFIG_375M  # noqa: E402
from oracle.scheduler import EDMDPMSolverMultistepScheduler as OracleScheduler  # noqa: E402
from tests.synth import make_ar, oracle_threads, rel_l2, synth_known, tiny_ar_config  # noqa: E402

TOL = 1e-4


def to_hip(oracle_model, cfg):
    from ladcast_amd.models import LaDCastTransformer3DModel

    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(oracle_model.state_dict(), strict=True)
    return m.to("cuda").eval()


from tests.conftest import RecordingNet as Rec  # noqa: E402


def _curve(a, b):
    return [rel_l2(x, y) for x, y in zip(a, b)]


def _fmt(c):
    return " ".join(f"{v:.1e}" for v in c)


# stated tolerances of the single-term `bf16` mode on the 375M chunk: measured x 2
from ladcast_amd.precision import tolerance  # noqa: E402  (the one table of stated tolerances)


def test_full_375m_chunk_matches_oracle(fullsize_chunk_oracle):
    """BASELINE configs[1], literally: 375M, 1 member, 20 solver steps, one R = 4 chunk - the 39-forward Heun sampler (`edm`) and
    the 20-forward DPM-Solver++(2M) loop (`pipeline`), exact-fp32, split-bf16 and single-term bf16 arithmetic, against the CPU
    oracle; the per-evaluation error of the network INPUT (state drift) and OUTPUT is printed.  The known latent is the (normalised)
    oracle-DCAE latent of the synthetic frame, so that the Heun chunk's oracle run is shared with tests/test_gpu_cfg5.py."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    fx = fullsize_chunk_oracle
    o = fx.ar
    g = to_hip(o, dict(CONFIG_375M))
    known, ts = fx.known, fx.ts
    for sampler, n_fwd in (("edm", 39), ("pipeline", 20)):
        if sampler == "edm":
            want, ins, outs, t_cpu = fx.want, fx.ins, fx.outs, fx.seconds
        elif fx.pipeline is not None:  # committed oracle outputs (tests/golden/fullsize_375m_chunk.npz)
            want, ins, outs, t_cpu = fx.pipeline.want, fx.pipeline.ins, fx.pipeline.outs, fx.pipeline.seconds
        else:
            ro = Rec(o)
            t0 = time.perf_counter()
            want = OP.ensemble_AR_sampler(OP.AutoRegressive2DPipeline(ro, OracleScheduler()), 1, 4, 20, known_latents=known, timestamps=ts, sampler_type=sampler)
            t_cpu = time.perf_counter() - t0
            ins, outs = ro.ins, ro.outs
        assert len(outs) == n_fwd
        for mode in ("fp32", "bf16x3", "bf16"):
            # bf16: the single-term mixed-precision mode (BASELINE configs[4]); own stated tolerance per chunk
            tol = tolerance("bf16", f"chunk_{sampler}") if mode == "bf16" else TOL
            tol_out = tolerance("bf16", "chunk_network_output") if mode == "bf16" else TOL  # fp32 / bf16x3: the 1e-4 budget, no slack
            g.set_gemm_precision(mode)
            rg = Rec(g)
            got = ensemble_AR_sampler(AutoRegressive2DPipeline(rg, EDMDPMSolverMultistepScheduler()), 1, 4, 20, known_latents=known.cuda(),
                                      timestamps=ts.cuda(), sampler_type=sampler, device="cuda")
            assert len(rg.outs) == n_fwd
            e_in, e_out, e = _curve(rg.ins, ins), _curve(rg.outs, outs), rel_l2(got.cpu(), want)
            print(f"\n375M {sampler} chunk ({n_fwd} forwards, oracle {t_cpu:.0f} s) [{mode}]: sample rel-L2 {e:.2e}")
            print(f"  network-input  error per evaluation: {_fmt(e_in)}")
            print(f"  network-output error per evaluation: {_fmt(e_out)}")
            assert e < tol, (sampler, mode, e)
            assert max(e_in) < tol and max(e_out) < tol_out, (sampler, mode, max(e_in), max(e_out))
            # the graph-replayed chunk (what bench.py times) gives the same sample bit for bit
            g.enable_hip_graph(True)
            got_g = ensemble_AR_sampler(AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), 1, 4, 20, known_latents=known.cuda(),
                                        timestamps=ts.cuda(), sampler_type=sampler, device="cuda")
            g.enable_hip_graph(False)
            assert torch.equal(got_g, got), (sampler, mode)
        g.set_gemm_precision("fp32")


def test_chunk_replays_are_bitwise_reproducible(oracle_375m):
    """Soak: the captured 375M Heun chunk (39 forwards x 52 launches: stream-K GEMMs with in-launch hand-offs, the split attention,
    the sampler updates) replayed 25 times from the same noise and conditioning, in the split-bf16 and the single-term mode, with
    another workload's chunks in between (different noise -> different data in every workspace and LDS region): every replay of
    the first workload returns the same bits.  A race, a stale-LDS read or an un-re-armed counter shows up here as a flipped bit."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    g = to_hip(oracle_375m, dict(CONFIG_375M))
    known, ts = synth_known(1).cuda(), torch.tensor([2018010100]).cuda()
    other = (synth_known(1) * 1.7 + 0.3).cuda()
    pipe = AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler())
    for mode in ("bf16x3", "bf16"):
        g.set_gemm_precision(mode).enable_hip_graph(True)
        ref = ensemble_AR_sampler(pipe, 1, 4, 20, known_latents=known, timestamps=ts, sampler_type="edm", device="cuda")
        assert torch.isfinite(ref).all()
        for it in range(25):
            if it % 3 == 0:
                ensemble_AR_sampler(pipe, 1, 4, 20, known_latents=other, timestamps=ts, sampler_type="edm", device="cuda", member_ids=[it + 1])
            got = ensemble_AR_sampler(pipe, 1, 4, 20, known_latents=known, timestamps=ts, sampler_type="edm", device="cuda")
            assert torch.equal(got, ref), (mode, it)
        g.enable_hip_graph(False)
    g.set_gemm_precision("fp32")


def test_ten_chunk_chain_bf16x3_error_growth():
    """BASELINE configs[2]'s share of one GPU in shape: 2 members x 40 lead steps = 10 chained chunks of R = 4 (20 Heun steps each,
    390 forwards per member), tiny widths so the CPU oracle finishes in about a minute.  Each chunk starts from the previous
    chunk's last frame, so arithmetic differences are fed back ten times.  Stated bound: every chunk's frames stay within 1e-4
    rel-L2 of the oracle and the error does not grow faster than linearly in the chunk index (chunk c <= (c + 1) x 3e-5)."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    o = make_ar(cfg)
    g = to_hip(o, cfg)
    g_ = torch.Generator().manual_seed(9)
    mu, sd = torch.randn(84, generator=g_) * 0.2, torch.rand(84, generator=g_) + 0.5
    targs = {"mean": mu.tolist(), "std": sd.tolist(), "target_std": 0.5}
    ic = synth_known(1)[0] * 2 * sd[:, None, None, None] + mu[:, None, None, None]

    class FakeAE:
        device = torch.device("cpu")
        config = type("c", (), dict(latent_channels=84, out_channels=89, static_channels=5))

        def encode(self, x, static_conditioning_tensor=None):
            return type("o", (), dict(latent=ic.permute(1, 0, 2, 3)))

    t0 = [datetime(2018, 1, 1, 0)]
    kw = dict(ensemble_size=2, num_inference_steps=20, return_seq_len=4, latent_transform_args=targs, total_lead_time_hour=240, sampler_type="edm",
              return_latent=True)
    tc = time.perf_counter()
    with oracle_threads(16):
        want = OP.roll_out_serial(lambda t: torch.zeros(84, 1, 120, 240), t0, OP.AutoRegressive2DPipeline(o, OracleScheduler()), encdec_model=FakeAE(),
                                  static_tensor4encdec=torch.zeros(5, 120, 240), **kw)
    tc = time.perf_counter() - tc
    assert want.shape == (1, 2, 84, 41, 15, 30)
    for mode in ("bf16x3", "fp32"):
        g.set_gemm_precision(mode).enable_hip_graph(True)
        got = roll_out_serial(None, t0, AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), known_latents_override=ic, **kw)
        g.enable_hip_graph(False)
        assert got.shape == want.shape and not torch.isnan(got).any()
        per_chunk = [rel_l2(got[:, :, :, 1 + 4 * c : 5 + 4 * c], want[:, :, :, 1 + 4 * c : 5 + 4 * c]) for c in range(10)]
        print(f"\n10-chunk chain, 2 members x 40 lead steps (oracle {tc:.0f} s) [{mode}] per-chunk rel-L2: {_fmt(per_chunk)}")
        assert max(per_chunk) < TOL, (mode, per_chunk)
        for c, e in enumerate(per_chunk):
            assert e <= (c + 1) * 3e-5, (mode, c, e)
    g.set_gemm_precision("fp32")


def _chained_oracle(name, o, fake_ae, t0, kw, dense=False):
    """the oracle's chained roll_out_serial result: the committed run of tests/golden/make_fullsize_golden.py (same seeds, same stand-in encoder)
    when the file is there, else computed here (minutes of CPU).  `dense`: a file kept as every 2nd value is spread back over a NaN tensor, so
    that slices compare on the kept half (rel_l2 below ignores NaN positions of `want`)."""
    import os

    from tests.synth import Sub, load_fullsize_golden

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    if os.path.exists(path):
        g = load_fullsize_golden(path)
        w = g["want"]
        if isinstance(w, Sub):
            full = torch.full((int(torch.tensor(w.shape).prod()),), float("nan"))
            full[:: w.stride] = w.values
            w = full.reshape(w.shape)
        return w, float(g["seconds"][0])
    tc = time.perf_counter()
    want = OP.roll_out_serial(lambda t: torch.zeros(84, 1, 120, 240), t0, OP.AutoRegressive2DPipeline(o, OracleScheduler()), encdec_model=fake_ae,
                              static_tensor4encdec=torch.zeros(5, 120, 240), **kw)
    return want, time.perf_counter() - tc


def _rel_kept(got, want):
    """rel-L2 over the positions `want` holds (NaN = not kept in the committed file)"""
    keep = ~torch.isnan(want)
    return rel_l2(got[keep], want[keep])


def test_1_6b_heun_step_truncated_chunk():
    """BASELINE configs[3] in shape: the 1.6B model, 10 lead steps at R = 4 means the LAST chunk is truncated; here 6 lead steps =
    one full chunk + one chunk cut to 2 frames, 3 solver steps per chunk (5 forwards each: Euler + Heun correction twice, then the final
    Euler step - VERDICT r02 weak 3 asked for a 1.6B chunk longer than 3 forwards; 7 forwards each measured the same errors,
    profiles/r03_w_1p6B_chunks_7_forwards.log, at 100 s more oracle time), both modes."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from oracle.ar_model import CONFIG_1_6B

    o = make_ar(dict(CONFIG_1_6B))
    g = to_hip(o, dict(CONFIG_1_6B))
    targs = {"mean": [0.1] * 84, "std": [1.3] * 84, "target_std": 0.5}
    ic = synth_known(1)[0] * 2.6 + 0.1

    class FakeAE:
        device = torch.device("cpu")
        config = type("c", (), dict(latent_channels=84, out_channels=89, static_channels=5))

        def encode(self, x, static_conditioning_tensor=None):
            return type("o", (), dict(latent=ic.permute(1, 0, 2, 3)))

    t0 = [datetime(2018, 1, 1, 0)]
    kw = dict(ensemble_size=1, num_inference_steps=3, return_seq_len=4, latent_transform_args=targs, total_lead_time_hour=36, sampler_type="edm",
              return_latent=True)
    want, tc = _chained_oracle("fullsize_1p6b_truncated_chunks.npz", o, FakeAE(), t0, kw)
    del o
    assert tuple(want.shape) == (1, 1, 84, 7, 15, 30) and not torch.isnan(want).any()
    for mode in ("fp32", "bf16x3"):
        g.set_gemm_precision(mode)
        got = roll_out_serial(None, t0, AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), known_latents_override=ic, **kw)
        assert got.shape == want.shape and not torch.isnan(got).any()
        e1, e2 = rel_l2(got[:, :, :, 1:5], want[:, :, :, 1:5]), rel_l2(got[:, :, :, 5:7], want[:, :, :, 5:7])
        print(f"\n1.6B, 6 lead steps = chunk + truncated chunk, 5 forwards each (oracle {tc:.0f} s) [{mode}]: rel-L2 {e1:.2e} / {e2:.2e}")
        assert e1 < TOL and e2 < TOL, (mode, e1, e2)


def test_375m_two_members_three_chained_full_size_chunks():
    """BASELINE configs[2] in shape AT FULL WIDTH (the 375M at B = 2 was one forward, the multi-chunk chain ran at tiny
    width): the 375M model, 2 members, 12 lead steps = three chained R = 4 chunks - every chunk starts from the previous chunk's last frame of
    its member - at 3 solver steps per chunk (5 forwards each; the 39-forward chunk is test_full_375m_chunk_matches_oracle), both modes, per
    chunk against the CPU oracle.  The conditioning batch of a chunk is 3 noise levels x 2 members here."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    o = make_ar(dict(CONFIG_375M))
    g = to_hip(o, dict(CONFIG_375M))
    targs = {"mean": [0.1] * 84, "std": [1.3] * 84, "target_std": 0.5}
    ic = synth_known(1)[0] * 2.6 + 0.1

    class FakeAE:
        device = torch.device("cpu")
        config = type("c", (), dict(latent_channels=84, out_channels=89, static_channels=5))

        def encode(self, x, static_conditioning_tensor=None):
            return type("o", (), dict(latent=ic.permute(1, 0, 2, 3)))

    t0 = [datetime(2018, 1, 1, 0)]
    kw = dict(ensemble_size=2, num_inference_steps=3, return_seq_len=4, latent_transform_args=targs, total_lead_time_hour=72, sampler_type="edm",
              return_latent=True)
    want, tc = _chained_oracle("fullsize_375m_2members_3chunks.npz", o, FakeAE(), t0, kw, dense=True)
    del o
    assert tuple(want.shape) == (1, 2, 84, 13, 15, 30)
    nn_ = lambda t: torch.nan_to_num(t)  # noqa: E731  (positions the committed file does not keep count as 0 on both sides)
    assert rel_l2(nn_(want[:, 0]), nn_(want[:, 1])) > 1e-2  # the two members really differ (their own noise)
    d01, d12 = rel_l2(nn_(want[:, :, :, 1:5]), nn_(want[:, :, :, 5:9])), rel_l2(nn_(want[:, :, :, 5:9]), nn_(want[:, :, :, 9:13]))
    print(f"\nchunk-to-chunk change of the oracle's frames: {d01:.2e} {d12:.2e}")
    assert d01 > 1e-3 and d12 > 1e-3  # ... and so do consecutive chunks
    for mode in ("fp32", "bf16x3"):
        g.set_gemm_precision(mode).enable_hip_graph(True)
        got = roll_out_serial(None, t0, AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), known_latents_override=ic, **kw)
        g.enable_hip_graph(False)
        assert got.shape == want.shape and not torch.isnan(got).any()
        per_chunk = [_rel_kept(got[:, :, :, 1 + 4 * c : 5 + 4 * c], want[:, :, :, 1 + 4 * c : 5 + 4 * c]) for c in range(3)]
        print(f"\n375M, 2 members x 12 lead steps = 3 chained chunks, 5 forwards each (oracle {tc:.0f} s) [{mode}] per-chunk rel-L2: {_fmt(per_chunk)}")
        assert max(per_chunk) < TOL, (mode, per_chunk)
    g.set_gemm_precision("fp32")


def _literal_chunk_vs_golden(golden_dir, name, cfg, members, label):
    """one 20-step Heun chunk (39 forwards) at literal width against the committed oracle run (tests/golden/make_fullsize_golden.py), fp32 and
    split-bf16, with the per-evaluation error curves; eager (recorded) and graph-replayed samples must agree bit for bit"""
    import os

    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from tests.synth import load_fullsize_golden

    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated (python tests/golden/make_fullsize_golden.py)")
    fx = load_fullsize_golden(path)
    assert len(fx["outs"]) == 39 and tuple(fx["want"].shape) == (members, 84, 4, 15, 30)
    o = make_ar(dict(cfg))  # the same seeded weights the golden run used
    g = to_hip(o, dict(cfg))
    del o
    known, ts = synth_known(1).cuda(), torch.tensor([2018010100]).cuda()
    for mode in ("fp32", "bf16x3"):
        g.set_gemm_precision(mode)
        rg = Rec(g)
        got = ensemble_AR_sampler(AutoRegressive2DPipeline(rg, EDMDPMSolverMultistepScheduler()), members, 4, 20, known_latents=known, timestamps=ts,
                                  sampler_type="edm", device="cuda")
        assert len(rg.outs) == 39
        e_in, e_out, e = _curve(rg.ins, fx["ins"]), _curve(rg.outs, fx["outs"]), rel_l2(got.cpu(), fx["want"])
        print(f"\n{label}: {members} member(s), 20-step Heun chunk, 39 forwards (oracle {float(fx['seconds'][0]):.0f} s in the build container) [{mode}]: sample rel-L2 {e:.2e}")
        print(f"  network-input  error per evaluation: {_fmt(e_in)}")
        print(f"  network-output error per evaluation: {_fmt(e_out)}")
        assert e < TOL and max(e_in) < TOL and max(e_out) < TOL, (mode, e, max(e_in), max(e_out))
        g.enable_hip_graph(True)
        got_g = ensemble_AR_sampler(AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), members, 4, 20, known_latents=known, timestamps=ts,
                                    sampler_type="edm", device="cuda")
        g.enable_hip_graph(False)
        assert torch.equal(got_g, got), mode
    g.set_gemm_precision("fp32")


def test_1_6b_full_length_heun_chunk_vs_committed_oracle(golden_dir):
    """VERDICT r04 item 4 / missing 4: the 1.6B model at the reference's default chunk length (pipelines/edm_sampler.py:60-113: 20 steps, 39
    network evaluations), 1 member - until round 4 the 1.6B was compared at 5 forwards per chunk only."""
    from oracle.ar_model import CONFIG_1_6B

    _literal_chunk_vs_golden(golden_dir, "fullsize_1p6b_chunk.npz", CONFIG_1_6B, 1, "1.6B")


def test_375m_two_members_full_length_heun_chunk_vs_committed_oracle(golden_dir):
    """the 375M at batch 2 (cfg 3's share of one GPU) at the literal 20 solver steps: 39 forwards of a 2-member batch"""
    _literal_chunk_vs_golden(golden_dir, "fullsize_375m_2members.npz", CONFIG_375M, 2, "375M")


def _literal_chain_vs_golden(golden_dir, name, cfg, members, lead_hours, label, frames_per_chunk):
    """a chained roll_out_serial at the reference's LITERAL chunk length (20 solver steps = 39 forwards per chunk, pipelines/edm_sampler.py:60-113; chunks
    chained through each member's own last frame, pipelines/utils.py:533-563) against the committed oracle run (tests/golden/make_fullsize_golden.py
    literal_chain: ~13 - 26 min of CPU in the build container, < 3 s per mode on the GPU), fp32 and split-bf16, graph-replayed as the product runs it,
    per-chunk rel-L2 printed; eager launches must give the same bits"""
    import os

    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from tests.synth import load_fullsize_golden

    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated (python tests/golden/make_fullsize_golden.py literal_chain)")
    fx = load_fullsize_golden(path)
    want = fx["want"]
    total = lead_hours // 6
    assert tuple(want.shape) == (1, members, 84, 1 + total, 15, 30) and not torch.isnan(want).any()
    o = make_ar(dict(cfg))  # the same seeded weights the golden run used
    g = to_hip(o, dict(cfg))
    del o
    targs = {"mean": [0.1] * 84, "std": [1.3] * 84, "target_std": 0.5}
    ic = synth_known(1)[0] * 2.6 + 0.1
    t0 = [datetime(2018, 1, 1, 0)]
    kw = dict(ensemble_size=members, num_inference_steps=20, return_seq_len=4, latent_transform_args=targs, total_lead_time_hour=lead_hours, sampler_type="edm",
              return_latent=True)
    bounds = [(1 + 4 * c, min(5 + 4 * c, 1 + total)) for c in range(-(-total // 4))]
    assert [b - a for a, b in bounds] == frames_per_chunk
    if members > 1:
        assert rel_l2(want[:, 0], want[:, 1]) > 1e-2  # the members really differ (their own noise)
    for mode in ("fp32", "bf16x3"):
        g.set_gemm_precision(mode).enable_hip_graph(True)
        got = roll_out_serial(None, t0, AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), known_latents_override=ic, **kw)
        g.enable_hip_graph(False)
        assert got.shape == want.shape and not torch.isnan(got).any() and torch.equal(got[:, :, :, 0], want[:, :, :, 0])  # slot 0: the IC latent
        per_chunk = [rel_l2(got[:, :, :, a:b], want[:, :, :, a:b]) for a, b in bounds]
        print(f"\n{label}: {members} member(s) x {total} lead steps = {len(bounds)} chained chunks x 39 forwards (oracle {float(fx['seconds'][0]):.0f} s in the build "
              f"container) [{mode}] per-chunk rel-L2: {_fmt(per_chunk)}")
        assert max(per_chunk) < TOL, (mode, per_chunk)
        if mode == "fp32":  # eager launches = the graph-replayed chunks, bit for bit (first chunk suffices: the chain feeds on it)
            eager = roll_out_serial(None, t0, AutoRegressive2DPipeline(g, EDMDPMSolverMultistepScheduler()), known_latents_override=ic, **dict(kw, total_lead_time_hour=24))
            assert torch.equal(eager[:, :, :, 1:5], got[:, :, :, 1:5])
    g.set_gemm_precision("fp32")


def test_cfg4_literal_1_6b_ten_lead_steps_three_chunks_vs_committed_oracle(golden_dir):
    """BASELINE configs[3] AS WRITTEN, one GPU's share (VERDICT r05 item 4 / missing 2): the 1.6B model, 1 member, 20 solver steps, 10 lead steps = two
    full chunks and one cut to 2 frames - 117 network evaluations chained through the member's last frame"""
    from oracle.ar_model import CONFIG_1_6B

    _literal_chain_vs_golden(golden_dir, "fullsize_1p6b_literal_chain.npz", CONFIG_1_6B, 1, 60, "cfg 4 literal (1.6B)", [4, 4, 2])


def test_cfg3_literal_375m_two_members_three_chained_chunks_vs_committed_oracle(golden_dir):
    """BASELINE configs[2] as written, one GPU's share (2 of the 16 members), the first 3 of its 10 chunks at the literal 20 solver steps: 2 members x 3
    chunks x 39 forwards, every chunk conditioned on its member's previous last frame"""
    _literal_chain_vs_golden(golden_dir, "fullsize_375m_2members_literal_chain.npz", CONFIG_375M, 2, 72, "cfg 3 literal (375M)", [4, 4, 4])
