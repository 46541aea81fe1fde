"""The PRODUCT's sampler path (HIP state-update kernels, scheduler, ensemble driver) against outputs of the reference's own
sampler code (tests/golden/sampler_ref.npz, see tests/test_oracle_reference_pins.py): the same elementwise toy network stands in
for the transformer on both sides, so what is compared is exactly pipelines/edm_sampler.py:60-113, pipelines/pipeline_AR.py:77-102
and pipelines/utils.py:682-741 as executed by the reference vs by ladcast_amd on the GPU.  Tolerance 1e-6 rel-L2 (the toy network's
torch ops run on the GPU here and on the CPU there; the fp64 / fp32 update kernels themselves are bit-exact, test_gpu_ops.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.synth import ToyNet  # noqa: E402


def _gens(n):
    return [torch.Generator("cpu").manual_seed(k) for k in range(n)]


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.double()
    return ((a - b).norm() / b.norm()).item()


def test_product_samplers_reproduce_the_reference_code(golden_dir):
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, edm_AR_sampler, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    z = np.load(f"{golden_dir}/sampler_ref.npz")
    z = {k: torch.from_numpy(z[k]) for k in z.files}
    net = ToyNet(6, device="cuda")
    ts = torch.tensor([2018010100]).cuda()
    known1, known3 = z["known1"].cuda(), z["known3"].cuda()
    got = edm_AR_sampler(net, EDMDPMSolverMultistepScheduler(), batch_size=3, return_seq_len=2, num_inference_steps=5, known_latents=known3, timestamps=ts,
                         generator=_gens(3), device="cuda")
    assert _rel(got, z["edm_n5"]) < 1e-6
    got = edm_AR_sampler(net, EDMDPMSolverMultistepScheduler(), batch_size=1, return_seq_len=4, num_inference_steps=1, known_latents=known1, timestamps=None,
                         generator=_gens(1), device="cuda")
    assert _rel(got, z["edm_n1"]) < 1e-6
    for name, churn in (("edm_churn_lo", 1.5), ("edm_churn_cap", 40.0)):  # the stochastic-churn branch, noise from the caller's randn_like
        gc = torch.Generator("cpu").manual_seed(77)
        got = edm_AR_sampler(net, EDMDPMSolverMultistepScheduler(), batch_size=3, return_seq_len=2, num_inference_steps=5, known_latents=known3,
                             timestamps=ts, generator=_gens(3), deterministic=False, S_churn=churn, S_min=0.05, S_max=50.0, S_noise=1.003,
                             randn_like=lambda x: torch.randn(x.shape, generator=gc, dtype=x.dtype).to(x.device), device="cuda")
        assert _rel(got, z[name]) < 1e-6
    pipe = AutoRegressive2DPipeline(net, EDMDPMSolverMultistepScheduler())
    got = pipe(batch_size=3, return_seq_len=2, known_latents=known3, timestamps=ts, generator=_gens(3), num_inference_steps=6, return_dict=False)[0]
    assert _rel(got, z["pipe_n6"]) < 1e-6
    got = pipe(batch_size=1, return_seq_len=1, known_latents=known1, timestamps=ts, generator=_gens(1), num_inference_steps=20).fields
    assert _rel(got, z["pipe_n20"]) < 1e-6
    got = ensemble_AR_sampler(pipe, 5, 3, 4, known_latents=known1, timestamps=ts, batch_size=2, sampler_type="edm", device="cuda")
    assert _rel(got, z["ens_edm"]) < 1e-6
    got = ensemble_AR_sampler(pipe, 4, 2, 4, known_latents=known1, timestamps=ts, batch_size=3, sampler_type="pipeline", device="cuda")
    assert _rel(got, z["ens_pipe"]) < 1e-6


def test_product_transformer_reproduces_the_reference_forward_code(golden_dir):
    """tests/golden/ar_forward_ref.npz holds the tiny model's outputs computed by the REFERENCE's forward code (every class it defines, bound onto
    seeded parameter containers - make_golden.py::ar_forward_fixtures).  The HIP model with the same weights, fp32 and split-bf16 arithmetic,
    against those numbers directly (no oracle in between): the fp32 budget of the north star, 1e-4; measured ~3e-6 / ~5e-6."""
    from ladcast_amd.models import LaDCastTransformer3DModel
    from tests.synth import make_ar, synth_known, tiny_ar_config

    z = np.load(f"{golden_dir}/ar_forward_ref.npz")
    cfg = tiny_ar_config()
    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(make_ar(cfg).state_dict(), strict=True)
    m = m.cuda().eval()
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5)):
        m.set_gemm_precision(prec)
        for name, (B, R, Bt, stamp) in {"a": (2, 4, 1, 2018010100), "b": (1, 1, 1, 2019063012), "c": (3, 2, 3, None)}.items():
            x = torch.randn(B, 84, R, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
            te = None if stamp is None else torch.tensor([stamp]).cuda()
            y = m(x, torch.linspace(-1.2, 1.0, Bt).cuda(), synth_known(B).cuda(), time_elapsed=te).sample.double().flatten().cpu()
            want = torch.from_numpy(z[name]).double()
            assert ((y[::7] - want).norm() / want.norm()).item() < tol, (prec, name)
            assert abs(y.norm().item() / float(z[name + "_norm"]) - 1) < tol


def test_product_nope_reproduces_the_reference_forward_code(golden_dir):
    """`nope=True` (round 5; models/LaDCast_3D_model.py:710-712,897-918): temporal-only rotary tables over the whole head dimension, in both
    attention paths (fp32: qk_rmsnorm_rope kernel; split modes: the QKV GEMM epilogue's compact table), against the fixture made by the
    reference's forward code; the sampler chunk (graph) runs with it too."""
    from ladcast_amd.models import LaDCastTransformer3DModel
    from tests.synth import make_ar, synth_known, tiny_ar_config

    z = np.load(f"{golden_dir}/ar_forward_ref.npz")
    cfg = dict(tiny_ar_config(), nope=True)
    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(make_ar(tiny_ar_config()).state_dict(), strict=True)
    m = m.cuda().eval()
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    want = torch.from_numpy(z["nope"]).double()
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5), ("bf16", 5e-3)):
        m.set_gemm_precision(prec)
        y = m(x, torch.tensor([0.3]).cuda(), synth_known(2).cuda(), time_elapsed=torch.tensor([2018010100]).cuda()).sample.double().flatten().cpu()
        assert ((y[::7] - want).norm() / want.norm()).item() < tol, prec
    m.set_gemm_precision("fp32")


def test_product_patch_sizes_reproduce_the_reference_forward_code(golden_dir):
    """`patch_size` / `patch_size_t` != 1 (round 6: round 5 raised NotImplementedError; models/LaDCast_3D_model.py:657-663,758,866-871,1044-1062): the
    HIP model re-orders a patch's values into the channel axis and runs the patch grid as a patch-size-1 problem.  Against the fixtures made by the
    reference's forward code, three arithmetic modes; the graph-replayed forward and a sampler chunk (conditioning prepared once for all noise
    levels, the chunk captured as one hipGraph) run with it too."""
    from ladcast_amd.models import LaDCastTransformer3DModel
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler
    from oracle import pipelines as OP
    from oracle.scheduler import EDMDPMSolverMultistepScheduler as OracleScheduler
    from tests.synth import make_ar, tiny_ar_config

    z = np.load(f"{golden_dir}/ar_forward_ref.npz")
    for name, (p_, pt_, t_in) in {"patch3": (3, 1, 1), "patch5_t2": (5, 2, 2)}.items():
        cfg = dict(tiny_ar_config(), patch_size=p_, patch_size_t=pt_)
        o = make_ar(cfg)
        m = LaDCastTransformer3DModel.from_config(cfg)
        m.load_state_dict(o.state_dict(), strict=True)
        m = m.cuda().eval()
        x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
        known = (0.5 * torch.randn(2, 84, t_in, 15, 30, generator=torch.Generator().manual_seed(2))).cuda()
        want = torch.from_numpy(z[name]).double()
        for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5), ("bf16", 5e-3)):
            m.set_gemm_precision(prec)
            y = m(x, torch.tensor([0.3]).cuda(), known, time_elapsed=torch.tensor([2018010100]).cuda()).sample
            assert tuple(y.shape) == (2, 84, 4, 15, 30)
            e = ((y.double().flatten().cpu()[::7] - want).norm() / want.norm()).item()
            print(f"\n{name} [{prec}]: rel-L2 vs the reference-code fixture {e:.2e}")
            assert e < tol, (name, prec, e)
        m.set_gemm_precision("fp32")
        eager = m(x, torch.tensor([0.3]).cuda(), known, time_elapsed=torch.tensor([2018010100]).cuda()).sample
        m.enable_hip_graph(True)
        assert torch.equal(m(x, torch.tensor([0.3]).cuda(), known, time_elapsed=torch.tensor([2018010100]).cuda()).sample, eager)
        # a 3-step Heun chunk through the samplers (batched conditioning + chunk graph) against the oracle's sampler
        ts = torch.tensor([2018010100])
        want_s = OP.ensemble_AR_sampler(OP.AutoRegressive2DPipeline(o, OracleScheduler()), 2, 4, 3, known_latents=known[:1].cpu(), timestamps=ts, sampler_type="edm")
        got_s = ensemble_AR_sampler(AutoRegressive2DPipeline(m, EDMDPMSolverMultistepScheduler()), 2, 4, 3, known_latents=known[:1], timestamps=ts.cuda(),
                                    sampler_type="edm", device="cuda")
        es = _rel(got_s, want_s)
        print(f"{name}: 3-step Heun chunk, 2 members, graph-replayed: rel-L2 vs the oracle sampler {es:.2e}")
        assert es < 1e-4
        m.enable_hip_graph(False)


def test_product_scale_attn_by_lat_reproduces_the_reference_forward_code(golden_dir):
    """`scale_attn_by_lat=True`: the per-key score bias of every attention call (refiner: cond keys; blocks: pred + cond keys) against the
    fixtures made by the reference's forward code, with the reference's weights and with them amplified 200x; all three arithmetic modes."""
    from ladcast_amd.models import LaDCastTransformer3DModel
    from tests.synth import make_ar, synth_known, tiny_ar_config

    z = np.load(f"{golden_dir}/ar_forward_ref.npz")
    cfg = dict(tiny_ar_config(), scale_attn_by_lat=True)
    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(make_ar(tiny_ar_config()).state_dict(), strict=True)
    m = m.cuda().eval()
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3)).cuda()
    base = m.attn_lat_weights.clone()
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5), ("bf16", 5e-3)):
        m.set_gemm_precision(prec)
        for name, amp in (("lat", 1.0), ("lat200", 200.0)):
            m.set_attn_lat_weights(amp * base)
            y = m(x, torch.tensor([0.3]).cuda(), synth_known(2).cuda(), time_elapsed=torch.tensor([2018010100]).cuda()).sample.double().flatten().cpu()
            want = torch.from_numpy(z[name]).double()
            assert ((y[::7] - want).norm() / want.norm()).item() < tol, (prec, name)
    # hipGraph replay gives the same bits
    m.set_gemm_precision("bf16x3")
    y0 = m(x, torch.tensor([0.3]).cuda(), synth_known(2).cuda(), time_elapsed=torch.tensor([2018010100]).cuda()).sample
    m.enable_hip_graph(True)
    y1 = m(x, torch.tensor([0.3]).cuda(), synth_known(2).cuda(), time_elapsed=torch.tensor([2018010100]).cuda()).sample
    assert torch.equal(y0, y1)


def test_product_dcae_reproduces_the_reference_forward_code(golden_dir):
    """the same for the autoencoder: tests/golden/dcae_forward_ref.npz (reference forward code of every DCAE class incl. AutoencoderDC.encode /
    decode) vs the HIP autoencoder with the same seeded weights"""
    from ladcast_amd.models import AutoencoderDC
    from tests.synth import make_dcae, synth_field, tiny_dcae_config

    z = np.load(f"{golden_dir}/dcae_forward_ref.npz")
    cfg = tiny_dcae_config()
    g = AutoencoderDC.from_config(cfg)
    g.load_state_dict(make_dcae(cfg).state_dict(), strict=True)
    g = g.cuda().eval()
    f, st = synth_field(2, 8, 48, 64).cuda(), synth_field(1, 5, 48, 64, seed=1).cuda()
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5)):
        g.set_gemm_precision(prec)
        lat = g.encode(f, static_conditioning_tensor=st).latent
        zz = torch.from_numpy(z["z"]).cuda()
        rec = g.decode(zz, return_static=True).sample
        plain = g.decode(zz).sample
        assert _rel(lat, torch.from_numpy(z["z"])) < tol and _rel(rec, torch.from_numpy(z["y"])) < tol and _rel(plain, torch.from_numpy(z["y_nostatic"])) < tol


def test_product_dcae_without_a_full_resolution_stage_reproduces_the_reference_forward_code(golden_dir):
    """`layers_per_block[0] == 0` (round 6; models/DCAE.py:559-579,696-712 - the DC-AE family's f64 / f128 form): no stage at full resolution, the
    encoder's conv_in is a DCDownBlock2d and the decoder's conv_out a DCUpBlock2d, both WITHOUT shortcut (ABI 5: ldc_pixel_unshuffle_shortcut_split
    with x = NULL, ldc_pixel_shuffle_to_chan).  Round 5 raised NotImplementedError.  The HIP autoencoder with the oracle's seeded weights (strict
    load) against the fixture made by the REFERENCE's forward code (`z_layers0`, `y_layers0`), three arithmetic modes, eager and graph; also the
    interpolate up-sampling form of that conv_out against the oracle."""
    from ladcast_amd.models import AutoencoderDC
    from tests.synth import make_dcae, synth_field, tiny_dcae_config

    z = np.load(f"{golden_dir}/dcae_forward_ref.npz")
    cfg = dict(tiny_dcae_config(), encoder_layers_per_block=(0, 1, 1, 1), decoder_layers_per_block=(0, 1, 1, 1))
    g = AutoencoderDC.from_config(cfg)
    g.load_state_dict(make_dcae(cfg).state_dict(), strict=True)
    g = g.cuda().eval()
    f, st = synth_field(2, 8, 48, 64).cuda(), synth_field(1, 5, 48, 64, seed=1).cuda()
    z0, y0 = torch.from_numpy(z["z_layers0"]), torch.from_numpy(z["y_layers0"])
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5), ("bf16", 2e-2)):
        g.set_gemm_precision(prec)
        lat = g.encode(f, static_conditioning_tensor=st).latent
        rec = g.decode(z0.cuda(), return_static=True).sample
        plain = g.decode(z0.cuda()).sample
        e1, e2 = _rel(lat, z0), _rel(rec, y0)
        print(f"\nDC-AE without a full-resolution stage [{prec}]: encode {e1:.2e}, decode {e2:.2e}")
        assert lat.shape == z0.shape and rec.shape == y0.shape and e1 < tol and e2 < tol, (prec, e1, e2)
        assert plain.shape == (2, 8, 48, 64) and torch.equal(plain, rec[:, :8])  # the static channels are dropped by the same kernel
        g.enable_hip_graph(True)
        assert torch.equal(g.encode(f, static_conditioning_tensor=st).latent, lat) and torch.equal(g.decode(z0.cuda(), return_static=True).sample, rec)
        g.enable_hip_graph(False)
    # the interpolate form of the same conv_out (nearest x2, conv at the output width, no shortcut) against the oracle
    cfg_i = dict(cfg, upsample_block_type="interpolate")
    o = make_dcae(cfg_i)
    gi = AutoencoderDC.from_config(cfg_i)
    gi.load_state_dict(o.state_dict(), strict=True)
    gi = gi.cuda().eval()
    with torch.no_grad():
        want = o.decode(z0, return_static=True).sample
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5)):
        gi.set_gemm_precision(prec)
        assert _rel(gi.decode(z0.cuda(), return_static=True).sample, want) < tol, prec


def test_product_timestep_conditioned_dcae_reproduces_the_reference_forward_code(golden_dir):
    """`temb_channels` (round 5; models/DCAE.py:36-64,147-153,193-198,256-257,351-365,845-850,982-984,1067-1085): the timestep-conditioned ResBlock
    (scale / shift between its convs) and linear-attention block (AdaLayerNormZeroSingle4Sana before, gate after), raw timesteps through
    time_proj + timestep_embedder.  The HIP autoencoder with the oracle's seeded weights (strict load: same parameter names) against the
    fixture made by the REFERENCE's forward code (`z_temb`, `y_temb` of dcae_forward_ref.npz), three arithmetic modes, eager and graph;
    `forward(time_elapsed=...)`; an already-embedded temb; and the error for a missing temb."""
    import pytest

    from ladcast_amd.models import AutoencoderDC
    from tests.synth import make_dcae, synth_field, tiny_dcae_config

    z = np.load(f"{golden_dir}/dcae_forward_ref.npz")
    cfg = dict(tiny_dcae_config(), temb_channels=48)
    o = make_dcae(cfg)
    g = AutoencoderDC.from_config(cfg)
    g.load_state_dict(o.state_dict(), strict=True)
    g = g.cuda().eval()
    f, st = synth_field(2, 8, 48, 64).cuda(), synth_field(1, 5, 48, 64, seed=1).cuda()
    tt = torch.tensor([0.3, 1.7]).cuda()
    zt, yt = torch.from_numpy(z["z_temb"]), torch.from_numpy(z["y_temb"])
    for prec, tol in (("fp32", 2e-5), ("bf16x3", 5e-5), ("bf16", 2e-2)):
        g.set_gemm_precision(prec)
        lat = g.encode(f, temb=tt, static_conditioning_tensor=st).latent
        rec = g.decode(zt.cuda(), temb=tt, return_static=True).sample
        e1, e2 = _rel(lat, zt), _rel(rec, yt)
        print(f"\ntimestep-conditioned tiny DC-AE [{prec}]: encode {e1:.2e}, decode {e2:.2e}")
        assert e1 < tol and e2 < tol, (prec, e1, e2)
        g.enable_hip_graph(True)
        assert torch.equal(g.encode(f, temb=tt, static_conditioning_tensor=st).latent, lat) and torch.equal(g.decode(zt.cuda(), temb=tt, return_static=True).sample, rec)
        g.enable_hip_graph(False)
    g.set_gemm_precision("fp32")
    full = g(f, time_elapsed=tt, static_conditioning_tensor=st, return_static=True).sample
    assert _rel(full, yt) < 5e-5  # decode of its OWN latent: both halves' errors
    with torch.no_grad():  # an already-embedded temb (embedded_t=True), as AutoencoderDC.forward hands it to encode / decode
        from oracle.layers import get_timestep_embedding

        emb = o.timestep_embedder(get_timestep_embedding(tt.cpu(), 256)).cuda()
    assert _rel(g.encode(f, temb=emb, embedded_t=True, static_conditioning_tensor=st).latent, zt) < 2e-5
    with pytest.raises(ValueError):
        g.encode(f, static_conditioning_tensor=st)  # built with temb_channels: the blocks need it
    plain = AutoencoderDC.from_config(tiny_dcae_config()).cuda().eval()
    with pytest.raises(ValueError):
        plain.decode(torch.zeros(1, 8, 6, 8).cuda(), temb=tt[:1])  # and the unconditioned model refuses one

