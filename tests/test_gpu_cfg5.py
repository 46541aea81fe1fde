"""BASELINE configs[4]: "DCAE full encode -> AR (375M, 20 steps) -> decode end-to-end ... fp16/bf16 mixed", in the mode it names.

Both models run `set_gemm_precision("bf16")` (one bf16 MFMA per product; ladcast_amd/precision.py) and the WHOLE chain - DCAE encode
of the initial condition, chained sampler chunks, DCAE decode of every lead step - is compared with the fp32 CPU oracle, per chunk.

Like-for-like comparator: the reference's own mixed precision is `torch.autocast` around its fp32-weight models with three fp32
islands (models/LaDCast_3D_model.py:953, models/DCAE.py:162,180).  `oracle/autocast.py` runs the ORACLE that way; every stage of the
HIP path must be at least as close to the fp32 oracle as the autocast oracle is (it keeps the residual stream, the norms and the
GEMM outputs in fp32, which autocast does not), and inside the mode's stated per-stage tolerance (measured x 2).
"""
from datetime import datetime

import pytest
import torch

pytestmark = pytest.mark.gpu

from ladcast_amd.precision import tolerance  # noqa: E402
from oracle import autocast as OA  # noqa: E402
from oracle import pipelines as OP  # noqa: E402
from oracle.dcae import CONFIG_DCAE_84  # noqa: E402
from oracle.scheduler import EDMDPMSolverMultistepScheduler as OracleScheduler  # noqa: E402
from tests.synth import make_ar, make_dcae, oracle_threads, rel_l2, synth_field, tiny_ar_config  # noqa: E402


def _hip_ar(o, cfg):
    from ladcast_amd.models import LaDCastTransformer3DModel

    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(o.state_dict(), strict=True)
    return m.to("cuda").eval()


def _hip_ae(o, cfg):
    from ladcast_amd.models import AutoencoderDC

    m = AutoencoderDC.from_config(cfg)
    m.load_state_dict(o.state_dict(), strict=True)
    return m.cuda().eval()


def _fmt(c):
    return " ".join(f"{v:.1e}" for v in c)


def test_cfg5_end_to_end_three_chunks_bf16_mixed_vs_oracle_and_autocast():
    """encode -> 3 chained 10-step Heun chunks (R = 4, 12 lead steps, 2 members) -> decode of all 12 frames; tiny widths so that the
    fp32 oracle and the two autocast runs of the oracle finish in about a minute."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    ae_cfg = dict(CONFIG_DCAE_84, encoder_block_out_channels=(84, 84, 84, 168), decoder_block_out_channels=(84, 84, 84, 168),
                  encoder_layers_per_block=(1, 1, 1, 1), decoder_layers_per_block=(1, 1, 1, 1))
    oae = make_dcae(ae_cfg)
    gae = _hip_ae(oae, ae_cfg)
    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    oar = make_ar(cfg)
    gar = _hip_ar(oar, cfg)
    field = synth_field(84, 1, 120, 240)
    static = synth_field(1, 5, 120, 240, seed=1)[0]
    g_ = torch.Generator().manual_seed(9)
    mu, sd = torch.randn(84, generator=g_) * 0.2, torch.rand(84, generator=g_) + 0.5
    targs = {"mean": mu.tolist(), "std": sd.tolist(), "target_std": 0.5}
    fmu, fsd = torch.randn(84, generator=g_), torch.rand(84, generator=g_) + 0.5
    t0 = [datetime(2018, 1, 1, 0)]
    R, chunks = 4, 3
    kw = dict(ensemble_size=2, num_inference_steps=10, return_seq_len=R, static_tensor4encdec=static, latent_transform_args=targs,
              total_lead_time_hour=6 * R * chunks, sampler_type="edm", encdec_model_type="ae")

    def oracle_run(return_latent):
        return OP.roll_out_serial(lambda t: field, t0, OP.AutoRegressive2DPipeline(oar, OracleScheduler()), mean_tensor=fmu, std_tensor=fsd,
                                  encdec_model=oae, return_latent=return_latent, **kw)

    def per_chunk(got, want):
        return [rel_l2(got[:, :, :, 1 + R * c : 1 + R * (c + 1)].float(), want[:, :, :, 1 + R * c : 1 + R * (c + 1)]) for c in range(chunks)]

    with oracle_threads(16):
        want = oracle_run(False)
        want_lat = oracle_run(True)
        assert want.shape == (1, 2, 84, 1 + R * chunks, 120, 240) and not torch.isnan(want[:, :, :, 1:]).any()
        # policy "cuda": the cast lists of the reference's device; the literal CPU lists ("cpu") give the same figures to 3 digits
        # (profiles/r03_a_gpu_tests.log)
        with OA.reference_autocast("cuda"):
            best_dec, best_lat = per_chunk(oracle_run(False), want), per_chunk(oracle_run(True), want_lat)
    print(f"\ncfg5 tiny, oracle under autocast vs fp32 oracle, per chunk: decoded {_fmt(best_dec)} | latent {_fmt(best_lat)}")

    gae.set_gemm_precision("bf16")
    gar.set_gemm_precision("bf16")
    pipe = AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler())
    hip_kw = dict(kw, normalization_param_dict={"mean": fmu, "std": fsd}, encdec_model=gae)
    got = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, **hip_kw)
    got_lat = roll_out_serial(lambda t: field, t0, pipe, return_latent=True, **hip_kw)
    gar.enable_hip_graph(True)  # what bench.py times: the captured chunk
    got_graph = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, **hip_kw)
    gar.enable_hip_graph(False)
    assert torch.equal(got_graph[:, :, :, 1:], got[:, :, :, 1:])
    # round 6: `decode_batch_frames` - the chunks' latents stay in HBM and are decoded after the last chunk in large batches (32: all 12 lead
    # steps x 2 members in one decoder call; 5: two lead steps per call, six calls).  Same decoder, same latents; the conv kernels pick their schedule
    # from the launch size, so the fp32 accumulation order differs - and in THIS mode every activation is then rounded to bf16, which turns a last-bit
    # difference into an occasional 2^-9 one: the two decode orders agree at the mode's own error level (1.1e-3 measured, against 2.6e-3 to the
    # oracle), and each is held to the oracle at the stated tolerance.  Slot 0 (the raw IC field) bit for bit.  In the exact-fp32 mode the two orders
    # agree to rounding (below).
    for nb in (32, 5):
        got_def = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, decode_batch_frames=nb, **hip_kw)
        e_def, e_or = rel_l2(got_def[:, :, :, 1:], got[:, :, :, 1:]), per_chunk(got_def, want)
        print(f"cfg5 tiny [bf16 mode], decode deferred in batches of <= {nb} frames: vs per-chunk decode rel-L2 {e_def:.1e}; vs fp32 oracle per chunk {_fmt(e_or)}")
        assert got_def.shape == got.shape and torch.equal(got_def[:, :, :, 0], got[:, :, :, 0])
        assert e_def < tolerance("bf16", "rollout_decoded") and max(e_or) < tolerance("bf16", "rollout_decoded")
    mean_def = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, return_ensemble_mean=True, decode_batch_frames=32, **hip_kw)
    assert mean_def.shape == (1, 1, 84, 1 + R * chunks, 120, 240) and rel_l2(mean_def[:, 0, :, 1:], got[:, :, :, 1:].mean(dim=1)) < tolerance("bf16", "rollout_decoded")
    gae.set_gemm_precision("fp32")
    gar.set_gemm_precision("fp32")
    f_chunk = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, **hip_kw)
    f_def = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, decode_batch_frames=32, **hip_kw)
    e_f = rel_l2(f_def[:, :, :, 1:], f_chunk[:, :, :, 1:])
    print(f"cfg5 tiny [fp32 mode], decode deferred in one batch vs per-chunk decode: rel-L2 {e_f:.1e}; vs fp32 oracle per chunk {_fmt(per_chunk(f_def, want))}")
    assert e_f < 1e-5 and max(per_chunk(f_def, want)) < 1e-4
    gae.set_gemm_precision("bf16")
    gar.set_gemm_precision("bf16")
    e_dec, e_lat = per_chunk(got, want), per_chunk(got_lat, want_lat)
    e_ic = rel_l2(got_lat[:, :, :, 0], want_lat[:, :, :, 0])
    print(f"cfg5 tiny, HIP bf16 mode vs fp32 oracle: IC latent {e_ic:.1e}; per chunk: decoded {_fmt(e_dec)} | latent {_fmt(e_lat)}")
    tol = tolerance("bf16", "rollout_decoded")
    assert e_ic < tolerance("bf16", "dcae_encode")
    for c in range(chunks):
        assert 1e-5 < e_dec[c] < tol, (c, e_dec)  # lower bound: the mode is really on
        assert e_dec[c] <= best_dec[c] and e_lat[c] <= best_lat[c], (c, e_dec, best_dec, e_lat, best_lat)  # at least as close as autocast
    assert max(e_dec) <= 2.0 * e_dec[0], e_dec  # no run-away growth through the chain


def test_cfg5_full_size_single_chunk_bf16_mixed(full_dcae_oracle, fullsize_chunk_oracle):
    """cfg5's share of one GPU at LITERAL size, one member, one chunk: full DCAE encode of the 84 x 120 x 240 frame -> normalise -> 375M,
    20 Heun steps (39 forwards) -> de-normalise -> full DCAE decode of the 4 lead steps, everything in the `bf16` mode, against the
    fp32 oracle chain (shared session run), stage by stage; plus the autocast oracle for the two DCAE stages and one 375M forward."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline, roll_out_serial
    from ladcast_amd.schedulers import EDMDPMSolverMultistepScheduler

    d, fx = full_dcae_oracle, fullsize_chunk_oracle
    gae = _hip_ae(d.model, d.cfg).set_gemm_precision("bf16")
    gar = _hip_ar(fx.ar, fx.cfg).set_gemm_precision("bf16")
    t0 = [datetime(2018, 1, 1, 0)]
    kw = dict(ensemble_size=1, num_inference_steps=20, return_seq_len=4, static_tensor4encdec=d.st[0], latent_transform_args=fx.targs,
              total_lead_time_hour=24, sampler_type="edm", encdec_model_type="ae", encdec_model=gae)
    pipe = AutoRegressive2DPipeline(gar, EDMDPMSolverMultistepScheduler())
    field = d.f[0][:, None]  # (84, 1, 120, 240)
    lat = roll_out_serial(lambda t: field, t0, pipe, return_latent=True, **kw)
    dec = roll_out_serial(lambda t: field, t0, pipe, return_latent=False, **kw)
    assert lat.shape == (1, 1, 84, 5, 15, 30) and dec.shape == (1, 1, 84, 5, 120, 240)
    e_enc = rel_l2(lat[0, :, :, 0], d.z)
    e_lat = rel_l2(lat[0, :, :, 1:], fx.latents)
    e_dec = rel_l2(dec[0, :, :, 1:], fx.decoded)
    # the decoder alone, on the oracle's latents
    y = gae.decode(fx.latents[0].permute(1, 0, 2, 3).contiguous().cuda()).sample
    e_dec_only = rel_l2(y.cpu().permute(1, 0, 2, 3)[None], fx.decoded)  # (frames, C, H, W) -> the fixture's (1, C, frames, H, W)
    with torch.no_grad(), OA.reference_autocast("cuda"):
        a_enc = rel_l2(d.model.encode(d.f, static_conditioning_tensor=d.st).latent.float(), d.z)
        a_dec = rel_l2(d.model.decode(d.z).sample.float(), d.y)
    w_fwd = fx.fwd_in20  # the fp32 oracle's forward of network input 20 at t = 0.3 (session fixture / committed golden)
    with torch.no_grad(), OA.reference_autocast("cuda"):
        a_fwd = rel_l2(fx.ar(fx.in20, torch.tensor([0.3]), fx.known, time_elapsed=fx.ts).sample.float(), w_fwd)
    g_fwd = rel_l2(gar(fx.in20.cuda(), torch.tensor([0.3]).cuda(), fx.known.cuda(), time_elapsed=fx.ts.cuda()).sample.cpu(), w_fwd)
    e_dec1 = rel_l2(gae.decode(d.z.cuda()).sample.cpu(), d.y)
    print(f"\ncfg5 full size, 1 member, 1 chunk [bf16 mode] vs fp32 oracle: encode {e_enc:.2e} -> chunk latents {e_lat:.2e} -> decoded fields {e_dec:.2e}"
          f" (decoder alone {e_dec_only:.2e})")
    print(f"  autocast oracle vs fp32 oracle: encode {a_enc:.2e}, decode {a_dec:.2e}, one 375M forward {a_fwd:.2e};"
          f" HIP bf16: encode {e_enc:.2e}, decode {e_dec1:.2e}, the same forward {g_fwd:.2e}")
    assert e_enc < tolerance("bf16", "dcae_encode") and e_dec_only < tolerance("bf16", "dcae_decode")
    assert e_dec < tolerance("bf16", "rollout_decoded")
    assert g_fwd < tolerance("bf16", "forward")
    assert e_enc <= a_enc and e_dec1 <= a_dec and g_fwd <= a_fwd, (e_enc, a_enc, e_dec1, a_dec, g_fwd, a_fwd)
