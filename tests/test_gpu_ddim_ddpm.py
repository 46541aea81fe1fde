"""`ladcast_amd.schedulers.DDIMScheduler` / `DDPMScheduler` (the scheduler classes the reference's pipeline loop names,
pipelines/pipeline_AR.py:19-21,85-102) on the GPU: the fused step kernels against the oracle's op-by-op restatement of diffusers
0.32.1 - BIT-EXACT, as for the EDM class (`test_scheduler_indexing_is_bit_exact`) - and both classes through
`AutoRegressive2DPipeline.__call__` on the HIP model (eager, and for DDIM the captured whole loop) against the oracle pipeline."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipelines as OP  # noqa: E402
from oracle import scheduler as OS  # noqa: E402
from tests.synth import make_ar, oracle_threads, rel_l2, synth_known, tiny_ar_config  # noqa: E402


def _gens(n, base=0):
    return [torch.Generator().manual_seed(base + k) for k in range(n)]


@pytest.mark.parametrize("pred", ["epsilon", "sample", "v_prediction"])
@pytest.mark.parametrize("clip", [True, False])
def test_ddim_step_is_bit_exact(pred, clip):
    from ladcast_amd.schedulers import DDIMScheduler

    for kw in (dict(), dict(beta_schedule="scaled_linear", timestep_spacing="trailing", set_alpha_to_one=False),
               dict(beta_schedule="squaredcos_cap_v2", timestep_spacing="linspace", clip_sample_range=0.7)):
        kw = dict(kw, prediction_type=pred, clip_sample=clip)
        a, b = DDIMScheduler(**kw), OS.DDIMScheduler(**kw)
        assert torch.equal(a.alphas_cumprod, b.alphas_cumprod)
        a.set_timesteps(20), b.set_timesteps(20)
        assert torch.equal(a.timesteps, b.timesteps) and a.timesteps.dtype == torch.int64
        x = torch.randn(2, 84, 2, 15, 30, generator=torch.Generator().manual_seed(0))
        xa, xb = x.cuda(), x.clone()
        for i, t in enumerate(b.timesteps):
            assert a.scale_model_input(xa, t) is xa
            f = torch.tanh(1.3 * xb) + 0.1
            step_kw = [dict(), dict(use_clipped_model_output=True), dict(eta=0.6), dict(eta=1.0, use_clipped_model_output=True)][i % 4]
            ga, gb = (_gens(2, 10 * i), _gens(2, 10 * i)) if "eta" in step_kw else (None, None)
            ra = a.step(f.cuda(), t.expand(2).cuda(), xa, generator=ga, **step_kw)
            rb = b.step(f, t.expand(2), xb, generator=gb, **step_kw)
            assert torch.equal(ra.prev_sample.cpu(), rb.prev_sample), (kw, i)
            assert torch.equal(ra.pred_original_sample.cpu(), rb.pred_original_sample), (kw, i)
            xa, xb = ra.prev_sample, rb.prev_sample
        assert torch.isfinite(xb).all()
    noise = torch.randn(2, 84, 2, 15, 30, generator=torch.Generator().manual_seed(5))
    ra = a.step(f.cuda(), 300, x.cuda(), eta=0.5, variance_noise=noise.cuda(), return_dict=False)
    rb = b.step(f, 300, x, eta=0.5, variance_noise=noise, return_dict=False)
    assert torch.equal(ra[0].cpu(), rb[0]) and torch.equal(ra[1].cpu(), rb[1])
    with pytest.raises(ValueError):
        a.step(f.cuda(), 300, x.cuda(), eta=0.5, variance_noise=noise.cuda(), generator=_gens(2))
    with pytest.raises(RuntimeError):
        a.step(f, 300, x)  # host tensors: there is no CPU path


@pytest.mark.parametrize("variance_type", ["fixed_small", "fixed_small_log", "fixed_large"])
@pytest.mark.parametrize("pred", ["epsilon", "sample", "v_prediction"])
def test_ddpm_step_is_bit_exact(variance_type, pred):
    from ladcast_amd.schedulers import DDPMScheduler

    for kw in (dict(), dict(beta_schedule="scaled_linear", timestep_spacing="trailing", clip_sample=False)):
        kw = dict(kw, prediction_type=pred, variance_type=variance_type)
        a, b = DDPMScheduler(**kw), OS.DDPMScheduler(**kw)
        a.set_timesteps(12), b.set_timesteps(12)
        assert torch.equal(a.timesteps, b.timesteps)
        x = torch.randn(3, 84, 1, 15, 30, generator=torch.Generator().manual_seed(1))
        xa, xb = x.cuda(), x.clone()
        for i, t in enumerate(b.timesteps):
            f = torch.sin(xb) * 0.8
            ra = a.step(f.cuda(), t.expand(3).cuda(), xa, generator=_gens(3, 7 * i))
            rb = b.step(f, t.expand(3), xb, generator=_gens(3, 7 * i))
            assert torch.equal(ra.prev_sample.cpu(), rb.prev_sample), (kw, i)
            assert torch.equal(ra.pred_original_sample.cpu(), rb.pred_original_sample), (kw, i)
            xa, xb = ra.prev_sample, rb.prev_sample
        assert int(b.timesteps[-1]) == 0 or kw.get("timestep_spacing") == "trailing"
        # one generator for the whole batch (diffusers' other randn_tensor rule)
        ra = a.step(f.cuda(), int(b.timesteps[3]), x.cuda(), generator=torch.Generator().manual_seed(3))
        rb = b.step(f, int(b.timesteps[3]), x, generator=torch.Generator().manual_seed(3))
        assert torch.equal(ra.prev_sample.cpu(), rb.prev_sample)


def test_add_noise_matches_the_definition():
    from ladcast_amd.schedulers import DDIMScheduler, DDPMScheduler

    x0 = torch.randn(3, 84, 1, 15, 30, generator=torch.Generator().manual_seed(2))
    n = torch.randn(3, 84, 1, 15, 30, generator=torch.Generator().manual_seed(3))
    ts = torch.tensor([10, 500, 999])
    for cls in (DDIMScheduler, DDPMScheduler):
        s = cls()
        got = s.add_noise(x0.cuda(), n.cuda(), ts.cuda()).cpu()
        ac = s.alphas_cumprod[ts].view(3, 1, 1, 1, 1)
        want = ac**0.5 * x0 + (1 - ac) ** 0.5 * n
        assert torch.equal(got, want)
        assert s.init_noise_sigma == 1.0 and len(s) == 1000


@pytest.fixture(scope="module")
def tiny_pair():
    from ladcast_amd.models import LaDCastTransformer3DModel

    cfg = tiny_ar_config(heads=2, layers=1, single=1, refiner=1)
    o = make_ar(cfg)
    m = LaDCastTransformer3DModel.from_config(cfg)
    m.load_state_dict(o.state_dict(), strict=True)
    return o, m.to("cuda").eval()


def test_ddim_and_ddpm_through_the_pipeline_loop(tiny_pair):
    """BASELINE's "20-step DDIM" literally, at tiny width: `AutoRegressive2DPipeline.__call__` with the product's DDIMScheduler on the
    HIP model against the oracle pipeline with the oracle's - eager, and as ONE captured hipGraph of the whole loop (bit-identical
    to eager); the same with eta > 0 and with DDPM (noise per step from the step kwargs' generators: eager launches)."""
    from ladcast_amd.pipelines import AutoRegressive2DPipeline
    from ladcast_amd.schedulers import DDIMScheduler, DDPMScheduler

    o, g = tiny_pair
    known, ts = synth_known(2), torch.tensor([2018010106])
    call = dict(batch_size=2, return_seq_len=4, timestamps=None, num_inference_steps=20, return_dict=False)
    cases = [("ddim (diffusers defaults)", DDIMScheduler, OS.DDIMScheduler, dict(), {}),
             ("ddim eta=0.5, no clipping", DDIMScheduler, OS.DDIMScheduler, dict(clip_sample=False), dict(eta=0.5)),
             ("ddpm (diffusers defaults)", DDPMScheduler, OS.DDPMScheduler, dict(), {})]
    for name, P, O, kw, step_kw in cases:
        noisy = bool(step_kw) or P is DDPMScheduler
        mk = lambda: dict(step_kw, generator=_gens(2, 100)) if noisy else dict(step_kw)  # noqa: E731
        with oracle_threads():
            want = OP.AutoRegressive2DPipeline(o, O(**kw), scheduler_step_kwargs=mk())(known_latents=known, generator=_gens(2), **dict(call, timestamps=ts))[0]
        outs = []
        for graphs in (False, True):
            g.enable_hip_graph(graphs)
            pipe = AutoRegressive2DPipeline(g, P(**kw), scheduler_step_kwargs=mk())
            outs.append(pipe(known_latents=known.cuda(), generator=_gens(2), **dict(call, timestamps=ts.cuda()))[0])
            if graphs and not noisy:  # the whole loop was captured once: one graph entry keyed on the scheduler's signature
                assert any(isinstance(k, tuple) and k and k[0] == "pipeline_loop" for k in g._graphs)
        g.enable_hip_graph(False)
        assert torch.equal(outs[0], outs[1]), name
        e = rel_l2(outs[0].cpu(), want)
        print(f"\n{name}, 20 steps, tiny AR model through AutoRegressive2DPipeline: rel-L2 vs the oracle pipeline {e:.2e}")
        assert torch.isfinite(want).all() and e < 1e-4, (name, e)
