"""INDEPENDENT derivations of the third-party (diffusers 0.32.1) leaves that ``oracle/layers.py`` and ``oracle/scheduler.py``
restate.  diffusers is absent here, so these are not comparisons with its code; each test derives the expected value a
different way from the oracle's own arithmetic - a torch built-in that implements the published operator, complex-number
rotation, the closed-form solution of the probability-flow ODE, or the formulas of the papers (Karras et al. 2022 "EDM",
Lu et al. 2022 "DPM-Solver++") evaluated in fp64 - so a transcription slip in the oracle cannot cancel against itself.
What still has no independent check is listed in ``oracle/__init__.py``.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import layers as L
from oracle.scheduler import EDMDPMSolverMultistepScheduler


# -- RMSNorm: torch's own operator (torch >= 2.4 ships F.rms_norm / nn.RMSNorm) ------------------------------------------
def test_rmsnorm_equals_torch_builtin():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 7, 128, generator=g) * 3.0
    m = L.RMSNorm(128, eps=1e-7)
    with torch.no_grad():
        m.weight.copy_(1.0 + 0.1 * torch.randn(128, generator=g))
    want = F.rms_norm(x, (128,), weight=m.weight, eps=1e-7)
    assert torch.allclose(m(x), want, rtol=2e-6, atol=1e-7)
    ref = torch.nn.RMSNorm(128, eps=1e-7)
    with torch.no_grad():
        ref.weight.copy_(m.weight)
    assert torch.allclose(m(x), ref(x), rtol=2e-6, atol=1e-7)
    # fp64 definition: x / sqrt(mean(x^2) + eps) * w
    xd = x.double()
    defn = xd / torch.sqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-7) * m.weight.double()
    assert (m(x).double() - defn).abs().max() < 5e-6


# -- rotary embedding: rotation of the complex number (x_even + i x_odd) by exp(i pos omega) -----------------------------
def test_rope_equals_complex_rotation():
    g = torch.Generator().manual_seed(1)
    S, dim, theta = 37, 56, 256.0
    pos = torch.randn(S, generator=g) * 4.0  # the reference feeds real-valued (radian) grid positions
    cos, sin = L.get_1d_rotary_pos_embed(dim, pos, theta)
    assert cos.shape == sin.shape == (S, dim)
    omega = torch.tensor([theta ** (-2.0 * j / dim) for j in range(dim // 2)], dtype=torch.float64)  # RoFormer eq. 15
    ang = torch.outer(pos.double(), omega)
    rot = torch.polar(torch.ones_like(ang), ang)  # exp(i pos omega)
    x = torch.randn(2, 3, S, dim, generator=g)
    xc = torch.view_as_complex(x.double().reshape(2, 3, S, dim // 2, 2).contiguous())
    want = torch.view_as_real(xc * rot[None, None]).reshape(2, 3, S, dim)
    got = L.apply_rotary_emb(x, (cos, sin))
    assert (got.double() - want).abs().max() < 3e-6
    # a rotation preserves every pair's norm, and position 0 is the identity
    pair = lambda t: t.reshape(*t.shape[:-1], -1, 2).pow(2).sum(-1)  # noqa: E731
    assert torch.allclose(pair(got), pair(x), rtol=1e-5, atol=1e-6)
    c0, s0 = L.get_1d_rotary_pos_embed(dim, torch.zeros(S), theta)
    assert torch.equal(L.apply_rotary_emb(x, (c0, s0)), x)
    # relative-position property: <R(p)q, R(p')k> depends on p - p' only
    q, k = torch.randn(1, 1, 1, dim, generator=g), torch.randn(1, 1, 1, dim, generator=g)

    def dot(p, p2):
        cq = L.get_1d_rotary_pos_embed(dim, torch.tensor([p]), theta)
        ck = L.get_1d_rotary_pos_embed(dim, torch.tensor([p2]), theta)
        return (L.apply_rotary_emb(q, cq).double() * L.apply_rotary_emb(k, ck).double()).sum().item()

    assert abs(dot(1.25, 0.5) - dot(3.0, 2.25)) < 1e-5


# -- sinusoidal timestep embedding: closed form in fp64 + the half-angle structure ----------------------------------------
def test_timestep_embedding_closed_form():
    t = torch.tensor([0.0, -1.3862944, 1.0954452, 0.25 * math.log(80.0)])
    e = L.get_timestep_embedding(t, 256)
    assert e.shape == (4, 256)
    k = np.arange(128)
    f = 10000.0 ** (-k / 128.0)  # "Attention is all you need" sec. 3.5 with max_period 1e4, no frequency shift
    arg = t.double().numpy()[:, None] * f[None]
    want = np.concatenate([np.cos(arg), np.sin(arg)], axis=1)  # flip_sin_to_cos=True: cosines first
    assert np.abs(e.double().numpy() - want).max() < 2e-6
    assert torch.equal(e[0, :128], torch.ones(128)) and torch.equal(e[0, 128:], torch.zeros(128))
    assert torch.allclose(e[:, :128] ** 2 + e[:, 128:] ** 2, torch.ones(4, 128), atol=1e-6)


# -- Karras sigma table ---------------------------------------------------------------------------------------------------
def _karras_fp64(n, smin=0.002, smax=80.0, rho=7.0):
    ramp = np.linspace(0.0, 1.0, n)  # fp64
    return (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho  # EDM eq. 5


def test_karras_table_vs_fp64_and_the_linspace_question():
    """The oracle builds the ramp with fp32 ``torch.linspace``; if diffusers 0.32.1 used an fp64 ``np.linspace`` ramp and
    rounded at the end, the fp32 table would differ by the amount measured here.  Quantified so the residual risk in
    VERDICT r01 has a number: measured 1.3e-6 relative at worst (the rho = 7 power amplifies the ramp's fp32 rounding
    seven-fold), bound 3e-6 here - a factor 30+ below the 1e-4 parity budget (a sigma perturbation of that size changes c_in / c_skip / c_out by the same relative amount)."""
    for n in (20, 50):
        s = EDMDPMSolverMultistepScheduler()
        s.set_timesteps(n)
        sig32 = s.sigmas[:-1].double().numpy()
        sig64 = _karras_fp64(n)
        rel = np.abs(sig32 - sig64) / sig64
        assert rel.max() < 3e-6, rel.max()
        ulps = np.abs(sig32 - sig64.astype(np.float32).astype(np.float64)) / np.spacing(sig64.astype(np.float32)).astype(np.float64)
        assert ulps.max() <= 32, ulps.max()
        assert abs(sig32[0] - 80.0) < 1e-4 and abs(sig32[-1] - 0.002) < 1e-8  # (80^(1/7))^7 in fp32 is 79.99998 and s.sigmas[-1].item() == 0.0
        assert np.all(np.diff(sig32) < 0)  # strictly decreasing
        # c_noise = ln(sigma) / 4 (EDM table 1)
        assert np.abs(s.timesteps.double().numpy() - 0.25 * np.log(sig64)).max() < 2e-7 * 4


def test_preconditioning_is_edm_table_1():
    s = EDMDPMSolverMultistepScheduler()
    sd = 0.5
    for sigma in (80.0, 3.7, 0.5, 0.002):
        x, f = torch.tensor([1.7]), torch.tensor([-0.3])
        c_in = 1.0 / math.sqrt(sigma**2 + sd**2)
        c_skip = sd**2 / (sigma**2 + sd**2)
        c_out = sigma * sd / math.sqrt(sigma**2 + sd**2)
        assert abs(s.precondition_inputs(x, torch.tensor(sigma)).item() - 1.7 * c_in) < 1e-6 * abs(1.7 * c_in) + 1e-9
        want = c_skip * 1.7 + c_out * -0.3
        assert abs(s.precondition_outputs(x, f, torch.tensor(sigma)).item() - want) < 2e-6 * max(abs(want), 1e-3)
        # the identity the parametrisation is built on: c_skip^2 sigma_data^2... Var target = 1  (EDM eq. 117-120)
        assert abs((1 - c_skip) ** 2 * sd**2 + c_skip**2 * sigma**2 - c_out**2) < 1e-9 * max(1.0, c_out**2)


# -- DPM-Solver++(2M), midpoint, alpha == 1: the paper's update evaluated in fp64 ----------------------------------------
def _drive(s, n, x, model):
    s.set_timesteps(n)
    xs = [x]
    for t in s.timesteps:
        xin = s.scale_model_input(x, t)
        x = s.step(model(xin, x, s.sigmas[s.step_index]), t, x, return_dict=False)[0]
        xs.append(x)
    return xs


def test_dpmpp_constant_denoiser_is_exact():
    """If the network's data prediction is a constant x0, the diffusion ODE dx/dsigma = (x - x0)/sigma has the solution
    x(sigma) = x0 + (sigma/sigma_s)(x_s - x0); every DPM-Solver++ step (first and second order: D1 = 0) must land on it and
    the last step (sigma -> 0, lower_order_final) must return x0 itself."""
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 4, generator=g, dtype=torch.float64)
    s = EDMDPMSolverMultistepScheduler()
    sd = 0.5

    def model(xin, x, sigma):  # F such that c_skip x + c_out F == x0
        sigma = sigma.double()
        c_skip = sd**2 / (sigma**2 + sd**2)
        c_out = sigma * sd / (sigma**2 + sd**2) ** 0.5
        return (x0 - c_skip * x) / c_out

    x_init = torch.randn(2, 4, generator=g, dtype=torch.float64) * 80.0
    xs = _drive(s, 20, x_init, model)
    sig = s.sigmas.double()
    for i, x in enumerate(xs):
        want = x0 + (sig[i] / sig[0]) * (x_init - x0)
        assert (x - want).abs().max() < 1e-5 * max(1.0, want.abs().max().item()), i
    assert (xs[-1] - x0).abs().max() < 1e-6  # fp32 sigma table; state carried in fp64 here


def test_dpmpp_2m_update_matches_paper_formula():
    """Lu et al. 2022, Algorithm 2 (DPM-Solver++(2M)) with alpha_t = 1, lambda = -ln sigma:
    x_i = (sigma_i/sigma_{i-1}) x_{i-1} - (e^{-h_i} - 1) D_i,  D_i = (1 + 1/(2 r_i)) x0_{i-1} - (1/(2 r_i)) x0_{i-2},
    r_i = h_{i-1}/h_i; first step and (final_sigmas_type == "zero") last step first order.  A smooth non-linear
    denoiser exercises every branch; the recurrence is re-evaluated here in fp64 from the sigma table alone."""
    g = torch.Generator().manual_seed(4)
    A = torch.randn(4, 4, generator=g, dtype=torch.float64) * 0.3
    s = EDMDPMSolverMultistepScheduler()
    sd = 0.5
    n = 20

    def x0_pred(x, sigma):
        return torch.tanh(x @ A) / (1.0 + sigma)

    def model(xin, x, sigma):
        sigma = sigma.double()
        c_skip = sd**2 / (sigma**2 + sd**2)
        c_out = sigma * sd / (sigma**2 + sd**2) ** 0.5
        return (x0_pred(x, sigma) - c_skip * x) / c_out

    x_init = torch.randn(3, 4, generator=g, dtype=torch.float64) * 80.0
    xs = _drive(s, n, x_init, model)
    sig = s.sigmas.double()
    lam = -torch.log(sig[:-1])
    x, hist = x_init, []
    for i in range(n):
        d = x0_pred(x, sig[i])
        hist.append(d)
        if i == n - 1:  # sigma_{i+1} = 0: e^{-h} = 0, sigma ratio 0
            x = d
        else:
            h = lam[i + 1] - lam[i]
            if i == 0:
                D = d
            else:
                r = (lam[i] - lam[i - 1]) / h
                D = (1 + 1 / (2 * r)) * hist[-1] - (1 / (2 * r)) * hist[-2]
            x = (sig[i + 1] / sig[i]) * x - (torch.exp(-h) - 1.0) * D
        assert (xs[i + 1] - x).abs().max() < 2e-5 * max(1.0, x.abs().max().item()), i  # the oracle divides by c_out in fp64 too; fp32 table only
    assert s.step_index == n


def test_index_for_timestep_rules():
    s = EDMDPMSolverMultistepScheduler()
    s.set_timesteps(20)
    for i, t in enumerate(s.timesteps):
        assert s.index_for_timestep(t) == i  # unique values: first (only) match
    assert s.index_for_timestep(torch.tensor(123.0)) == len(s.timesteps) - 1  # not in the table: last index
    s.timesteps = torch.tensor([3.0, 2.0, 2.0, 1.0])
    assert s.index_for_timestep(torch.tensor(2.0)) == 2  # duplicated value: SECOND match (image-to-image safety rule)


# -- EDM Heun sampler (Karras et al., Algorithm 1, S_churn = 0): exact on a constant denoiser ------------------------------
def test_heun_sampler_constant_denoiser_returns_x0():
    from oracle import pipelines as OP

    x0 = 0.3 * torch.randn(1, 84, 2, 3, 5, generator=torch.Generator().manual_seed(5))
    sd = 0.5

    class Net:
        dtype = torch.float32
        device = torch.device("cpu")

        class config:
            out_channels = 84

        calls = 0

        def __call__(self, x, t, known, time_elapsed=None, **kw):
            # t = c_noise = ln(sigma)/4; x = c_in * x_hat  ->  F with c_skip x_hat + c_out F = x0 (what a perfect denoiser returns)
            Net.calls += 1
            sigma = torch.exp(4.0 * t.double()).reshape(-1, 1, 1, 1, 1)
            c_in = 1.0 / (sigma**2 + sd**2) ** 0.5
            x_hat = x.double() / c_in
            c_skip = sd**2 / (sigma**2 + sd**2)
            c_out = sigma * sd / (sigma**2 + sd**2) ** 0.5
            f = ((x0.double() - c_skip * x_hat) / c_out).float()
            return type("O", (), {"sample": f})()

    out = OP.edm_AR_sampler(Net(), EDMDPMSolverMultistepScheduler(), batch_size=1, return_seq_len=2, num_inference_steps=6,
                            generator=[torch.Generator().manual_seed(0)], known_latents=torch.zeros(1, 84, 1, 3, 5), timestamps=torch.tensor([2018010100]))
    assert Net.calls == 2 * 6 - 1  # Heun: 2N - 1 evaluations
    # dx/dsigma = (x - x0)/sigma is linear in sigma along the solution: Euler and the trapezoid correction are both exact
    assert ((out.double() - x0.double()).norm() / x0.double().norm()).item() < 2e-5  # fp32 network I/O at sigma = 80


def test_randn_tensor_member_seeding_contract():
    """member k's noise must not depend on which other members share its batch (pipelines/utils.py:703-706): one generator
    per member, each drawing its own (1, ...) block"""
    shape = (3, 4, 2, 3, 5)
    gens = [torch.Generator().manual_seed(k) for k in (7, 8, 9)]
    full = L.randn_tensor(shape, generator=gens)
    for j, k in enumerate((7, 8, 9)):
        solo = torch.randn((1,) + shape[1:], generator=torch.Generator().manual_seed(k))
        assert torch.equal(full[j : j + 1], solo)


# -- DDIM / DDPM (oracle/scheduler.py; diffusers 0.32.1 restated) against the papers' closed forms, evaluated another way ---------------
def test_ddim_on_the_exact_epsilon_walks_the_closed_form_marginal():
    """Song et al. (DDIM) eq. 12 with sigma = 0: if the network returns the TRUE noise of x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps, every
    step lands on x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps with the same (x0, eps), and the last step (a_prev = 1) returns x0.
    The expected values come from the fp64 cumulative product of the betas alone, not from the scheduler's own tables."""
    from oracle.scheduler import DDIMScheduler

    for pred in ("epsilon", "sample", "v_prediction"):
        s = DDIMScheduler(clip_sample=False, prediction_type=pred)
        s.set_timesteps(20)
        assert s.timesteps.tolist() == list(range(950, -1, -50))  # "leading": multiples of T // n, descending
        abar = torch.cumprod(1.0 - torch.linspace(1e-4, 2e-2, 1000, dtype=torch.float64), 0)
        g = torch.Generator().manual_seed(0)
        x0, eps = torch.randn(4, 6, generator=g), torch.randn(4, 6, generator=g)
        t0 = int(s.timesteps[0])
        x = (abar[t0].sqrt() * x0 + (1 - abar[t0]).sqrt() * eps).float()
        for t in s.timesteps.tolist():
            a = abar[t]
            out = {"epsilon": eps, "sample": x0, "v_prediction": (a.sqrt() * eps - (1 - a).sqrt() * x0).float()}[pred]
            x = s.step(out, t, x).prev_sample
            ap = abar[t - 50] if t >= 50 else torch.tensor(1.0, dtype=torch.float64)
            want = ap.sqrt() * x0 + (1 - ap).sqrt() * eps
            assert (x - want).abs().max() < 2e-5, (pred, t)
        assert (x - x0).abs().max() < 2e-5


def test_ddpm_posterior_mean_and_variance_match_ho_et_al():
    """Ho et al. (DDPM) eq. 6-7 on the SPACED schedule: q(x_prev | x_t, x0) = N(mu, beta~) with the step's own alpha_t = abar_t / abar_prev,
    mu = sqrt(abar_prev) beta_t / (1 - abar_t) x0 + sqrt(alpha_t) (1 - abar_prev) / (1 - abar_t) x_t, beta~ = (1 - abar_prev) / (1 - abar_t) beta_t.
    The scheduler's deterministic part (a generator whose draw is subtracted back out) and its noise scale against these, in fp64."""
    from oracle.scheduler import DDPMScheduler

    abar = torch.cumprod(1.0 - torch.linspace(1e-4, 2e-2, 1000, dtype=torch.float64), 0)
    for vt in ("fixed_small", "fixed_small_log", "fixed_large"):
        s = DDPMScheduler(clip_sample=False, variance_type=vt)
        s.set_timesteps(10)
        g = torch.Generator().manual_seed(1)
        x0, eps = torch.randn(3, 5, generator=g), torch.randn(3, 5, generator=g)
        for t in s.timesteps.tolist():
            tp = t - 100
            a, ap = abar[t], (abar[tp] if tp >= 0 else torch.tensor(1.0, dtype=torch.float64))
            xt = (a.sqrt() * x0 + (1 - a).sqrt() * eps).float()
            got = s.step(eps, t, xt, generator=torch.Generator().manual_seed(9)).prev_sample
            z = torch.randn(3, 5, generator=torch.Generator().manual_seed(9))
            alpha_t = a / ap
            beta_t = 1 - alpha_t
            mu = ap.sqrt() * beta_t / (1 - a) * x0 + alpha_t.sqrt() * (1 - ap) / (1 - a) * xt.double()
            var = {"fixed_small": (1 - ap) / (1 - a) * beta_t, "fixed_small_log": (1 - ap) / (1 - a) * beta_t, "fixed_large": beta_t}[vt]
            want = mu + (var.clamp(min=1e-20).sqrt() * z if t > 0 else 0.0)
            assert (got - want).abs().max() < 3e-5, (vt, t)
        assert int(s.previous_timestep(0)) == -1 and int(s.previous_timestep(300)) == 200


def test_product_ddim_ddpm_schedules_equal_the_oracles():
    """host side of the product classes (no kernel): tables, spacing rules, previous-timestep rule, config surface"""
    from ladcast_amd.schedulers import DDIMScheduler, DDPMScheduler
    from oracle import scheduler as OS

    for kw in (dict(), dict(beta_schedule="scaled_linear"), dict(beta_schedule="squaredcos_cap_v2"), dict(timestep_spacing="trailing"),
               dict(timestep_spacing="linspace"), dict(steps_offset=1)):
        for P, O in ((DDIMScheduler, OS.DDIMScheduler), (DDPMScheduler, OS.DDPMScheduler)):
            a, b = P(**kw), O(**kw)
            assert torch.equal(a.betas, b.betas) and torch.equal(a.alphas_cumprod, b.alphas_cumprod) and torch.equal(a.timesteps, b.timesteps)
            for n in (1, 7, 20, 50):
                a.set_timesteps(n), b.set_timesteps(n)
                assert torch.equal(a.timesteps, b.timesteps) and a.num_inference_steps == n
            assert a.config.num_train_timesteps == 1000 and a.init_noise_sigma == 1.0 and a.order == 1
    d, od = DDPMScheduler(), OS.DDPMScheduler()
    d.set_timesteps(20), od.set_timesteps(20)
    for t in d.timesteps.tolist():
        assert int(d.previous_timestep(t)) == int(od.previous_timestep(t))
        assert torch.equal(d._get_variance(t), od._get_variance(t))
    s = DDIMScheduler()
    with pytest.raises(ValueError):
        s.set_timesteps(2000)
    with pytest.raises(ValueError):
        s.step(None, 10, None)  # set_timesteps first
    with pytest.raises(NotImplementedError):
        DDIMScheduler(thresholding=True)
    with pytest.raises(NotImplementedError):
        DDPMScheduler(variance_type="learned")
    assert DDIMScheduler.launch_only and not DDPMScheduler.launch_only
    sig = s.graph_signature()
    s.set_timesteps(20)
    assert s.graph_signature() != sig and hash(s.graph_signature()) is not None
