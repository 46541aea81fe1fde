"""Seeded synthetic configs / weights / inputs shared by tests, smoke and bench
(weights = default torch init under manual_seed(1234), IC latent
0.5*N(0,1) seed 2, fields N(0,1) seed 0/1, timestamp 2018010100)."""
import torch

from oracle.ar_model import CONFIG_375M, LaDCastTransformer3DModel
from oracle.dcae import CONFIG_DCAE_84, AutoencoderDC


def tiny_ar_config(heads=2, layers=1, single=1, refiner=1):
    return dict(CONFIG_375M, num_attention_heads=heads, num_layers=layers, num_single_layers=single, num_refiner_layers=refiner)


def tiny_dcae_config():
    return dict(
        CONFIG_DCAE_84,
        in_channels=13,
        out_channels=13,
        latent_channels=8,
        encoder_block_out_channels=(32, 64, 64, 128),
        decoder_block_out_channels=(32, 64, 64, 128),
        encoder_layers_per_block=(1, 1, 1, 1),
        decoder_layers_per_block=(1, 1, 1, 1),
    )


def randomize_norms(model, seed=4321):
    """Default init leaves every norm weight at 1 and bias at 0, which would hide a
    dropped affine term; perturb all 1-D parameters deterministically."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return model


def make_ar(cfg, seed=1234):
    torch.manual_seed(seed)
    return randomize_norms(LaDCastTransformer3DModel.from_config(cfg)).eval()


def stress_ar_(model, q_gain=4.0, k_gain=3.0, outliers=8, outlier_scale=50.0, ada_scale=8.0, seed=77):
    """Stand-in for the statistics of TRAINED weights nobody has offline - in place, on an oracle model:
    * q / k RMSNorm gains x `q_gain` / x `k_gain`: score std ~ q_gain * k_gain = 12, |logit| up to ~50 over 2250 keys -> peaked softmax
      (default init: logits ~N(0, 1), a diffuse softmax that hides score errors);
    * `outliers` output channels of the two patch embeds (weights and bias) x `outlier_scale`: a few channels that dominate every
      token's LayerNorm statistics, as the massive-activation channels of trained transformers do;
    * every AdaLN modulation Linear (dual / single blocks, output head, refiner gates) x `ada_scale`: gates, shifts and scales O(1)
      instead of ~0.3.
    Returns the model (same object)."""
    g = torch.Generator().manual_seed(seed)
    D = model.x_embedder.proj.weight.shape[0]
    with torch.no_grad():
        for name, p in model.named_parameters():
            leaf = name.rsplit(".", 2)
            if name.endswith((".norm_q.weight", ".norm_added_q.weight")):
                p.mul_(q_gain)
            elif name.endswith((".norm_k.weight", ".norm_added_k.weight")):
                p.mul_(k_gain)
            elif len(leaf) == 3 and leaf[1] == "linear" and leaf[0].split(".")[-1] in ("norm1", "norm1_context", "norm", "norm_out"):
                p.mul_(ada_scale)  # AdaLayerNormZero / -ZeroSingle / -Continuous / HunyuanVideoAdaNorm modulation Linears
        for emb in (model.x_embedder, model.context_embedder):
            idx = torch.randperm(D, generator=g)[:outliers]
            emb.proj.weight[idx] *= outlier_scale
            emb.proj.bias[idx] *= outlier_scale
    return model


def make_ar_stress(cfg, seed=1234, **kw):
    """`make_ar` + `stress_ar_`: seeded default-init weights pushed to trained-model statistics"""
    return stress_ar_(make_ar(cfg, seed), **kw)


def make_dcae(cfg, seed=1234):
    torch.manual_seed(seed)
    return randomize_norms(AutoencoderDC.from_config(cfg)).eval()


def synth_known(batch=1, t_in=1, seed=2):
    return 0.5 * torch.randn(batch, 84, t_in, 15, 30, generator=torch.Generator().manual_seed(seed))


def synth_field(batch, channels, h, w, seed=0):
    return torch.randn(batch, channels, h, w, generator=torch.Generator().manual_seed(seed))


class Sub:
    """A tensor kept as every `stride`-th value of its flattening (+ its shape and full 2-norm): the form in which the full-size oracle
    outputs are committed (tests/golden/fullsize_*.npz, made by tests/golden/make_fullsize_golden.py).  `rel_l2(got, Sub)` compares the
    same subsample of `got`: an unbiased estimate of the full relative error (thousands of values), at 1/stride of the bytes."""

    def __init__(self, values, stride, shape, norm):
        self.values, self.stride, self.shape, self.norm = values, int(stride), tuple(int(v) for v in shape), float(norm)

    @staticmethod
    def of(t, stride):
        f = t.detach().cpu().float().contiguous().flatten()
        return Sub(f[::stride].clone(), stride, t.shape, f.double().norm().item())

    def pick(self, t):
        assert tuple(t.shape) == self.shape, (tuple(t.shape), self.shape)
        return t.detach().cpu().float().contiguous().flatten()[:: self.stride]


def load_fullsize_golden(path):
    """tests/golden/fullsize_*.npz -> dict: full arrays as tensors, `<key>__sub` / `<key>__meta` pairs as Sub, `<key>__n` lists as lists"""
    import numpy as np

    z = np.load(path)
    out, lists = {}, {}
    for k in z.files:
        if k.endswith("__sub"):
            base = k[: -len("__sub")]
            meta = z[base + "__meta"]
            out[base] = Sub(torch.from_numpy(z[k]), int(meta[0]), [int(v) for v in meta[2:]], float(meta[1]))
        elif k.endswith("__meta"):
            continue
        elif k.endswith("__n"):
            lists[k[: -len("__n")]] = int(z[k][0])
        else:
            out[k] = torch.from_numpy(z[k])
    for base, n in lists.items():
        out[base] = [out.pop(f"{base}_{i}") for i in range(n)]
    return out


def rel_l2(a, b):
    if isinstance(b, Sub):
        a, b = b.pick(a), b.values
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()


class DuckDDIMScheduler:
    """A DDIM-shaped scheduler with NOTHING but the four members the reference's pipeline loop touches (pipelines/pipeline_AR.py:
    85-102): `set_timesteps(n)`, `.timesteps`, `scale_model_input(sample, t)`, `step(model_output, t, sample, return_dict=False)`.
    Deterministic DDIM (eta = 0, epsilon prediction, linear betas, integer timesteps); no `.config`, no `.sigmas`, no
    `init_noise_sigma`, no device argument - anything else the loop reached for would raise AttributeError.  Plain torch,
    device-agnostic: the same object drives the HIP model and the CPU oracle."""

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
        self.T = num_train_timesteps
        self.alphas_cumprod = torch.cumprod(1.0 - torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float64), dim=0)
        self.calls = []

    def set_timesteps(self, num_inference_steps):
        self.stride = self.T // num_inference_steps
        self.timesteps = torch.arange(num_inference_steps - 1, -1, -1, dtype=torch.int64) * self.stride

    def scale_model_input(self, sample, t):
        self.calls.append(("scale", int(t)))
        return sample

    def step(self, model_output, t, sample, return_dict=True, **kw):
        t = int(t.reshape(-1)[0])  # the loop hands over t expanded to the batch, on the model's device
        self.calls.append(("step", t))
        a_t = self.alphas_cumprod[t].item()
        a_prev = self.alphas_cumprod[t - self.stride].item() if t - self.stride >= 0 else 1.0
        x0 = (sample - (1.0 - a_t) ** 0.5 * model_output) / a_t**0.5
        prev = a_prev**0.5 * x0 + (1.0 - a_prev) ** 0.5 * model_output
        return (prev,) if not return_dict else {"prev_sample": prev}


class oracle_threads:
    """tiny-width oracle runs on a 128-core host are slower with all cores than with a few (the ops are small): limit torch's
    intra-op pool for the block (results do not depend on it beyond BLAS blocking, ~1e-7)"""

    def __init__(self, n=16):
        self.n = n

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(min(self.n, self.old))

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


class ToyNet:
    """elementwise stand-in for the noise-prediction model (same call signature, `.config.out_channels`, `.dtype`,
    `.device`): no matrix product and no transcendental, so the sampler outputs are reproducible bit for bit on any CPU"""

    def __init__(self, channels, device="cpu"):
        from types import SimpleNamespace

        self.config = SimpleNamespace(out_channels=channels)
        self.dtype = torch.float32
        self.device = torch.device(device)

    def __call__(self, x, t, known, time_elapsed=None, return_dict=True):
        from types import SimpleNamespace

        t = t.reshape(-1, 1, 1, 1, 1).to(x.dtype)
        ts = 0.0 if time_elapsed is None else (time_elapsed.reshape(-1, 1, 1, 1, 1) % 100).to(x.dtype) * 0.01
        y = 0.75 * x - 0.25 * x / (1.0 + x.abs()) + 0.5 * known.mean(dim=2, keepdim=True) + 0.0625 * t + ts  # bounded, IEEE-exact ops only
        return SimpleNamespace(sample=y) if return_dict else (y,)


# ---- inputs / seeded oracle modules of the building-block fixtures (tests/golden/pieces_ref.npz, made by make_golden.py::piece_fixtures) ----
def piece_inputs():
    """seeded inputs of the reference-owned building blocks (shared by the generator and tests/test_oracle_reference_pins.py)"""
    g = torch.Generator().manual_seed(31)
    return {
        "down_x": torch.randn(2, 8, 6, 8, generator=g), "up_x": torch.randn(2, 16, 3, 4, generator=g), "proj_x": torch.randn(2, 96, 6, 8, generator=g),
        "attn_x": torch.randn(2, 64, 6, 8, generator=g), "patch_x": torch.randn(2, 12, 3, 5, 6, generator=g), "dec_z": torch.randn(2, 4, 3, 2, 3, generator=g),
    }


def piece_modules():
    """the oracle's modules for those blocks with seeded weights (torch CPU RNG: the same numbers on every machine)"""
    from oracle import ar_model as OM
    from oracle import dcae as OD

    torch.manual_seed(77)
    m = {
        "down": OD.DCDownBlock2d(8, 16), "up": OD.DCUpBlock2d(16, 8), "proj": OD.SanaMultiscaleAttentionProjection(32, 1, 5),
        "attn": OD.SanaMultiscaleLinearAttention(64, 64, attention_head_dim=32, kernel_sizes=(5,)), "patch": OM.HunyuanVideoPatchEmbed((1, 1, 1), 12, 40),
    }
    with torch.no_grad():
        m["attn"].norm_out.weight.uniform_(0.5, 1.5)
        m["attn"].norm_out.bias.uniform_(-0.5, 0.5)
    m["up_interp"] = OD.DCUpBlock2d(16, 8, interpolate=True)  # (round 5; drawn AFTER everything above: the older fixtures keep their weights)
    return {k: v.eval() for k, v in m.items()}


class ToyDecoder:
    """decode(z) -> object with .sample; elementwise + a channel repeat (for decode_latent_ens)"""

    device = torch.device("cpu")

    def decode(self, z):
        from types import SimpleNamespace

        return SimpleNamespace(sample=(z * 1.5 - 0.25).repeat_interleave(2, dim=1))
