"""The oracle under the reference's mixed precision (oracle/autocast.py): the like-for-like comparator of the product's single-term
`bf16` mode (BASELINE configs[4] "fp16/bf16 mixed"; reference: torch.autocast around the fp32-weight models with the hand-written fp32
islands at models/LaDCast_3D_model.py:953-969 and models/DCAE.py:162-175).  CPU checks: the fp32 oracle is untouched by the island
code, the autocast run really rounds to bf16, and the islands really are fp32."""
import torch

from oracle import autocast as OA
from tests.synth import make_ar, make_dcae, rel_l2, synth_field, synth_known, tiny_ar_config, tiny_dcae_config


def _ar_forward(o, x, known, ts):
    with torch.no_grad():
        return o(x, torch.tensor([0.3]), known.expand(x.shape[0], -1, -1, -1, -1), time_elapsed=ts).sample


def test_ar_oracle_under_reference_autocast():
    o = make_ar(tiny_ar_config(heads=2, layers=1, single=1, refiner=1))
    x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
    known, ts = synth_known(1), torch.tensor([2018010100])
    want = _ar_forward(o, x, known, ts)
    errs = {}
    for policy in ("cpu", "cuda"):
        with OA.reference_autocast(policy):
            assert OA.active()
            got = _ar_forward(o, x, known, ts)
        assert got.dtype == torch.bfloat16  # the last Linear ran under autocast
        errs[policy] = rel_l2(got.float(), want)
        assert 5e-4 < errs[policy] < 5e-2, errs  # bf16 roundings: 2^-9 per operand and a bf16 residual stream
    assert not OA.active()
    assert torch.equal(_ar_forward(o, x, known, ts), want)  # the fp32 oracle is bit for bit what it was (F.layer_norm restored)
    assert torch.nn.functional.layer_norm.__module__ == "torch.nn.functional"


def test_conditioning_embedding_island_is_fp32():
    """inside `reference_autocast` the conditioning embedding (temb) is computed from fp32 inputs with fp32 Linears: hook the main
    time_text_embed and look at what it gets and returns"""
    o = make_ar(tiny_ar_config(heads=2, layers=1, single=1, refiner=1))
    seen = {}

    def hook(mod, args, out):
        seen["in"], seen["out"], seen["autocast"] = args[1].dtype, out.dtype, torch.is_autocast_enabled("cpu")

    h = o.time_text_embed.register_forward_hook(hook)
    x = torch.randn(1, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
    with OA.reference_autocast("cuda"):
        _ar_forward(o, x, synth_known(1), torch.tensor([2018010100]))
    h.remove()
    assert seen == {"in": torch.float32, "out": torch.float32, "autocast": False}


def test_dcae_oracle_under_reference_autocast():
    ae = make_dcae(tiny_dcae_config())
    f = synth_field(1, 13, 120, 240)
    with torch.no_grad():
        z = ae.encode(f).latent
        y = ae.decode(z).sample
        with OA.reference_autocast("cuda"):
            z2 = ae.encode(f).latent
            y2 = ae.decode(z).sample
        assert 1e-3 < rel_l2(z2.float(), z) < 5e-2 and 1e-3 < rel_l2(y2.float(), y) < 5e-2
        assert torch.equal(ae.encode(f).latent, z)


def test_reference_autocast_does_not_nest_and_restores_on_error():
    try:
        with OA.reference_autocast("cuda"):
            raise KeyError("x")
    except KeyError:
        pass
    assert not OA.active() and torch.nn.functional.layer_norm.__module__ == "torch.nn.functional"
    with OA.reference_autocast("cpu"):
        try:
            with OA.reference_autocast("cuda"):
                raise AssertionError("nesting must be refused")
        except RuntimeError:
            pass
