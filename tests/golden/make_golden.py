"""Generate the committed golden fixtures under tests/golden/.

Run in the BUILD container only (needs /root/reference for the one importable
reference module):  python tests/golden/make_golden.py

* sphere_conv_ref.npz   -- inputs / weights / outputs of the REFERENCE class
  ladcast.models.sphere_conv.SphereConv2d (imported from /root/reference), plus
  its docstring known-answer vector (models/sphere_conv.py:142-172).
* scoring_ref.npz       -- inputs / outputs of the REFERENCE scoring functions
  (ladcast/evaluate/utils.py:9-149).  That module imports xarray at the top,
  which is not installed, and none of these six functions uses it: their
  definitions are taken from the reference file's syntax tree at generation
  time and executed as they stand (nothing of the file is copied into the repo).
* oracle_pins.npz       -- outputs of THIS repo's oracle on seeded synthetic
  inputs (regression pins; they are not reference outputs -- the reference's
  diffusers-dependent path cannot be imported here, see oracle/__init__.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def sphere_conv_fixtures():
    sys.path.insert(0, "/root/reference")
    from ladcast.models.sphere_conv import SphereConv2d as Ref

    out = {}
    cases = [  # (cin, cout, k, groups, bias, H, W)
        (3, 5, 3, 1, True, 7, 12),
        (4, 4, 5, 4, False, 6, 8),
        (6, 6, 3, 6, True, 5, 10),
        (8, 4, 5, 2, True, 9, 16),
        (12, 8, 3, 1, False, 15, 30),
    ]
    torch.manual_seed(7)
    for i, (ci, co, k, g, b, H, W) in enumerate(cases):
        m = Ref(ci, co, k, 1, k // 2, groups=g, bias=b)
        x = torch.randn(2, ci, H, W)
        with torch.no_grad():
            y = m(x)
            xp = Ref.sphere_pad(x, (k // 2, k // 2))
        out[f"c{i}_meta"] = np.array([ci, co, k, g, int(b), H, W])
        out[f"c{i}_x"] = x.numpy()
        out[f"c{i}_w"] = m.weight.detach().numpy()
        if b:
            out[f"c{i}_b"] = m.bias.detach().numpy()
        out[f"c{i}_y"] = y.numpy()
        out[f"c{i}_pad"] = xp.numpy()
    # docstring KAT
    tmp = torch.arange(0, 24).view(1, 1, 3, 8)
    c = Ref(1, 1, 5, 1, 2)
    c.weight.data = torch.tensor(
        [[[[0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 1, 0]]]], dtype=torch.float32
    )
    c.bias.data = torch.tensor([0.0])
    with torch.no_grad():
        out["kat_y"] = c(tmp.float()).numpy()
    out["kat_pad"] = Ref.sphere_pad(tmp, (2, 2)).numpy()
    np.savez_compressed(os.path.join(HERE, "sphere_conv_ref.npz"), **out)


def scoring_fixtures():
    import ast
    from typing import Optional, Union

    src = open("/root/reference/ladcast/evaluate/utils.py").read()
    want = {"get_lat_weights_from_lat_tensor", "get_normalized_lat_weights_based_on_cos", "pointwise_crps_skill",
            "pointwise_crps_spread", "get_crps", "get_acc"}
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {n.name for n in body} == want
    ns = {"torch": torch, "np": np, "Optional": Optional, "Union": Union}
    exec(compile(ast.Module(body=body, type_ignores=[]), "ladcast/evaluate/utils.py", "exec"), ns)

    out = {}
    g = torch.Generator().manual_seed(11)
    for i, (M, C, H, W) in enumerate([(5, 3, 6, 8), (1, 2, 4, 6), (16, 4, 9, 12), (50, 2, 8, 10)]):
        fc = torch.randn(M, C, H, W, generator=g) * 2 + 0.3
        tr = torch.randn(C, H, W, generator=g)
        cl = torch.randn(C, H, W, generator=g) * 0.5
        if i == 0:  # SST-like NaNs over "land" in channel 1 (same points in forecast and truth), one tie
            mask = torch.rand(H, W, generator=g) < 0.3
            fc[:, 1][:, mask] = float("nan")
            tr[1][mask] = float("nan")
            fc[1, 0, 0, 0] = fc[0, 0, 0, 0]
        lat = torch.linspace(-88.5, 88.5, H)
        w_cos = ns["get_normalized_lat_weights_based_on_cos"](lat)
        w_area = ns["get_lat_weights_from_lat_tensor"](lat[None])[0]
        out[f"s{i}_fc"], out[f"s{i}_tr"], out[f"s{i}_cl"], out[f"s{i}_lat"] = fc.numpy(), tr.numpy(), cl.numpy(), lat.numpy()
        out[f"s{i}_w_cos"], out[f"s{i}_w_area"] = w_cos.numpy(), w_area.numpy()
        out[f"s{i}_skill"] = ns["pointwise_crps_skill"](fc, tr.unsqueeze(0), 0).numpy()
        out[f"s{i}_spread"] = ns["pointwise_crps_spread"](fc, 0).numpy()
        out[f"s{i}_crps"] = ns["get_crps"](fc, tr.unsqueeze(0), 0).numpy()
        out[f"s{i}_acc_w"] = ns["get_acc"](fc.mean(dim=0), tr, cl, w_cos.view(1, -1, 1)).numpy()
        out[f"s{i}_acc"] = ns["get_acc"](fc.mean(dim=0), tr, cl, None).numpy()
    np.savez_compressed(os.path.join(HERE, "scoring_ref.npz"), **out)


def oracle_pins():
    from tests.synth import tiny_ar_config, tiny_dcae_config, make_ar, make_dcae, synth_known, synth_field

    from oracle.pipelines import AutoRegressive2DPipeline, ensemble_AR_sampler
    from oracle.scheduler import EDMDPMSolverMultistepScheduler

    out = {}

    def pin(name, t):
        """keep every 13th value (<= 4096 of them) plus the l2 norm: small fixture, still sensitive"""
        flat = t.detach().double().flatten()
        out[name] = flat[::13][:4096].numpy()
        out[name + "_norm"] = np.array(flat.norm().item())

    m = make_ar(tiny_ar_config())
    known = synth_known(1)
    ts = torch.tensor([2018010100])
    with torch.no_grad():
        x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
        y = m(x, torch.tensor([0.3]), known.expand(2, -1, -1, -1, -1), time_elapsed=ts).sample
        pin("tiny_ar_fwd", y)
        pipe = AutoRegressive2DPipeline(m, EDMDPMSolverMultistepScheduler())
        pin("tiny_edm", ensemble_AR_sampler(pipe, 2, 4, 4, known_latents=known, timestamps=ts, sampler_type="edm"))
        pin("tiny_pipeline", ensemble_AR_sampler(pipe, 2, 4, 4, known_latents=known, timestamps=ts, sampler_type="pipeline"))
        ae = make_dcae(tiny_dcae_config())
        f, st = synth_field(1, 8, 48, 64), synth_field(1, 5, 48, 64, seed=1)
        z = ae.encode(f, static_conditioning_tensor=st).latent
        pin("tiny_dcae_z", z)
        pin("tiny_dcae_y", ae.decode(z).sample)
    np.savez_compressed(os.path.join(HERE, "oracle_pins.npz"), **out)


if __name__ == "__main__":
    sphere_conv_fixtures()
    scoring_fixtures()
    oracle_pins()
    print("wrote", sorted(os.listdir(HERE)))
