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


def _ref_functions(path, names, ns, class_methods=None):
    """compile the named top-level functions (and, for class_methods = {class: [method, ...]}, those methods as plain functions
    named <class>_<method>) of a reference file from its syntax tree into the namespace ns - the rest of the file (its
    third-party imports) is never executed"""
    import ast

    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    for cls, methods in (class_methods or {}).items():
        cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
        for m in cdef.body:
            if isinstance(m, ast.FunctionDef) and m.name in methods:
                m.name = f"{cls}_{m.name}"
                body.append(m)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


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


def sampler_fixtures():
    """Outputs of the reference's OWN sampler / driver / transform / calendar code (compiled from its syntax tree) with the
    oracle's scheduler and an elementwise toy network plugged in: pins the restated loops (oracle.pipelines) and the calendar
    embedding (oracle.ar_model) to the reference's arithmetic.  The scheduler class itself is third-party (diffusers) and stays
    unpinned; here it is the same object on both sides."""
    import copy
    import math
    from datetime import datetime
    from types import SimpleNamespace
    from typing import Dict, List, Optional, Tuple, Union

    from einops import repeat

    from oracle.layers import randn_tensor
    from oracle.scheduler import EDMDPMSolverMultistepScheduler

    R = "/root/reference/ladcast/"
    typing_ns = {"torch": torch, "np": np, "math": math, "copy": copy, "datetime": datetime, "Optional": Optional, "Union": Union, "List": List,
                 "Dict": Dict, "Tuple": Tuple}
    ns = dict(typing_ns, randn_tensor=randn_tensor, repeat=repeat, Fields2DPipelineOutput=lambda fields: SimpleNamespace(fields=fields))
    _ref_functions(R + "pipelines/edm_sampler.py", ["edm_AR_sampler"], ns)
    _ref_functions(R + "pipelines/utils.py", ["ensemble_AR_sampler"], ns)
    _ref_functions(R + "pipelines/pipeline_AR.py", [], ns, class_methods={"AutoRegressive2DPipeline": ["__call__"]})
    emb = _ref_functions(R + "models/embeddings.py", ["convert_int_to_datetime", "compute_year_progress", "timestamp_tensor_to_time_elapsed",
                                                      "get_year_sincos_embedding"], dict(typing_ns))
    tr = _ref_functions(R + "dataloader/utils.py", ["normalize_transform_3D", "inverse_normalize_transform_3D", "get_transform_3D", "get_inv_transform_3D"],
                        dict(typing_ns))

    out = {}
    C, H, W = 6, 5, 8
    net = ToyNet(C)
    g = torch.Generator().manual_seed(21)
    known1 = torch.randn(1, C, 1, H, W, generator=g) * 0.5
    known3 = torch.randn(3, C, 2, H, W, generator=g) * 0.5
    out["known1"], out["known3"] = known1.numpy(), known3.numpy()
    ts = torch.tensor([2018010100])

    class RefPipe:  # the attributes AutoRegressive2DPipeline.__call__ and ensemble_AR_sampler read
        def __init__(self):
            self.ar_model, self.scheduler, self.scheduler_step_kwargs, self._execution_device = net, EDMDPMSolverMultistepScheduler(), {}, torch.device("cpu")

        def __call__(self, **kw):
            return ns["AutoRegressive2DPipeline___call__"](self, **kw)

    gens = lambda n: [torch.Generator("cpu").manual_seed(k) for k in range(n)]  # noqa: E731
    out["edm_n5"] = ns["edm_AR_sampler"](net, EDMDPMSolverMultistepScheduler(), batch_size=3, return_seq_len=2, num_inference_steps=5, known_latents=known3,
                                         timestamps=ts, generator=gens(3)).numpy()
    out["edm_n1"] = ns["edm_AR_sampler"](net, EDMDPMSolverMultistepScheduler(), batch_size=1, return_seq_len=4, num_inference_steps=1, known_latents=known1,
                                         timestamps=None, generator=gens(1)).numpy()
    out["pipe_n6"] = RefPipe()(batch_size=3, return_seq_len=2, known_latents=known3, timestamps=ts, generator=gens(3), num_inference_steps=6, return_dict=False)[0].numpy()
    out["pipe_n20"] = RefPipe()(batch_size=1, return_seq_len=1, known_latents=known1, timestamps=ts, generator=gens(1), num_inference_steps=20).fields.numpy()
    out["ens_edm"] = ns["ensemble_AR_sampler"](RefPipe(), 5, 3, 4, known_latents=known1, timestamps=ts, batch_size=2, sampler_type="edm").numpy()
    out["ens_pipe"] = ns["ensemble_AR_sampler"](RefPipe(), 4, 2, 4, known_latents=known1, timestamps=ts, batch_size=3, sampler_type="pipeline").numpy()

    stamps = torch.tensor([2018010100, 2018063012, 2020022906, 2019123118, 2020123118])
    out["stamps"] = stamps.numpy()
    out["year_progress"] = emb["timestamp_tensor_to_time_elapsed"](stamps).numpy()
    out["year_emb_256"] = emb["get_year_sincos_embedding"](stamps, 256).numpy()
    out["year_emb_10"] = emb["get_year_sincos_embedding"](stamps[:2], 10, max_period=100).numpy()

    x = torch.randn(C, 3, H, W, generator=g) * 3 + 1
    mean, std = (torch.randn(C, generator=g)).tolist(), (torch.rand(C, generator=g) + 0.3).tolist()
    out["tr_x"], out["tr_mean"], out["tr_std"] = x.numpy(), np.array(mean, dtype=np.float64), np.array(std, dtype=np.float64)
    args = {"mean": mean, "std": std, "target_std": 0.5}
    out["tr_fwd"] = tr["get_transform_3D"]("normalize", args)(x).numpy()
    out["tr_inv"] = tr["get_inv_transform_3D"]("normalize", args)(x).numpy()
    out["tr_fwd_nots"] = tr["get_transform_3D"]("normalize", {"mean": mean, "std": std})(x).numpy()
    np.savez_compressed(os.path.join(HERE, "sampler_ref.npz"), **out)


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
    sampler_fixtures()
    oracle_pins()
    print("wrote", sorted(os.listdir(HERE)))
