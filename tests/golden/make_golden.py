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
* sampler_ref.npz, pieces_ref.npz, ar_forward_ref.npz, dcae_forward_ref.npz -- outputs of
  the REFERENCE's own sampler / driver / transform / calendar functions, of its torch-only
  classes, and of every forward method it defines for the transformer and the DCAE, compiled
  the same way (definitions from the syntax tree; the files' diffusers / xarray imports are
  never executed) and run with the oracle's scheduler / leaf layers / seeded parameter
  containers: see sampler_fixtures, piece_fixtures, ar_forward_fixtures,
  dcae_forward_fixtures and tests/test_oracle_reference_pins.py.
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
                m.decorator_list = [d for d in m.decorator_list if "apply_forward_hook" not in ast.unparse(d)]  # accelerate offload hook: a no-op here
                body.append(m)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


from tests.synth import ToyNet  # noqa: E402  (lives in tests/synth.py: the GPU box imports it without importing this script)


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
    # stochastic churn (deterministic=False): the caller's randn_like feeds every step; gamma = S_churn / N (a Python float) and the
    # capped sqrt(2) - 1 (a numpy scalar) - the two ways the reference's expression evaluates - with S_min / S_max cutting steps out
    for name, churn in (("edm_churn_lo", 1.5), ("edm_churn_cap", 40.0)):
        gc = torch.Generator("cpu").manual_seed(77)
        out[name] = ns["edm_AR_sampler"](net, EDMDPMSolverMultistepScheduler(), batch_size=3, return_seq_len=2, num_inference_steps=5, known_latents=known3,
                                         timestamps=ts, generator=gens(3), deterministic=False, S_churn=churn, S_min=0.05, S_max=50.0, S_noise=1.003,
                                         randn_like=lambda x: torch.randn(x.shape, generator=gc, dtype=x.dtype)).numpy()
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


def _ref_classes(path, names, ns):
    """compile the named top-level classes of a reference file from its syntax tree into ns (only classes whose definition
    needs nothing but torch and what ns already holds)"""
    import ast

    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


from tests.synth import ToyDecoder, piece_inputs, piece_modules  # noqa: E402,F401  (in tests/synth.py: the GPU box uses them without importing this script)


def piece_fixtures():
    """Outputs of the reference's own building-block classes that need nothing but torch and the (importable) SphereConv2d -
    DCDownBlock2d, DCUpBlock2d, SanaMultiscaleAttentionProjection (models/DCAE.py:67-93,447-536), the ReLU linear-attention processor
    SanaMultiscaleAttnProcessor2_0 + apply_linear_attention (models/DCAE.py:158-178,200-267) run on the oracle's parameter container,
    HunyuanVideoPatchEmbed (models/embeddings.py:38-59), decode_latent_ens (pipelines/utils.py:51-80) - on seeded weights."""
    import types
    from typing import Optional, Tuple, Union

    import torch.nn as nn
    import torch.nn.functional as F
    from einops import rearrange

    sys.path.insert(0, "/root/reference")
    from ladcast.models.sphere_conv import SphereConv2d

    R = "/root/reference/ladcast/"
    ns = {"torch": torch, "nn": nn, "F": F, "Optional": Optional, "Tuple": Tuple, "Union": Union, "SphereConv2d": SphereConv2d,
          "SanaMultiscaleLinearAttention": object}
    _ref_classes(R + "models/DCAE.py", ["DCDownBlock2d", "DCUpBlock2d", "SanaMultiscaleAttentionProjection", "SanaMultiscaleAttnProcessor2_0"], ns)
    _ref_functions(R + "models/DCAE.py", [], ns, class_methods={"SanaMultiscaleLinearAttention": ["apply_linear_attention"]})
    emb = _ref_classes(R + "models/embeddings.py", ["HunyuanVideoPatchEmbed"], {"torch": torch, "nn": nn, "Union": Union, "Tuple": Tuple})
    tr = _ref_functions(R + "dataloader/utils.py", ["inverse_normalize_transform_3D"], {"torch": torch})
    pu = _ref_functions(R + "pipelines/utils.py", ["decode_latent_ens"], {"torch": torch, "Optional": Optional, "rearrange": rearrange,
                                                                         "inverse_normalize_transform_3D": tr["inverse_normalize_transform_3D"]})
    x, om = piece_inputs(), piece_modules()
    out = {}
    with torch.no_grad():
        ref = ns["DCDownBlock2d"](8, 16, downsample=True, shortcut=True)
        ref.load_state_dict(om["down"].state_dict(), strict=True)
        out["down"] = ref(x["down_x"]).numpy()
        ref = ns["DCUpBlock2d"](16, 8, interpolate=False, shortcut=True)
        ref.load_state_dict(om["up"].state_dict(), strict=True)
        out["up"] = ref(x["up_x"]).numpy()
        ref = ns["DCUpBlock2d"](16, 8, interpolate=True, shortcut=True)  # upsample_block_type = "interpolate" (models/DCAE.py:498-525,677-682)
        ref.load_state_dict(om["up_interp"].state_dict(), strict=True)
        out["up_interp"] = ref(x["up_x"]).numpy()
        ref = ns["SanaMultiscaleAttentionProjection"](32, 1, 5)
        ref.load_state_dict(om["proj"].state_dict(), strict=True)
        out["proj"] = ref(x["proj_x"]).numpy()
        # the reference's processor + linear-attention method on the oracle's parameter container (the reference's own container
        # class cannot be constructed: its norm comes from diffusers.get_normalization)
        hybrid = om["attn"]
        hybrid.norm_type, hybrid.nonlinearity = "rms_norm", nn.ReLU()
        hybrid.apply_linear_attention = types.MethodType(ns["SanaMultiscaleLinearAttention_apply_linear_attention"], hybrid)
        out["attn"] = ns["SanaMultiscaleAttnProcessor2_0"]()(hybrid, x["attn_x"]).numpy()
        ref = emb["HunyuanVideoPatchEmbed"]((1, 1, 1), 12, 40)
        ref.load_state_dict(om["patch"].state_dict(), strict=True)
        out["patch"] = ref(x["patch_x"]).numpy()
        mean, std = torch.linspace(-1, 1, 8), torch.linspace(0.5, 2, 8)
        out["dec_all"] = pu["decode_latent_ens"](ToyDecoder(), x["dec_z"], mean, std).numpy()
        out["dec_first"] = pu["decode_latent_ens"](ToyDecoder(), x["dec_z"], None, None, extract_first=1).numpy()
    np.savez_compressed(os.path.join(HERE, "pieces_ref.npz"), **out)


def dcae_forward_fixtures():
    """The reference's own forward code of every DCAE class it defines - ResBlock, GLUMBConv, EfficientViTBlock, SanaMultiscaleLinearAttention
    (+ processor, apply_linear_attention), DCDownBlock2d, DCUpBlock2d, Encoder, Decoder, AutoencoderDC.encode / decode / forward
    (models/DCAE.py:93-732,948-1087) - bound onto the oracle's tiny
    autoencoder (same attribute names, seeded weights).  The leaves it calls are torch modules, the oracle's RMSNorm (third-party in the
    reference, unpinned) and the pinned SphereConv2d.  Flags the reference sets in its constructors are set here from the shipped config
    (configs/DC_AE_84_pretrain.yaml: rms_norm, silu, pixel_(un)shuffle, shortcuts on)."""
    import types
    from typing import Optional, Tuple, Union

    import torch.nn as nn
    import torch.nn.functional as F

    from oracle import dcae as OD
    from tests.synth import make_dcae, synth_field, tiny_dcae_config

    R = "/root/reference/ladcast/models/DCAE.py"
    ns = {"torch": torch, "nn": nn, "F": F, "Optional": Optional, "Tuple": Tuple, "Union": Union, "SanaMultiscaleLinearAttention": object}
    _ref_classes(R, ["SanaMultiscaleAttnProcessor2_0"], ns)
    methods = {"SanaMultiscaleLinearAttention": ["apply_linear_attention", "forward"], "GLUMBConv": ["forward"], "ResBlock": ["forward"],
               "EfficientViTBlock": ["forward"], "DCDownBlock2d": ["forward"], "DCUpBlock2d": ["forward"], "Encoder": ["forward"], "Decoder": ["forward"],
               "AutoencoderDC": ["_encode", "encode", "_decode", "decode", "forward"], "AdaLayerNormZeroSingle4Sana": ["forward"]}
    from types import SimpleNamespace

    ns["EncoderOutput"] = lambda latent: SimpleNamespace(latent=latent)
    ns["DecoderOutput"] = lambda sample: SimpleNamespace(sample=sample)
    _ref_functions(R, [], ns, class_methods=methods)

    class FP32LayerNormNoAffine(nn.Module):  # diffusers FP32LayerNorm(C, elementwise_affine=False, eps): layer_norm in fp32, cast back
        def __init__(self, dim, eps):
            super().__init__()
            self.dim, self.eps = dim, eps

        def forward(self, x):
            return F.layer_norm(x.float(), (self.dim,), None, None, self.eps).to(x.dtype)

    def bind(ae, temb):
        """the reference's forward methods onto the oracle's containers; temb: the timestep-conditioned variant keeps its time_emb_porj /
        norm_in sub-modules (the shipped config's containers get the None the reference's constructors would set)"""
        from oracle.layers import get_timestep_embedding

        no_t = {} if temb else dict(norm_in=None, time_emb_porj=None)
        shims = {
            OD.SanaMultiscaleLinearAttention: dict(norm_type="rms_norm", nonlinearity=nn.ReLU(), processor=ns["SanaMultiscaleAttnProcessor2_0"](), **no_t),
            OD.GLUMBConv: dict(nonlinearity=nn.SiLU(), norm_type="rms_norm", residual_connection=True),
            OD.ResBlock: dict(norm_type="rms_norm", **({} if temb else dict(time_emb_porj=None))),
            OD.EfficientViTBlock: {}, OD.DCDownBlock2d: dict(downsample=True), OD.DCUpBlock2d: dict(interpolate=False),
            OD.Encoder: dict(out_shortcut=True), OD.Decoder: dict(in_shortcut=True, conv_act=nn.ReLU()),
            OD.AutoencoderDC: dict(time_proj=(lambda t: get_timestep_embedding(t, 256))) if temb else {},
            OD.AdaLayerNormZeroSingle4Sana: dict(silu=nn.SiLU()),
        }
        bound = 0
        for mod in ae.modules():
            for cls, attrs in shims.items():
                if type(mod) is cls:
                    for k, v in attrs.items():
                        object.__setattr__(mod, k, v)  # plain attributes (not registered sub-modules: the state dict stays the oracle's)
                    if cls is OD.AdaLayerNormZeroSingle4Sana:
                        object.__setattr__(mod, "norm", FP32LayerNormNoAffine(mod.embedding_dim, 1e-15))
                    for meth in methods[cls.__name__]:
                        object.__setattr__(mod, meth, types.MethodType(ns[f"{cls.__name__}_{meth}"], mod))
                    bound += 1
        return bound

    ae = make_dcae(tiny_dcae_config())
    assert bind(ae, False) >= 12
    f, st = synth_field(2, 8, 48, 64), synth_field(1, 5, 48, 64, seed=1)
    with torch.no_grad():  # AutoencoderDC.encode / decode / forward are the reference's too (models/DCAE.py:948-1087)
        z = ae.encode(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        y = ae.decode(z, return_static=True).sample
        y_nostatic = ae.decode(z, return_dict=False)[0]
        full = ae.forward(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).sample
    assert y_nostatic.shape[1] == y.shape[1] - 5 and torch.equal(full, y_nostatic)
    # round 5: the timestep-conditioned variant (temb_channels; models/DCAE.py:36-64,147-153,193-198,256-257,351-365,845-850,982-984,1067-1071):
    # ResBlock scale / shift, AdaLayerNormZeroSingle4Sana + gate in the linear-attention blocks, raw timesteps through time_proj +
    # timestep_embedder in encode / decode, `time_elapsed` in forward - all the reference's own code; Timesteps / TimestepEmbedding /
    # FP32LayerNorm are diffusers leaves (the oracle's restatements)
    aet = make_dcae(dict(tiny_dcae_config(), temb_channels=48))
    assert bind(aet, True) >= 16
    tt = torch.tensor([0.3, 1.7])
    with torch.no_grad():
        zt = aet.encode(f, temb=tt, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        yt = aet.decode(zt, temb=tt, return_static=True).sample
        fullt = aet.forward(f, time_elapsed=tt, static_conditioning_tensor=st.expand(2, -1, -1, -1), return_static=True).sample
    assert torch.equal(fullt, yt) and (zt - z).abs().max() > 0.1
    # round 6: the `layers_per_block[0] == 0` form of the DC-AE family (models/DCAE.py:559-579,696-712: no stage at full resolution - the encoder's
    # conv_in is a DCDownBlock2d, the decoder's conv_out a DCUpBlock2d, both WITHOUT shortcut; their forwards are the reference's own code here)
    ae0 = make_dcae(dict(tiny_dcae_config(), encoder_layers_per_block=(0, 1, 1, 1), decoder_layers_per_block=(0, 1, 1, 1)))
    assert bind(ae0, False) >= 10 and type(ae0.encoder.conv_in) is OD.DCDownBlock2d and type(ae0.decoder.conv_out) is OD.DCUpBlock2d
    with torch.no_grad():
        z0 = ae0.encode(f, static_conditioning_tensor=st.expand(2, -1, -1, -1)).latent
        y0 = ae0.decode(z0, return_static=True).sample
    assert z0.shape == z.shape and y0.shape == y.shape
    np.savez_compressed(os.path.join(HERE, "dcae_forward_ref.npz"), z=z.numpy(), y=y.numpy(), y_nostatic=y_nostatic.numpy(), z_temb=zt.numpy(), y_temb=yt.numpy(),
                        z_layers0=z0.numpy(), y_layers0=y0.numpy())


def _strip_inner_imports(fn_node):
    """drop `from ... import ...` statements inside a function body (the processor imports apply_rotary_emb at call time); the names
    come from the namespace instead"""
    import ast

    class T(ast.NodeTransformer):
        def visit_ImportFrom(self, node):
            return ast.Pass()

    return ast.fix_missing_locations(T().visit(fn_node))


def ar_forward_fixtures():
    """The reference's own forward code of every class it defines on the transformer path - LaDCastAttnProcessor2_0.__call__, HunyuanVideoAdaNorm,
    the three token-refiner classes, LaDCastSingleTransformerBlock, LaDCastTransformerBlock, LaDCastTransformer3DModel.forward
    (models/LaDCast_3D_model.py:64-566,833-1071), LaDCastRotaryPosEmbed_from_grid (models/embeddings.py:252-327) and get_year_sincos_embedding -
    bound onto the oracle's tiny model (same attribute names, seeded weights).  The leaves they call are the oracle's restatements of the
    diffusers layers (Attention container, AdaLayerNorm*, FeedForward, embeddings, apply_rotary_emb, get_1d_rotary_pos_embed: third-party,
    unpinned) and torch.  What this pins is everything the reference itself wrote: token order, stream concatenation, which rotary table goes
    to which rows, the gating / residual wiring, the un-patchify permutation."""
    import ast
    import logging
    import types
    import warnings
    from numbers import Number
    from types import SimpleNamespace
    from typing import Any, Dict, List, Optional, Tuple, Union

    import torch.nn as nn
    import torch.nn.functional as F

    from oracle import ar_model as OM
    from oracle import layers as OL
    from tests.synth import make_ar, synth_known, tiny_ar_config

    R = "/root/reference/ladcast/models/"
    typing_ns = {"torch": torch, "nn": nn, "F": F, "np": np, "warnings": warnings, "Number": Number, "Optional": Optional, "Tuple": Tuple, "Union": Union,
                 "List": List, "Dict": Dict, "Any": Any, "math": __import__("math"), "datetime": __import__("datetime").datetime}
    emb = _ref_functions(R + "embeddings.py", ["convert_int_to_datetime", "compute_year_progress", "timestamp_tensor_to_time_elapsed",
                                               "get_year_sincos_embedding"], dict(typing_ns))
    rope_ns = dict(typing_ns, get_1d_rotary_pos_embed=lambda dim, pos, theta, use_real=True: OL.get_1d_rotary_pos_embed(dim, pos, theta))
    _ref_classes(R + "embeddings.py", ["LaDCastRotaryPosEmbed_from_grid"], rope_ns)

    ns = dict(typing_ns, Attention=object, AttentionProcessor=object, apply_rotary_emb=OL.apply_rotary_emb, USE_PEFT_BACKEND=False,
              logger=logging.getLogger("ladcast_ref"), Transformer2DModelOutput=lambda sample: SimpleNamespace(sample=sample),
              get_year_sincos_embedding=emb["get_year_sincos_embedding"], get_1d_rotary_pos_embed=rope_ns["get_1d_rotary_pos_embed"],
              is_torch_version=lambda *a: True, scale_lora_layers=None, unscale_lora_layers=None)
    tree = ast.parse(open(R + "LaDCast_3D_model.py").read())
    wanted = {"LaDCastAttnProcessor2_0": ["__call__"], "HunyuanVideoAdaNorm": ["forward"], "LaDCastIndividualTokenRefinerBlock": ["forward"],
              "LaDCastIndividualTokenRefiner": ["forward"], "LaDCastTokenRefiner": ["forward"], "LaDCastSingleTransformerBlock": ["forward"],
              "LaDCastTransformerBlock": ["forward"], "LaDCastTransformer3DModel": ["forward"]}
    body = []
    for cdef in tree.body:
        if isinstance(cdef, ast.ClassDef) and cdef.name in wanted:
            for m in cdef.body:
                if isinstance(m, ast.FunctionDef) and m.name in wanted[cdef.name]:
                    m = _strip_inner_imports(m)
                    m.name = f"{cdef.name}_{m.name}"
                    body.append(m)
    assert len(body) == 8
    exec(compile(ast.Module(body=body, type_ignores=[]), R + "LaDCast_3D_model.py", "exec"), ns)

    class RefProcessor:
        __call__ = ns["LaDCastAttnProcessor2_0___call__"]

    def bind_reference_forwards(model):
      """the reference's forward methods onto one oracle model's containers (in place)"""
      bind = {OM.HunyuanVideoAdaNorm: ("HunyuanVideoAdaNorm", dict(nonlinearity=nn.SiLU())), OM.RefinerBlock: ("LaDCastIndividualTokenRefinerBlock", {}),
              OM.IndividualTokenRefiner: ("LaDCastIndividualTokenRefiner", {}), OM.TokenRefiner: ("LaDCastTokenRefiner", {}),
              OM.SingleBlock: ("LaDCastSingleTransformerBlock", dict(act_mlp=nn.GELU(approximate="tanh"))), OM.DualBlock: ("LaDCastTransformerBlock", {})}
      n_bound = 0
      for mod in model.modules():
          if isinstance(mod, OL.Attention):
              object.__setattr__(mod, "processor", RefProcessor())
              n_bound += 1
          for cls, (ref_name, attrs) in bind.items():
              if type(mod) is cls:
                  for k, v in attrs.items():
                      object.__setattr__(mod, k, v)
                  object.__setattr__(mod, "forward", types.MethodType(ns[f"{ref_name}_forward"], mod))
                  n_bound += 1
      c = model.config
      Rope = rope_ns["LaDCastRotaryPosEmbed_from_grid"]
      for k, v in dict(scale_attn_by_lat=False, gradient_checkpointing=False,
                       rope=Rope(rope_dim_list=c.rope_axes_dim, patch_size_list=[c.patch_size_t, c.patch_size, c.patch_size], theta=c.rope_theta),
                       cond_rope=Rope(rope_dim_list=c.conditioning_tensor_rope_axes_dim, patch_size_list=[c.patch_size_t, c.patch_size, c.patch_size],
                                      theta=c.rope_theta)).items():
          object.__setattr__(model, k, v)
      object.__setattr__(model, "forward", types.MethodType(ns["LaDCastTransformer3DModel_forward"], model))
      assert n_bound >= 9, n_bound
      return model

    model = bind_reference_forwards(make_ar(tiny_ar_config()))

    out = {}
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, (B, Rr, Bt, stamp) in {"a": (2, 4, 1, 2018010100), "b": (1, 1, 1, 2019063012), "c": (3, 2, 3, None)}.items():
            x = torch.randn(B, 84, Rr, 15, 30, generator=torch.Generator().manual_seed(3))
            known = synth_known(B)
            t = torch.linspace(-1.2, 1.0, Bt)
            te = None if stamp is None else torch.tensor([stamp])
            y = model(x, t, known, time_elapsed=te).sample.double().flatten()
            out[name] = y[::7].float().numpy()  # every 7th value + the norm: small fixture, still sensitive to any wiring change
            out[name + "_norm"] = np.array(y.norm().item())
        # scale_attn_by_lat = True (off in both shipped configs): the reference forward builds a (1, 1, 1, keys) float mask from
        # attn_lat_weights (:873-880) and hands it to every block; the constructor lines that make attn_lat_weights (:684-693) are
        # re-run here with the reference's own weight function (evaluate/utils.py:40-48).  Amplified 200x in a second case so that the
        # mask (values ~1/450) moves the output by more than rounding.
        lw = _ref_functions("/root/reference/ladcast/evaluate/utils.py", ["get_normalized_lat_weights_based_on_cos"], dict(typing_ns))
        tmp = lw["get_normalized_lat_weights_based_on_cos"](np.linspace(-83.25, 84.75, 15))
        tmp = torch.from_numpy(tmp / tmp.sum()).float().repeat_interleave(30)
        object.__setattr__(model, "scale_attn_by_lat", True)
        for name, amp in (("lat", 1.0), ("lat200", 200.0)):
            object.__setattr__(model, "attn_lat_weights", (amp * tmp).view(1, 1, 1, -1))
            x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
            y = model(x, torch.tensor([0.3]), synth_known(2), time_elapsed=torch.tensor([2018010100])).sample.double().flatten()
            out[name] = y[::7].float().numpy()
            out[name + "_norm"] = np.array(y.norm().item())
        # nope = True (off in both shipped configs; round 5): the reference forward takes its rotary tables from get_1d_rotary_pos_embed over the
        # TEMPORAL coordinate alone, for the whole head dimension (:897-918).  get_1d_rotary_pos_embed is diffusers' (the oracle's restatement,
        # here with the reference's own call signature); which rows get which table, and the repeat over a frame's tokens, is reference code.
        from oracle.layers import get_1d_rotary_pos_embed as _g1d

        ns["get_1d_rotary_pos_embed"] = lambda dim, pos, theta, use_real=True: _g1d(dim, pos, theta)
        object.__setattr__(model, "scale_attn_by_lat", False)
        model.config.nope = True
        x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
        y = model(x, torch.tensor([0.3]), synth_known(2), time_elapsed=torch.tensor([2018010100])).sample.double().flatten()
        out["nope"] = y[::7].float().numpy()
        out["nope_norm"] = np.array(y.norm().item())
        model.config.nope = False
        # round 6: patch sizes != 1 (no shipped YAML; models/LaDCast_3D_model.py:657-663,758,866-871,885-896,1044-1062, models/embeddings.py:38-59):
        # Conv3d patch embeds with kernel = stride = (p_t, p, p), rotary grids over the PATCH grid, an output head of p_t p p C columns per token and
        # the reference's un-patchify permutation - all reference code here; the oracle's HunyuanVideoPatchEmbed is pinned by pieces_ref.npz
        for name, (p_, pt_, t_in) in {"patch3": (3, 1, 1), "patch5_t2": (5, 2, 2)}.items():
            mp = bind_reference_forwards(make_ar(dict(tiny_ar_config(), patch_size=p_, patch_size_t=pt_)))
            x = torch.randn(2, 84, 4, 15, 30, generator=torch.Generator().manual_seed(3))
            known = 0.5 * torch.randn(2, 84, t_in, 15, 30, generator=torch.Generator().manual_seed(2))
            y = mp(x, torch.tensor([0.3]), known, time_elapsed=torch.tensor([2018010100])).sample
            assert tuple(y.shape) == (2, 84, 4, 15, 30)
            y = y.double().flatten()
            out[name] = y[::7].float().numpy()
            out[name + "_norm"] = np.array(y.norm().item())
    np.savez_compressed(os.path.join(HERE, "ar_forward_ref.npz"), **out)


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
    piece_fixtures()
    dcae_forward_fixtures()
    ar_forward_fixtures()
    oracle_pins()
    print("wrote", sorted(os.listdir(HERE)))
