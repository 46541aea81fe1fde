"""checkpoint loader, hub-cache layout and training-checkpoint folder names (host side; CPU).

The reference loads through diffusers' ``ModelMixin.from_pretrained`` (evaluate/pred_rollout.py:299-324,
train_AR.py:516-518) and names its training folders at train_AR.py:561-570,799-803,1131-1135.  No trained weights exist
offline, so the files here are written by ``save_pretrained`` of seeded random-init models; what is checked is the file
layout, the key map (against the oracle's module tree) and the loader's rules.
"""
import json
import os

import pytest
import torch

from tests.synth import tiny_ar_config, tiny_dcae_config


def _models():
    from ladcast_amd.models import AutoencoderDC, LaDCastTransformer3DModel

    torch.manual_seed(7)
    return LaDCastTransformer3DModel.from_config(tiny_ar_config()), AutoencoderDC.from_config(tiny_dcae_config())


def _same(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    return list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)


def test_folder_roundtrip_keys_and_config(tmp_path):
    from ladcast_amd.models import AutoencoderDC, LaDCastTransformer3DModel
    from oracle.ar_model import LaDCastTransformer3DModel as OracleAR
    from safetensors import safe_open

    ar, ae = _models()
    ar.save_pretrained(tmp_path / "ar")
    ae.save_pretrained(tmp_path / "ae")
    assert sorted(os.listdir(tmp_path / "ar")) == ["config.json", "diffusion_pytorch_model.safetensors"]
    cfg = json.load(open(tmp_path / "ar" / "config.json"))
    assert cfg["_class_name"] == "LaDCastTransformer3DModel" and isinstance(cfg["rope_axes_dim"], list)
    # the file's key set is the reference module tree's (restated by the oracle), not a build-private packing
    with safe_open(str(tmp_path / "ar" / "diffusion_pytorch_model.safetensors"), "pt") as f:
        assert set(f.keys()) == set(OracleAR.from_config(tiny_ar_config()).state_dict())
    ar2 = LaDCastTransformer3DModel.from_pretrained(str(tmp_path / "ar"))
    ae2 = AutoencoderDC.from_pretrained(str(tmp_path), subfolder="ae")
    assert _same(ar, ar2) and _same(ae, ae2) and not ar2.training and not ae2.training
    assert ar2.config.to_dict() == json.loads(json.dumps(ar.config.to_dict()))  # tuples come back as lists, as with diffusers
    # the oracle's module tree takes the file as is: one on-disk format for checker and product
    from safetensors.torch import load_file

    OracleAR.from_config(tiny_ar_config()).load_state_dict(load_file(str(tmp_path / "ar" / "diffusion_pytorch_model.safetensors")), strict=True)
    assert LaDCastTransformer3DModel.from_pretrained(str(tmp_path / "ar"), torch_dtype=torch.float64).dtype == torch.float64


def test_hub_cache_layout_is_resolved_offline(tmp_path, monkeypatch):
    from ladcast_amd.models import AutoencoderDC
    from ladcast_amd.models.modeling_utils import hub_cache_dir, resolve_model_folder

    _, ae = _models()
    commit = "0123456789abcdef0123456789abcdef01234567"
    repo = tmp_path / "hub" / "models--tonyzyl--ladcast"
    (repo / "refs").mkdir(parents=True)
    (repo / "refs" / "main").write_text(commit)
    ae.save_pretrained(repo / "snapshots" / commit / "V0.1.X" / "DCAE")
    monkeypatch.setenv("HF_HOME", str(tmp_path))
    for var in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        monkeypatch.delenv(var, raising=False)
    assert hub_cache_dir() == str(tmp_path / "hub")
    # the reference's call, verbatim (train_AR.py:516-518 with the default --encdec_model, :423)
    got = AutoencoderDC.from_pretrained("tonyzyl/ladcast", subfolder="V0.1.X/DCAE")
    assert _same(ae, got)
    assert resolve_model_folder("tonyzyl/ladcast", "V0.1.X/DCAE", revision=commit).endswith("V0.1.X/DCAE")
    assert _same(ae, AutoencoderDC.from_pretrained("tonyzyl/ladcast", subfolder="V0.1.X/DCAE", cache_dir=str(tmp_path / "hub")))
    with pytest.raises(OSError, match="not in the local cache"):
        AutoencoderDC.from_pretrained("tonyzyl/ladcast", subfolder="V0.2.X/DCAE")
    with pytest.raises(OSError, match="not in the local cache"):
        AutoencoderDC.from_pretrained("someone/else")
    with pytest.raises(OSError, match="neither a directory"):
        AutoencoderDC.from_pretrained(str(tmp_path / "nope" / "deeper" / "x"))


def test_sharded_variant_and_bin_weights(tmp_path):
    from ladcast_amd.models import LaDCastTransformer3DModel as M

    ar, _ = _models()
    ar.save_pretrained(tmp_path / "sharded", max_shard_size=200_000)
    names = sorted(os.listdir(tmp_path / "sharded"))
    assert "diffusion_pytorch_model.safetensors.index.json" in names and len(names) > 3
    assert all(n.startswith("diffusion_pytorch_model-0") for n in names if n.endswith(".safetensors"))
    assert _same(ar, M.from_pretrained(str(tmp_path / "sharded")))
    idx = json.load(open(tmp_path / "sharded" / "diffusion_pytorch_model.safetensors.index.json"))
    assert idx["metadata"]["total_size"] == sum(p.numel() * 4 for p in ar.state_dict().values())
    os.remove(tmp_path / "sharded" / names[-1] if names[-1].endswith(".safetensors") else tmp_path / "sharded" / names[-2])
    with pytest.raises(OSError, match="shards that are not in"):
        M.from_pretrained(str(tmp_path / "sharded"))

    ar.save_pretrained(tmp_path / "var", variant="fp32")
    assert "diffusion_pytorch_model.fp32.safetensors" in os.listdir(tmp_path / "var")
    assert _same(ar, M.from_pretrained(str(tmp_path / "var"), variant="fp32"))
    with pytest.raises(OSError, match="no diffusion_pytorch_model.safetensors"):
        M.from_pretrained(str(tmp_path / "var"))

    os.makedirs(tmp_path / "bin")
    json.dump(ar.config.to_dict(), open(tmp_path / "bin" / "config.json", "w"))
    torch.save(ar.state_dict(), tmp_path / "bin" / "diffusion_pytorch_model.bin")
    assert _same(ar, M.from_pretrained(str(tmp_path / "bin")))


def test_strict_key_and_shape_checks(tmp_path):
    from ladcast_amd.models import LaDCastTransformer3DModel as M
    from safetensors.torch import load_file, save_file

    ar, _ = _models()
    ar.save_pretrained(tmp_path / "m")
    f = str(tmp_path / "m" / "diffusion_pytorch_model.safetensors")
    state = load_file(f)
    dropped = dict(state)
    dropped.pop("proj_out.bias")
    save_file(dropped, f)
    with pytest.raises(RuntimeError, match="proj_out.bias"):
        M.from_pretrained(str(tmp_path / "m"))
    assert M.from_pretrained(str(tmp_path / "m"), strict=False) is not None  # diffusers' lenient behaviour, on request
    save_file(dict(state, **{"proj_out.bias": torch.zeros(3)}), f)
    with pytest.raises(RuntimeError, match="size mismatch"):
        M.from_pretrained(str(tmp_path / "m"))
    save_file(dict(state, stranger=torch.zeros(1)), f)
    with pytest.raises(RuntimeError, match="stranger"):
        M.from_pretrained(str(tmp_path / "m"))
    os.remove(tmp_path / "m" / "config.json")
    with pytest.raises(OSError, match="no config.json"):
        M.from_pretrained(str(tmp_path / "m"))


def test_training_checkpoint_folders_and_ema_config(tmp_path):
    """``checkpoint-<step>/{ar_model,ar_model_ema}`` (train_AR.py:561-570), "latest" by integer step (:799-803); the EMA folder's
    config.json carries EMAModel's scalars next to the constructor arguments (:572-574)."""
    from ladcast_amd.models import LaDCastTransformer3DModel as M
    from ladcast_amd.models.modeling_utils import EMA_CONFIG_KEYS, checkpoint_model_folder, list_checkpoints

    ar, _ = _models()
    ema_scalars = dict(decay=0.9999, min_decay=0.0, optimization_step=1200, update_after_step=0, use_ema_warmup=True, inv_gamma=1.0, power=0.75)
    assert set(ema_scalars) == set(EMA_CONFIG_KEYS)
    out = tmp_path / "run"
    for step in (900, 1000, 10000):  # lexicographic order would pick 900
        ar.save_pretrained(out / f"checkpoint-{step}" / "ar_model")
        with torch.no_grad():
            ar.proj_out.bias.add_(1.0)
        ar.save_pretrained(out / f"checkpoint-{step}" / "ar_model_ema", extra_config=ema_scalars)
    (out / "logs").mkdir()
    (out / "ar_model").mkdir()  # the final save_pretrained target (train_AR.py:1200-1202) is not a checkpoint
    assert list_checkpoints(str(out)) == ["checkpoint-900", "checkpoint-1000", "checkpoint-10000"]
    latest = checkpoint_model_folder(str(out), "latest")
    assert latest == str(out / "checkpoint-10000" / "ar_model")
    assert checkpoint_model_folder(str(out), "/somewhere/else/checkpoint-1000", ema=True) == str(out / "checkpoint-1000" / "ar_model_ema")
    with pytest.raises(FileNotFoundError):
        checkpoint_model_folder(str(out), "checkpoint-5")
    with pytest.raises(FileNotFoundError):
        checkpoint_model_folder(str(tmp_path), "latest")

    ema_dir = checkpoint_model_folder(str(out), "latest", ema=True)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error")  # EMA scalars are expected extras: no warning, no TypeError from the constructor
        ema = M.from_pretrained(ema_dir)
    assert _same(ar, ema)
    init, rest = M.load_config(ema_dir, return_unused_kwargs=True)
    assert {k: rest[k] for k in EMA_CONFIG_KEYS} == ema_scalars and "num_layers" in init and "decay" not in init
    assert M.load_config(ema_dir)["optimization_step"] == 1200
    with pytest.warns(UserWarning, match="not constructor arguments"):
        M.from_config(dict(tiny_ar_config(), option_of_a_later_release=1))
