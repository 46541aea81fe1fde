"""Minimal diffusers-style model surface (``config``, ``from_config``, ``load_config``, ``from_pretrained``,
``save_pretrained``, ``dtype`` / ``device``) for the two models on the path.  Mirrors what the reference gets from
``diffusers.ModelMixin`` / ``ConfigMixin`` (models/LaDCast_3D_model.py:569-571,623; models/DCAE.py:735,781): a model
folder is ``config.json`` + ``diffusion_pytorch_model.safetensors`` (or its sharded / variant / ``.bin`` spellings).

Where the folder comes from, as the reference's call sites use it:

* a local directory - ``from_pretrained(args.ar_model_path)`` (evaluate/pred_rollout.py:306,320);
* a hub repository + subfolder - ``from_pretrained("tonyzyl/ladcast", subfolder="V0.1.X/DCAE")``
  (evaluate/pred_rollout.py:308-322, train_AR.py:516-518, evaluate_ens_gpu.py:138, track.py:767).  There is no network on
  the target machines, so the repository is resolved **only** in the local hub cache
  (``<cache>/models--tonyzyl--ladcast/refs/<revision>`` -> ``snapshots/<commit>/<subfolder>``), the layout
  ``huggingface_hub`` writes and ``huggingface-cli download`` fills; a miss is an ``OSError`` that says where it looked;
* a training output directory - ``<output_dir>/checkpoint-<step>/{ar_model,ar_model_ema}`` written by the
  save hook (train_AR.py:561-570,1131-1135), "latest" = highest step (train_AR.py:799-803); the EMA folder's
  ``config.json`` also carries the EMA bookkeeping scalars (train_AR.py:572-574: "contains ema_kwargs"), which the loader
  sets aside instead of passing them to the constructor.
"""
from __future__ import annotations

import inspect
import json
import os
import re
import warnings
from types import SimpleNamespace

import torch
import torch.nn as nn

# diffusers.EMAModel.state_dict() minus shadow_params: what EMAModel.save_pretrained registers into config.json
EMA_CONFIG_KEYS = ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power")
_REPO_ID = re.compile(r"^[\w.\-]+/[\w.\-]+$")


def hub_cache_dir(cache_dir=None):
    """The hub cache root, by huggingface_hub's precedence: argument, HF_HUB_CACHE, HUGGINGFACE_HUB_CACHE, HF_HOME/hub,
    XDG_CACHE_HOME/huggingface/hub, ~/.cache/huggingface/hub."""
    if cache_dir:
        return os.fspath(cache_dir)
    for var in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        if os.environ.get(var):
            return os.environ[var]
    if os.environ.get("HF_HOME"):
        return os.path.join(os.environ["HF_HOME"], "hub")
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    return os.path.join(base, "huggingface", "hub")


def resolve_model_folder(name_or_path, subfolder=None, revision=None, cache_dir=None):
    """Directory holding ``config.json`` for a local path or a cached hub repository (offline; never downloads)."""
    name = os.fspath(name_or_path)
    if os.path.isdir(name):
        folder = os.path.join(name, subfolder) if subfolder else name
        if not os.path.isdir(folder):
            raise OSError(f"{name!r} has no subfolder {subfolder!r}")
        return folder
    if not _REPO_ID.match(name):
        raise OSError(f"{name!r} is neither a directory nor a hub repository id ('org/name')")
    cache = hub_cache_dir(cache_dir)
    repo = os.path.join(cache, "models--" + name.replace("/", "--"))
    revision = revision or "main"
    commit = revision
    ref = os.path.join(repo, "refs", revision)
    if os.path.isfile(ref):
        with open(ref) as f:
            commit = f.read().strip()
    snap = os.path.join(repo, "snapshots", commit)
    folder = os.path.join(snap, subfolder) if subfolder else snap
    if not os.path.isdir(folder):
        raise OSError(
            f"hub repository {name!r} (revision {revision!r}, subfolder {subfolder!r}) is not in the local cache: looked for {folder!r}. "
            "This build never downloads; fetch the files on a connected machine (huggingface-cli download) and point HF_HOME / "
            "cache_dir at the cache, or pass the model directory itself."
        )
    return folder


def list_checkpoints(output_dir):
    """``checkpoint-<step>`` directories of a training output directory, ascending by step (train_AR.py:799-803)."""
    dirs = [d for d in os.listdir(output_dir) if d.startswith("checkpoint") and os.path.isdir(os.path.join(output_dir, d))]
    return sorted(dirs, key=lambda d: int(d.split("-")[1]))


def checkpoint_model_folder(output_dir, checkpoint="latest", ema=False):
    """``<output_dir>/<checkpoint>/ar_model`` or ``.../ar_model_ema`` (train_AR.py:561-570); ``checkpoint="latest"`` picks the
    highest step, any other value is reduced to its basename like the reference does (train_AR.py:797-798)."""
    if checkpoint == "latest":
        dirs = list_checkpoints(output_dir)
        if not dirs:
            raise FileNotFoundError(f"no checkpoint-* directory under {output_dir!r}")
        name = dirs[-1]
    else:
        name = os.path.basename(os.path.normpath(os.fspath(checkpoint)))
    folder = os.path.join(output_dir, name, "ar_model_ema" if ema else "ar_model")
    if not os.path.isdir(folder):
        raise FileNotFoundError(folder)
    return folder


def _weight_files(folder, stem, variant):
    """-> (kind, [files]) for the weight spellings diffusers 0.32 writes: single / sharded safetensors, then ``.bin``."""
    v = f".{variant}" if variant else ""
    single = os.path.join(folder, f"{stem}{v}.safetensors")
    if os.path.isfile(single):
        return "safetensors", [single], None
    index = os.path.join(folder, f"{stem}.safetensors.index{v}.json")
    if os.path.isfile(index):
        with open(index) as f:
            wmap = json.load(f)["weight_map"]
        files = sorted(set(wmap.values()))
        missing = [x for x in files if not os.path.isfile(os.path.join(folder, x))]
        if missing:
            raise OSError(f"{index} names shards that are not in {folder!r}: {missing}")
        return "safetensors", [os.path.join(folder, x) for x in files], wmap
    binf = os.path.join(folder, f"{stem}{v}.bin")
    if os.path.isfile(binf):
        return "bin", [binf], None
    raise OSError(f"no {stem}{v}.safetensors, {stem}.safetensors.index{v}.json or {stem}{v}.bin in {folder!r}")


class FrozenConfig(SimpleNamespace):
    def to_dict(self):
        return dict(vars(self))

    def get(self, k, default=None):
        return getattr(self, k, default)


class ModelMixin(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    def register_to_config(self, **kw):
        self.config = FrozenConfig(**kw)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def _upcast_to_fp32(self):
        """A model cast to bf16 / fp16 (`.to(torch.bfloat16)`, `from_pretrained(torch_dtype=torch.bfloat16)` - how the reference's mixed
        precision checkpoints are often held) keeps working: the kernels read fp32 parameter storage, so the parameters are up-cast
        once, with a warning; the arithmetic mode is chosen with `set_gemm_precision`, not with the parameter dtype."""
        import warnings

        dt = self.dtype
        if dt == torch.float32:
            return
        if dt not in (torch.bfloat16, torch.float16, torch.float64):
            raise NotImplementedError(f"parameters of dtype {dt} are not supported")
        warnings.warn(f"{type(self).__name__}: parameters are {dt}; up-casting them to float32 (the HIP kernels read fp32 parameter storage; "
                      "use set_gemm_precision('bf16') for reduced-precision arithmetic)", stacklevel=3)
        self.to(torch.float32)

    @classmethod
    def _ctor_keys(cls):
        return [k for k in inspect.signature(cls.__init__).parameters if k not in ("self", "args", "kwargs")]

    @classmethod
    def from_config(cls, config, return_unused_kwargs=False):
        """Build from a config dict: keys the constructor does not take (``_class_name``, ``_diffusers_version``, the EMA
        scalars, options of a later release) are set aside - diffusers ignores them too - and missing keys take the
        constructor's defaults."""
        if hasattr(config, "to_dict"):
            config = config.to_dict()
        config = dict(config)
        keys = cls._ctor_keys()
        init = {k: config[k] for k in keys if k in config}
        unused = {k: v for k, v in config.items() if k not in init}
        extra = [k for k in unused if not k.startswith("_") and k not in EMA_CONFIG_KEYS]
        if extra:
            warnings.warn(f"{cls.__name__}: config keys {extra} are not constructor arguments and were ignored", stacklevel=2)
        model = cls(**init)
        return (model, unused) if return_unused_kwargs else model

    @classmethod
    def load_config(cls, pretrained_model_name_or_path, subfolder=None, return_unused_kwargs=False, revision=None, cache_dir=None, **_unused):
        """The folder's ``config.json`` as a dict (``model_cls.load_config(path)``, train_AR.py:572-574).  With
        ``return_unused_kwargs`` -> ``(constructor part, rest)``; for an ``ar_model_ema`` folder the rest holds the EMA scalars."""
        folder = resolve_model_folder(pretrained_model_name_or_path, subfolder, revision, cache_dir)
        path = os.path.join(folder, cls.config_name)
        if not os.path.isfile(path):
            raise OSError(f"no {cls.config_name} in {folder!r}")
        with open(path) as f:
            cfg = json.load(f)
        if not return_unused_kwargs:
            return cfg
        keys = set(cls._ctor_keys())
        return {k: v for k, v in cfg.items() if k in keys}, {k: v for k, v in cfg.items() if k not in keys}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=None, variant=None, revision=None, cache_dir=None,
                        strict=True, **_unused):
        """Load ``config.json`` + weights from a directory or a locally cached hub repository (see the module docstring).

        Stricter than diffusers on purpose: a missing, unexpected or mis-shaped tensor is an error (diffusers 0.32 only warns on
        missing / unexpected keys and leaves such parameters at their random initialisation); ``strict=False`` restores the lenient
        behaviour.  Returns the model in ``eval()`` mode like ``ModelMixin.from_pretrained``."""
        folder = resolve_model_folder(pretrained_model_name_or_path, subfolder, revision, cache_dir)
        cfg = cls.load_config(folder)
        cls_name = cfg.get("_class_name")
        if cls_name is not None and cls_name != cls.__name__:
            warnings.warn(f"{folder}: config.json was written by {cls_name}, loading it as {cls.__name__}", stacklevel=2)
        model = cls.from_config(cfg)
        kind, files, wmap = _weight_files(folder, cls.weights_name[: -len(".safetensors")], variant)
        state = {}
        if kind == "safetensors":
            from safetensors.torch import load_file

            for f in files:
                part = load_file(f)
                if wmap is not None:
                    wrong = [k for k in part if wmap.get(k) != os.path.basename(f)]
                    if wrong:
                        raise OSError(f"{f}: tensors {wrong[:4]} are not assigned to this shard by the index")
                state.update(part)
            if wmap is not None and set(wmap) != set(state):
                raise OSError(f"{folder}: the shard index lists {len(wmap)} tensors, the shards hold {len(state)}")
        else:
            state = torch.load(files[0], map_location="cpu", weights_only=True)
        model.load_state_dict(state, strict=strict)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        model._name_or_path = os.fspath(pretrained_model_name_or_path)
        return model.eval()

    def save_pretrained(self, path, max_shard_size=None, variant=None, extra_config=None):
        """Write ``config.json`` + safetensors the way diffusers does (tuples as lists, ``_class_name``); ``max_shard_size``
        (bytes) switches to ``-0000k-of-0000n`` shards plus ``.safetensors.index.json``; ``extra_config`` is merged into
        ``config.json`` (what ``EMAModel.save_pretrained`` does with its scalars for the ``ar_model_ema`` folder)."""
        from safetensors.torch import save_file

        os.makedirs(path, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.to_dict().items()}
        cfg.update(extra_config or {})
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        state = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        stem = self.weights_name[: -len(".safetensors")]
        v = f".{variant}" if variant else ""
        if not max_shard_size:
            save_file(state, os.path.join(path, f"{stem}{v}.safetensors"), metadata={"format": "pt"})
            return
        shards, cur, cur_bytes = [], {}, 0
        for k, t in state.items():
            nbytes = t.numel() * t.element_size()
            if cur and cur_bytes + nbytes > max_shard_size:
                shards.append(cur)
                cur, cur_bytes = {}, 0
            cur[k] = t
            cur_bytes += nbytes
        shards.append(cur)
        wmap, total = {}, 0
        for i, shard in enumerate(shards):
            name = f"{stem}{v}-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(shard, os.path.join(path, name), metadata={"format": "pt"})
            for k, t in shard.items():
                wmap[k] = name
                total += t.numel() * t.element_size()
        with open(os.path.join(path, f"{stem}.safetensors.index{v}.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": wmap}, f, indent=2, sort_keys=True)

    def forward(self, *a, **k):  # pragma: no cover - subclasses implement
        raise NotImplementedError
