"""Minimal diffusers-style model surface (``config``, ``from_config``, ``from_pretrained``,
``dtype`` / ``device``) for the two models on the path.  Mirrors what the reference gets
from ``diffusers.ModelMixin`` / ``ConfigMixin`` (models/LaDCast_3D_model.py:569-571,623;
models/DCAE.py:735,781; loader use at evaluate/pred_rollout.py:299-324): a model folder is
``config.json`` + ``diffusion_pytorch_model.safetensors``.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn


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

    @classmethod
    def from_config(cls, cfg):
        if hasattr(cfg, "to_dict"):
            cfg = cfg.to_dict()
        return cls(**{k: v for k, v in dict(cfg).items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **_unused):
        from safetensors.torch import load_file

        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, cls.config_name)) as f:
            cfg = json.load(f)
        model = cls.from_config(cfg)
        model.load_state_dict(load_file(os.path.join(folder, cls.weights_name)), strict=True)
        return model.eval()

    def save_pretrained(self, path):
        from safetensors.torch import save_file

        os.makedirs(path, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.to_dict().items()}
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(path, self.weights_name))

    def forward(self, *a, **k):  # pragma: no cover - subclasses implement
        raise NotImplementedError
