"""YAML -> flat `args` namespace, the shape main.py:12-48 of the reference produces with OmegaConf
(omegaconf is not required: PyYAML + attribute dicts)."""
from __future__ import annotations

import argparse

import yaml


class AttrDict(dict):
    """dict with attribute access (enough of OmegaConf's DictConfig for `args.data_config.test_resolution` etc.)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def load_ldm_args(path: str, args: argparse.Namespace = None) -> argparse.Namespace:
    """Reads a reference-schema `configs/ldm/*.yaml` (model.params.{unetconfig,ddconfig,mlpconfig,ddpmconfig}, data.*).
    Training-only blocks (lossconfig, lr, ...) are optional."""
    with open(path) as f:
        cfg = _wrap(yaml.safe_load(f))
    args = args or argparse.Namespace()
    p = cfg.model.params
    args.data_config = cfg.data
    args.ddconfig, args.mlpconfig, args.unetconfig, args.ddpmconfig = p.ddconfig, p.mlpconfig, p.unetconfig, p.ddpmconfig
    args.loss_config = p.get("lossconfig")
    args.embed_dim = cfg.model.embed_dim
    args.resolution = p.ddconfig.resolution
    args.lr = cfg.model.get("lr", 1e-4)
    args.pretrained = cfg.model.get("pretrained", True)
    args.resume = cfg.model.get("resume", False)
    args.use_fp16 = cfg.model.get("use_fp16", True)
    args.domain = cfg.data.domain
    args.mode = cfg.data.mode
    args.DiT = cfg.model.get("DiT", False)
    return args
