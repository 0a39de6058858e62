"""SURVEY 8(f) rank 1 - checkpoint ingestion on the host (ddmi_amd/checkpoint.py): the three start modes of the reference's
trainers (tools/ldm/image.py:70-88, tools/ldm/occupancy.py:76-94), the EMA key mapping, and the restricted unpickler that
reads the published checkpoints' pickled `args` without omegaconf and without running anything the file names."""
import argparse
import enum
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from ddmi_amd import checkpoint as CK


@pytest.fixture
def fake_omegaconf():
    """A stand-in for the classes a real ldm-last.pt pickles inside `args` (omegaconf is not installed here): registered
    only while the checkpoint is WRITTEN, removed before it is read."""
    mod = types.ModuleType("omegaconf")
    sub = types.ModuleType("omegaconf.dictconfig")
    base = types.ModuleType("omegaconf.base")

    class DictConfig:
        def __init__(self, content):
            self.__dict__["_content"] = content

        def __getstate__(self):
            return {"_content": self._content, "_metadata": Metadata()}

        def __setstate__(self, st):
            self.__dict__.update(st)

    class Metadata:
        def __init__(self):
            self.ref_type = int
            self.flags = {"readonly": False}

    class SCMode(enum.Enum):
        DICT = 1

    DictConfig.__module__ = "omegaconf.dictconfig"
    DictConfig.__qualname__ = "DictConfig"
    Metadata.__module__ = "omegaconf.base"
    Metadata.__qualname__ = "Metadata"
    SCMode.__module__ = "omegaconf.base"
    SCMode.__qualname__ = "SCMode"
    sub.DictConfig, base.Metadata, base.SCMode = DictConfig, Metadata, SCMode
    mod.dictconfig, mod.base = sub, base
    sys.modules.update({"omegaconf": mod, "omegaconf.dictconfig": sub, "omegaconf.base": base})
    yield types.SimpleNamespace(DictConfig=DictConfig, SCMode=SCMode)
    for k in ("omegaconf", "omegaconf.dictconfig", "omegaconf.base"):
        sys.modules.pop(k, None)


def _tensors(prefix, n=3):
    g = torch.Generator().manual_seed(len(prefix))
    return {f"{prefix}{i}.weight": torch.randn(2, 3, generator=g) for i in range(n)}


def test_ldm_checkpoint_with_pickled_args_loads_without_omegaconf(tmp_path, fake_omegaconf):
    args = argparse.Namespace(lr=1e-4, unetconfig=fake_omegaconf.DictConfig({"model_channels": 224}),
                              mode=fake_omegaconf.SCMode.DICT, data_config=fake_omegaconf.DictConfig({"save_pth": "x"}))
    diffusion = {"model.a.weight": torch.ones(2), "mixing_logit": torch.full((1, 4, 1, 1), -6.0), "betas": torch.zeros(1000),
                 "lvlb_weights": torch.zeros(1000), "logvar": torch.zeros(1000)}
    ema = {"initted": torch.tensor(True), "step": torch.tensor(7)}
    ema.update({"online_model." + k: v for k, v in diffusion.items()})
    ema.update({"ema_model." + k: v * 2 for k, v in diffusion.items()})
    ck = {"args": args, "step": 3, "current_iters": 11, "vaemodel": _tensors("decoder."), "mlp": _tensors("net."),
          "diffusion": diffusion, "dae_opt": {"state": {0: {"exp_avg": torch.zeros(2)}}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.99)}]},
          "ema": ema, "scaler": None}
    path = os.path.join(tmp_path, "ldm-last.pt")
    torch.save(ck, path)
    for k in ("omegaconf", "omegaconf.dictconfig", "omegaconf.base"):   # the reader must not need them
        sys.modules.pop(k, None)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)         # torch's own safe loader refuses the file
    got = CK.read_checkpoint(path)
    assert got["step"] == 3 and got["current_iters"] == 11 and got["scaler"] is None
    assert isinstance(got["args"], argparse.Namespace) and got["args"].lr == 1e-4
    assert isinstance(got["args"].unetconfig, CK.Inert) and isinstance(got["args"].mode, CK.Inert)
    assert torch.equal(got["vaemodel"]["decoder.1.weight"], ck["vaemodel"]["decoder.1.weight"])
    assert got["dae_opt"]["param_groups"][0]["betas"] == (0.9, 0.99)
    st = CK.sampling_states(got)                       # EMA weights under `ema_model.` are the sampling weights
    assert st["kind"] == "ldm" and set(st["diffusion"]) == {"model.a.weight", "mixing_logit", "betas"}
    assert torch.equal(st["diffusion"]["model.a.weight"], torch.full((2,), 2.0))
    st2 = CK.sampling_states(got, use_ema=False)
    assert torch.equal(st2["diffusion"]["model.a.weight"], torch.ones(2))
    assert CK.checkpoint_sampler_config(got) == {"timesteps": 1000}
    assert CK.find_checkpoint(str(tmp_path), resume=False, pretrained=True) == path
    assert CK.find_checkpoint(str(tmp_path), resume=False, pretrained=False) is None   # that mode wants model-last.pt


@pytest.mark.parametrize("vae_key", ["model", "vaemodel"])
def test_stage1_checkpoint_layouts(tmp_path, vae_key):
    """tools/d2c_vae/image.py:120-133 saves the VAE under 'model'; the occupancy / video trainers under 'vaemodel' (+ 'pointnet')."""
    ck = {vae_key: {"module." + k: v for k, v in _tensors("decoder.").items()}, "mlp": _tensors("net."), "step": 1,
          "pointnet": _tensors("fc.")}
    path = os.path.join(tmp_path, "model-last.pt")
    torch.save(ck, path)
    st = CK.sampling_states(CK.read_checkpoint(path))
    assert st["kind"] == "stage1" and st["diffusion"] is None
    assert set(st["vae"]) == {"decoder.0.weight", "decoder.1.weight", "decoder.2.weight"}   # DDP 'module.' prefix removed
    assert CK.find_checkpoint(str(tmp_path), pretrained=False) == path


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned > /tmp/ddmi_pwned",))


def test_restricted_unpickler_never_calls_what_the_file_names(tmp_path):
    path = os.path.join(tmp_path, "ldm-last.pt")
    if os.path.exists("/tmp/ddmi_pwned"):
        os.remove("/tmp/ddmi_pwned")
    torch.save({"vaemodel": _tensors("decoder."), "mlp": _tensors("net."), "diffusion": {"betas": torch.zeros(10)}, "args": _Evil()}, path)
    got = CK.read_checkpoint(path)
    assert isinstance(got["args"], CK.Inert) and not os.path.exists("/tmp/ddmi_pwned")
    assert got["args"]._ddmi_origin[1] == "system"
    assert torch.equal(got["vaemodel"]["decoder.0.weight"], _tensors("decoder.")["decoder.0.weight"])
    with pytest.raises(pickle.PicklingError):
        pickle.dumps(got["args"])


def test_numpy_arrays_and_not_a_checkpoint(tmp_path):
    path = os.path.join(tmp_path, "x.pt")
    torch.save({"a": np.arange(4, dtype=np.float32), "b": np.float64(2.5)}, path)
    got = CK.read_checkpoint(path)
    assert np.array_equal(got["a"], np.arange(4, dtype=np.float32)) and float(got["b"]) == 2.5
    with pytest.raises(KeyError):
        CK.sampling_states(got)
