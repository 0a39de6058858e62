"""Image generation pipeline - the shape of tools/ldm/image.py:225-249 (`LDMTrainer.generate`) and of the 10k-sample
loop in evals/eval.py:215-226: sample(shape) -> decode -> mlp(coords, planes, si) -> clamp.

Construction mirrors exp/stage.py:194-203: the YAML blocks are splatted into the model constructors.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Sequence

import numpy as np

from . import initweights
from .autoencoder import Autoencoder
from .coords import generation_coords, get_scale_injection
from .ddpm import DDPM
from .dist import sample_seed
from .mlp import MLP
from .unet import UNetModel


class ImagePipeline:
    def __init__(self, unetconfig: Mapping, ddconfig: Mapping, mlpconfig: Mapping, ddpmconfig: Mapping, embed_dim: int,
                 test_resolution: int = 256, numerics: str = "bf16", max_batch: int = 8, device: str = "cuda:0"):
        self.device = device
        self.numerics = numerics
        self.max_batch = int(max_batch)
        self.unet = UNetModel(**dict(unetconfig), numerics=numerics, max_batch=max_batch)
        self.vae = Autoencoder(ddconfig, embed_dim, numerics=numerics, max_batch=max_batch)
        self.mlp = MLP(**dict(mlpconfig), numerics=numerics)
        self.ddpm = DDPM(model=self.unet, **dict(ddpmconfig))
        self.embed_dim = int(embed_dim)
        self.latent_size = int(ddpmconfig.get("image_size", unetconfig["image_size"]))
        self.test_resolution = int(test_resolution)
        self._coords = None

    # ---- weights -------------------------------------------------------------------------------------------
    def load_named_seed(self, tag: str = "afhq"):
        """Deterministic synthetic weights (no network for checkpoints): ddmi_amd.initweights."""
        self.unet.load_state_dict(initweights.fill_state(tag + ".unet", self.unet.manifest()))
        self.vae.load_state_dict(initweights.fill_state(tag + ".vae", self.vae.manifest()))
        self.mlp.load_state_dict(initweights.fill_state(tag + ".mlp", self.mlp.manifest()))
        self.ddpm.load_state_dict({"mixing_logit": initweights.fill_state(tag + ".ddpm", [("mixing_logit", (1, self.ddpm.channels, 1, 1))])["mixing_logit"]},
                                  strict=False)
        return self

    def load_checkpoint(self, path: str, use_ema: bool = True):
        """`ldm-last.pt` as written by tools/ldm/image.py:104-119: keys 'vaemodel', 'mlp', 'diffusion', 'ema'
        ('ema' = ema_pytorch state: the sampling weights live under the `ema_model.` prefix, tools/ldm/image.py:64,240)."""
        import torch
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.vae.load_state_dict(ck["vaemodel"], strict=True)
        self.mlp.load_state_dict(ck["mlp"], strict=True)
        diff = ck["diffusion"]
        if use_ema and "ema" in ck:
            ema = {k[len("ema_model."):]: v for k, v in ck["ema"].items() if k.startswith("ema_model.")}
            if ema:
                diff = ema
        diff = {k: v for k, v in diff.items() if k != "lvlb_weights" and k != "logvar"}
        self.ddpm.load_state_dict(diff, strict=True)
        return self

    def finalize(self):
        idx = int(str(self.device).split(":")[1]) if ":" in str(self.device) else 0
        self.unet.finalize(idx)
        self.vae.finalize(idx)
        self.mlp.finalize(idx)
        return self

    # ---- generation ------------------------------------------------------------------------------------------
    def coords(self):
        if self._coords is None:
            self._coords = generation_coords(self.test_resolution)
        return self._coords

    def generate(self, indices: Sequence[int], base_seed: int = 777, return_latent: bool = False, events=None,
                 rng: str = "philox"):
        """One batch of finished samples for the given GLOBAL sample indices.  Every chain draws x_T and its per-step
        noise from its own counter-based stream keyed by sample_seed(base_seed, index) (inside the sampler-step
        kernel), so results do not depend on how indices are sharded over GPUs or batched."""
        import torch
        B = len(indices)
        dev = torch.device(self.device)
        shape = (B, self.embed_dim, self.latent_size, self.latent_size)
        if rng == "philox":
            seeds = torch.tensor([sample_seed(base_seed, i) for i in indices], dtype=torch.int64, device=dev)
            z = self.ddpm.sample(list(shape), seeds=seeds, device=dev)
        else:  # rng == "torch": one torch device generator per chain (B small launches per step; kept for comparison)
            gens = []
            for i in indices:
                g = torch.Generator(device=dev)
                g.manual_seed(sample_seed(base_seed, i))
                gens.append(g)
            buf = torch.empty(shape, device=dev, dtype=torch.float32)

            def draw(_step):
                for j, g in enumerate(gens):
                    torch.randn(shape[1:], generator=g, out=buf[j])
                return buf

            z = self.ddpm.sample(list(shape), noise=draw(-1).clone(), step_noise=draw, device=dev)
        if events:
            events[0].record()
        planes = self.vae.decode(z)
        if events:
            events[1].record()
        img = self.mlp(self.coords(), hdbf=planes, si=get_scale_injection(self.test_resolution))
        img = img.clamp_(-1.0, 1.0)
        return (img, z) if return_latent else img


class VideoPipeline:
    """tools/ldm/video.py:214-236 (`LDMTrainer.generate`, video domain): sample([B, C, L]) -> VITAutoencoder.decode ->
    MLPVideo(coords, planes) -> clamp -> [B, 3, frames, R, R].  Construction mirrors exp/stage.py:245-262."""

    def __init__(self, unetconfig: Mapping, ddconfig: Mapping, mlpconfig: Mapping, ddpmconfig: Mapping, embed_dim: int,
                 frames: int = 16, test_resolution: int = 256, numerics: str = "bf16", max_batch: int = 2,
                 device: str = "cuda:0"):
        from .video import MLPVideo, UNetModel_Triplane, VITAutoencoder
        self.device = device
        self.numerics = numerics
        self.max_batch = int(max_batch)
        self.unet = UNetModel_Triplane(**dict(unetconfig), numerics=numerics, max_batch=max_batch)
        self.vae = VITAutoencoder(ddconfig, embed_dim, frames, numerics=numerics, max_batch=max_batch)
        self.mlp = MLPVideo(**dict(mlpconfig), numerics=numerics)
        cfg = dict(ddpmconfig)
        cfg["domain"] = "video"
        self.ddpm = DDPM(model=self.unet, **cfg)
        self.embed_dim = int(embed_dim)
        self.frames = int(frames)
        self.test_resolution = int(test_resolution)
        assert self.unet.seq_len == self.vae.seq_len, "unetconfig sizes and ddconfig resolution / frames disagree"
        self._coords = None

    def load_named_seed(self, tag: str = "sky"):
        self.unet.load_state_dict(initweights.fill_state(tag + ".unet3", self.unet.manifest()))
        self.vae.load_state_dict(initweights.fill_state(tag + ".vit", self.vae.manifest()))
        self.mlp.load_state_dict(initweights.fill_state(tag + ".mlpv", self.mlp.manifest()))
        self.ddpm.load_state_dict({"mixing_logit": initweights.fill_state(tag + ".ddpm", [("mixing_logit", (1, self.ddpm.channels, 1))])["mixing_logit"]},
                                  strict=False)
        return self

    def load_checkpoint(self, path: str, use_ema: bool = True):
        """`ldm-last.pt` as written by tools/ldm/video.py:93-109 (same keys as the image trainer)."""
        import torch
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.vae.load_state_dict(ck["vaemodel"], strict=False)   # encoder / TimeSformer entries are not on this path
        self.mlp.load_state_dict(ck["mlp"], strict=True)
        diff = ck["diffusion"]
        if use_ema and "ema" in ck:
            ema = {k[len("ema_model."):]: v for k, v in ck["ema"].items() if k.startswith("ema_model.")}
            if ema:
                diff = ema
        diff = {k: v for k, v in diff.items() if k != "lvlb_weights" and k != "logvar"}
        self.ddpm.load_state_dict(diff, strict=True)
        return self

    def finalize(self):
        idx = int(str(self.device).split(":")[1]) if ":" in str(self.device) else 0
        self.unet.finalize(idx)
        self.unet.set_uniform_t(True)
        self.vae.finalize(idx)
        self.mlp.finalize(idx)
        return self

    def coords(self):
        if self._coords is None:
            from .video import video_generation_coords
            self._coords = video_generation_coords(self.test_resolution, self.frames)
        return self._coords

    def generate(self, indices: Sequence[int], base_seed: int = 777, return_latent: bool = False):
        """Finished clips for the given GLOBAL sample indices; per-sample generators as in ImagePipeline.generate."""
        import torch
        B = len(indices)
        dev = torch.device(self.device)
        shape = (B, self.embed_dim, self.unet.seq_len)
        seeds = torch.tensor([sample_seed(base_seed, i) for i in indices], dtype=torch.int64, device=dev)
        z = self.ddpm.sample(list(shape), seeds=seeds, device=dev)
        outs = []
        for b0 in range(0, B, self.vae.max_batch):  # decode in slices: the 73k-token attention buffers are per sample
            zz = z[b0:b0 + self.vae.max_batch].contiguous()
            outs.append(self.mlp(self.coords(), self.vae.decode(zz)).clamp_(-1.0, 1.0))
        vid = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        return (vid, z) if return_latent else vid


class OccupancyPipeline:
    """tools/ldm/occupancy.py:205-230 (`LDMTrainer.eval` / `generate`, occupancy domain): sample([B, 3C, 16, 16]) -> per
    sample Generator3D.generate_mesh_fromdiffusion (Autoencoder3D.decode -> dense MLP3D logits -> marching cubes, all on the
    device) -> meshes.  Construction mirrors exp/stage.py:207-241; the generator settings are the ones get_generator
    (convocc/src/conv_onet/config.py:117-128) reads from the convocc yaml: test.threshold, generation.resolution_0,
    generation.upsampling_steps, data.padding."""

    def __init__(self, unetconfig: Mapping, ddconfig: Mapping, mlpconfig: Mapping, ddpmconfig: Mapping, embed_dim: int,
                 threshold: float = 0.2, resolution0: int = 256, upsampling_steps: int = 0, padding: float = 0.1,
                 numerics: str = "bf16", max_batch: int = 2, device: str = "cuda:0"):
        from .occupancy import MLP3D, Autoencoder3D, Generator3D
        self.device = device
        self.numerics = numerics
        self.max_batch = int(max_batch)
        ucfg = {k: v for k, v in dict(unetconfig).items() if k not in ("size1", "size2", "size3")}
        self.unet = UNetModel(**ucfg, numerics=numerics, max_batch=max_batch)
        self.vae = Autoencoder3D(ddconfig, embed_dim, numerics=numerics, max_batch=1)
        self.mlp = MLP3D(**dict(mlpconfig), numerics=numerics)
        cfg = {k: v for k, v in dict(ddpmconfig).items() if k != "domain"}   # 4-D latents: the image form of the sampler
        self.ddpm = DDPM(model=self.unet, **cfg)
        self.mesh_gen = Generator3D(threshold=threshold, resolution0=resolution0, upsampling_steps=upsampling_steps,
                                    padding=padding)
        self.latent_shape = (int(dict(unetconfig)["in_channels"]), int(dict(unetconfig)["image_size"]),
                             int(dict(unetconfig)["image_size"]))

    def load_named_seed(self, tag: str = "shapenet"):
        self.unet.load_state_dict(initweights.fill_state(tag + ".unet", self.unet.manifest()))
        self.vae.load_state_dict(initweights.fill_state(tag + ".vae", self.vae.manifest()))
        self.mlp.load_state_dict(initweights.fill_state(tag + ".mlp", self.mlp.manifest()))
        self.ddpm.load_state_dict({"mixing_logit": initweights.fill_state(tag + ".ddpm", [("mixing_logit", (1, self.ddpm.channels, 1, 1))])["mixing_logit"]},
                                  strict=False)
        return self

    def load_checkpoint(self, path: str, use_ema: bool = True):
        """`ldm-last.pt` of tools/ldm/occupancy.py (same keys as the image trainer; the PointNet encoder is not on this path)."""
        import torch
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.vae.load_state_dict(ck["vaemodel"], strict=False)
        self.mlp.load_state_dict(ck["mlp"], strict=True)
        diff = ck["diffusion"]
        if use_ema and "ema" in ck:
            ema = {k[len("ema_model."):]: v for k, v in ck["ema"].items() if k.startswith("ema_model.")}
            if ema:
                diff = ema
        diff = {k: v for k, v in diff.items() if k != "lvlb_weights" and k != "logvar"}
        self.ddpm.load_state_dict(diff, strict=True)
        return self

    def finalize(self):
        idx = int(str(self.device).split(":")[1]) if ":" in str(self.device) else 0
        self.unet.finalize(idx)
        self.unet.set_uniform_t(True)
        self.vae.finalize(idx)
        self.mlp.finalize(idx)
        return self

    def generate(self, indices: Sequence[int], base_seed: int = 777, return_latent: bool = False):
        """One TriMesh per GLOBAL sample index (per-sample noise streams, so the set does not depend on batching or on the
        number of GPUs); meshes are extracted one latent at a time, as the reference does (tools/ldm/occupancy.py:216-217)."""
        import torch
        B = len(indices)
        dev = torch.device(self.device)
        seeds = torch.tensor([sample_seed(base_seed, i) for i in indices], dtype=torch.int64, device=dev)
        z = self.ddpm.sample([B, *self.latent_shape], seeds=seeds, device=dev)
        meshes = [self.mesh_gen.generate_mesh_fromdiffusion(z[j:j + 1].contiguous(), self.vae, self.mlp, dev)[0] for j in range(B)]
        return (meshes, z) if return_latent else meshes
