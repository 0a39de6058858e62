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


RNG_MODES = ("philox", "torch", "reference")


def _load_checkpoint_into(pipe, path, use_ema: bool, vae_strict: bool):
    """Shared by the three pipelines: see ddmi_amd/checkpoint.py for the file layouts (tools/ldm/image.py:70-88,
    tools/ldm/video.py:68-83, tools/ldm/occupancy.py:76-94).  `path` is a file name or the dict `read_checkpoint` returned for
    it (multi-GB files are unpickled once per process).  Returns 'ldm' or 'stage1'."""
    from . import checkpoint as CK
    st = CK.sampling_states(path if isinstance(path, dict) else CK.read_checkpoint(path), use_ema=use_ema)
    pipe.vae.load_state_dict(st["vae"], strict=vae_strict)
    pipe.mlp.load_state_dict(st["mlp"], strict=True)
    if st["diffusion"] is not None:
        pipe.ddpm.load_state_dict(st["diffusion"], strict=True)
    return st["kind"]


def reference_draws(shape, device):
    """rng='reference' (SURVEY 8a a24): the noise of the reference's PyTorch-CPU path, `torch.randn(shape)` from the global
    CPU generator in the reference's draw order - x_T (ddpm.py:288 / :343), then one draw per step (ddpm.py:278 via
    noise_like; ddpm.py:363 randn_like on every non-final DDIM step, also when eta == 0).  Returns (x_T, step_noise)."""
    import torch
    x_T = torch.randn(tuple(shape))
    return x_T.to(device), (lambda i: torch.randn(tuple(shape)))


class ImagePipeline:
    def __init__(self, unetconfig: Mapping, ddconfig: Mapping, mlpconfig: Mapping, ddpmconfig: Mapping, embed_dim: int,
                 test_resolution: int = 256, numerics: str = "bf16", max_batch: int = 8, device: str = "cuda:0"):
        self.device = device
        self.numerics = numerics
        self.max_batch = int(max_batch)
        self.latent_size = int(ddpmconfig.get("image_size", unetconfig["image_size"]))
        # the reference's UNet is fully convolutional and never reads `image_size` in forward; the native handle plans its
        # arena for one resolution, so it is built for the latent the sampler will feed it (ddpmconfig.image_size) - e.g.
        # BASELINE config 1 runs the AFHQ nets (image_size 64) on a 16x16 latent
        self.unet = UNetModel(**dict(unetconfig, image_size=self.latent_size), numerics=numerics, max_batch=max_batch)
        self.vae = Autoencoder(ddconfig, embed_dim, numerics=numerics, max_batch=max_batch, latent_size=self.latent_size)
        self.mlp = MLP(**dict(mlpconfig), numerics=numerics)
        self.ddpm = DDPM(model=self.unet, **dict(ddpmconfig))
        self.embed_dim = int(embed_dim)
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
        """`ldm-last.pt` (resume / pretrained, tools/ldm/image.py:70-82) or the stage-1 `model-last.pt`
        (tools/ldm/image.py:83-88: VAE + MLP only, the diffusion model keeps its current weights).  Read through the
        restricted unpickler of ddmi_amd/checkpoint.py (the pickled `args` needs no omegaconf)."""
        return _load_checkpoint_into(self, path, use_ema, vae_strict=True) and self

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
                 rng: str = "philox", reseed: bool = True):
        """One batch of finished samples for the given GLOBAL sample indices.
        rng='philox' (default): every chain draws x_T, its per-step noise and its NoiseInjection noise from its own
        counter-based stream keyed by sample_seed(base_seed, index) inside the kernels, so results do not depend on how
        indices are sharded over GPUs or batched.
        rng='reference': the reference's own noise - `torch.manual_seed(base_seed)` (utils/general_utils.py:16-23; skipped
        with reseed=False to continue the stream as consecutive generate() calls of one reference process would) and
        CPU `randn` in the reference's order: x_T, one per step, then the 12 NoiseInjection draws [B,1,R,R]
        (blocks.py:293-295).  Same seed -> same samples as the reference's PyTorch-CPU path; depends on the batch."""
        import torch
        assert rng in RNG_MODES, rng
        B = len(indices)
        dev = torch.device(self.device)
        shape = (B, self.embed_dim, self.latent_size, self.latent_size)
        seeds = None
        if rng == "philox":
            seeds = torch.tensor([sample_seed(base_seed, i) for i in indices], dtype=torch.int64, device=dev)
            z = self.ddpm.sample(list(shape), seeds=seeds, device=dev)
        elif rng == "reference":
            if reseed:
                torch.manual_seed(int(base_seed))
            x_T, draw = reference_draws(shape, dev)
            z = self.ddpm.sample(list(shape), noise=x_T, step_noise=draw, device=dev)
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
        R = self.test_resolution
        inj = None
        if rng == "reference":   # the reference draws these whether or not noise.weight is zero (blocks.py:292-295)
            inj = torch.stack([torch.empty(B, 1, R, R).normal_() for _ in range(12)]).reshape(12, B, R * R)
            if not self.mlp.noise_nonzero:
                inj = None       # consumed from the generator, not needed by the kernel
        img = self.mlp(self.coords(), hdbf=planes, si=get_scale_injection(R), noise=inj,
                       seeds=seeds if inj is None else None)
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
        """`ldm-last.pt` (tools/ldm/video.py:68-79, written by :93-109) or the stage-1 `model-last.pt` (:80-82); the
        encoder / TimeSformer entries of 'vaemodel' are not on this path."""
        return _load_checkpoint_into(self, path, use_ema, vae_strict=False) and self

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

    def generate(self, indices: Sequence[int], base_seed: int = 777, return_latent: bool = False, rng: str = "philox",
                 reseed: bool = True):
        """Finished clips for the given GLOBAL sample indices; `rng` as in ImagePipeline.generate (MLPVideo has no
        NoiseInjection, so the reference draws x_T and the per-step noise only, tools/ldm/video.py:214-236)."""
        import torch
        assert rng in ("philox", "reference"), rng
        B = len(indices)
        dev = torch.device(self.device)
        shape = (B, self.embed_dim, self.unet.seq_len)
        if rng == "reference":
            if reseed:
                torch.manual_seed(int(base_seed))
            x_T, draw = reference_draws(shape, dev)
            z = self.ddpm.sample(list(shape), noise=x_T, step_noise=draw, device=dev)
        else:
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
                 numerics: str = "bf16", max_batch: int = 2, device: str = "cuda:0", decode_batch: Optional[int] = None):
        from .occupancy import MLP3D, Autoencoder3D, Generator3D
        self.device = device
        self.numerics = numerics
        self.max_batch = int(max_batch)
        # Meshes are extracted one latent at a time as in the reference (tools/ldm/occupancy.py:216-217), but the tri-plane DECODER may
        # run on several latents at once: 1.91 ms per shape at batch 1 (its ~60 launches are latency-bound) against 0.26 ms at batch 32
        # (tools/bench_occ_decode.py).  The parity mode keeps the reference's batch of 1.
        if decode_batch is None:
            decode_batch = 32 if numerics == "bf16" else 1
        self.decode_batch = max(1, min(int(decode_batch), self.max_batch))
        ucfg = {k: v for k, v in dict(unetconfig).items() if k not in ("size1", "size2", "size3")}
        self.unet = UNetModel(**ucfg, numerics=numerics, max_batch=max_batch)
        self.vae = Autoencoder3D(ddconfig, embed_dim, numerics=numerics, max_batch=self.decode_batch)
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
        """`ldm-last.pt` (tools/ldm/occupancy.py:76-87) or the stage-1 `model-last.pt` (:88-94: 'vaemodel', 'mlp',
        'pointnet'); the PointNet encoder is not on this path."""
        return _load_checkpoint_into(self, path, use_ema, vae_strict=False) and self

    def finalize(self):
        idx = int(str(self.device).split(":")[1]) if ":" in str(self.device) else 0
        self.unet.finalize(idx)
        self.unet.set_uniform_t(True)
        self.vae.finalize(idx)
        self.mlp.finalize(idx)
        return self

    def generate(self, indices: Sequence[int], base_seed: int = 777, return_latent: bool = False, rng: str = "philox",
                 reseed: bool = True, sink=None):
        """One TriMesh per GLOBAL sample index (per-sample noise streams, so the set does not depend on batching or on the
        number of GPUs); meshes are extracted one latent at a time, as the reference does (tools/ldm/occupancy.py:216-217).
        `rng` as in ImagePipeline.generate.  `sink`: a callable that takes each finished mesh (in index order) instead of the
        returned list - the way to run thousands of chains per call without holding every mesh in host memory (the reference
        writes each .obj as it is made); the list returned is then empty."""
        import torch
        assert rng in ("philox", "reference"), rng
        B = len(indices)
        dev = torch.device(self.device)
        if rng == "reference":
            if reseed:
                torch.manual_seed(int(base_seed))
            x_T, draw = reference_draws((B, *self.latent_shape), dev)
            z = self.ddpm.sample([B, *self.latent_shape], noise=x_T, step_noise=draw, device=dev)
        else:
            seeds = torch.tensor([sample_seed(base_seed, i) for i in indices], dtype=torch.int64, device=dev)
            z = self.ddpm.sample([B, *self.latent_shape], seeds=seeds, device=dev)
        meshes = []
        # One latent at a time, software-pipelined over two streams: the marching cubes of shape j (small kernels, two host
        # round-trips: the triangle count, then the mesh itself) run on a side stream under the 3 ms MLP3D launch of shape j + 1.
        from .occupancy import occupancy_grid_logits
        main, side = torch.cuda.current_stream(dev), (torch.cuda.Stream(dev) if self.decode_batch > 1 else None)
        pending = None

        def finish(p):
            grid, ready = p
            if side is None:
                return self.mesh_gen.mesh_from_grid(grid)[0]
            side.wait_event(ready)
            with torch.cuda.stream(side):
                grid.record_stream(side)
                return self.mesh_gen.mesh_from_grid(grid)[0]

        keep = meshes.append if sink is None else sink
        for j0 in range(0, B, self.decode_batch):
            planes = self.vae.decode(z[j0:j0 + self.decode_batch].contiguous())
            for j in range(planes[0][0].shape[0]):
                one = tuple([t[j:j + 1] for t in lst] for lst in planes)      # (a batch slice of an NHWC tensor is an NHWC tensor)
                grid = occupancy_grid_logits(self.mlp, one, self.mesh_gen.resolution0, self.mesh_gen.padding)
                ready = torch.cuda.Event()
                ready.record(main)
                if pending is not None:
                    keep(finish(pending))
                pending = (grid, ready)
        if pending is not None:
            keep(finish(pending))
        return (meshes, z) if return_latent else meshes
