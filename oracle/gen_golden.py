#!/usr/bin/env python3
"""Capture golden input/output vectors from the REAL reference (mlvlab/DDMI at /root/reference).

Runs ONLY in the build container (the reference does not exist on the GPU box); writes small
`.npz` / `.json` fixtures to `tests/golden/`.  Nothing of the reference itself (source, bytecode,
pickled modules) is written - only arrays.  Weights are never stored: they are regenerated from
`ddmi_amd.initweights` (named-seed fill) on both sides, and only `(key, shape)` manifests are kept.

Import recipe: SURVEY.md Appendix C - torchvision is absent (stubbed, only imported at module
top), `models/d2c_vae/op` JIT-compiles CUDA at import (replaced by its documented semantics,
fused_bias_act_kernel.cu:26-47 act=3 grad=0).  xformers is NOT stubbed, so the reference's own
`AttnBlock` (autoencoder_unet.py:158-215) is what runs.

Usage:  python oracle/gen_golden.py            (about a minute on 8 cores)
"""
import contextlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get("DDMI_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from ddmi_amd import initweights as iw  # noqa: E402


def install_stubs():
    sys.path.insert(0, REF)
    tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
    tvt, tvtf = types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    tvu.make_grid = tvu.save_image = lambda *a, **k: None
    tv.utils, tv.transforms, tvt.functional = tvu, tvt, tvtf
    sys.modules.update({"torchvision": tv, "torchvision.utils": tvu, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvtf})
    op = types.ModuleType("models.d2c_vae.op")

    def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
        return F.leaky_relu(x + bias.view(1, -1, *([1] * (x.ndim - 2))), negative_slope) * scale

    class FusedLeakyReLU(nn.Module):
        def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(channel))
            self.negative_slope, self.scale = negative_slope, scale

        def forward(self, x):
            return fused_leaky_relu(x, self.bias, self.negative_slope, self.scale)

    op.fused_leaky_relu, op.FusedLeakyReLU, op.upfirdn2d = fused_leaky_relu, FusedLeakyReLU, None
    sys.modules["models.d2c_vae.op"] = op


class DrawQueue:
    """Replaces torch.randn / torch.randn_like / Tensor.normal_ by pops from a list (oracle process only),
    so the reference's internal draws (ddpm.py:278,288,363; blocks.py:293-295) become explicit inputs."""

    def __init__(self, arrays):
        self.arrays = [torch.from_numpy(np.asarray(a, dtype=np.float32)) for a in arrays]
        self.pos = 0

    def pop(self, shape):
        a = self.arrays[self.pos]
        self.pos += 1
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return a.clone()

    @contextlib.contextmanager
    def patched(self):
        o_randn, o_like, o_normal = torch.randn, torch.randn_like, torch.Tensor.normal_

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            return self.pop(size)

        def randn_like(x, **kw):
            return self.pop(x.shape)

        def normal_(t, *a, **k):
            t.copy_(self.pop(t.shape))
            return t

        torch.randn, torch.randn_like, torch.Tensor.normal_ = randn, randn_like, normal_
        try:
            yield self
        finally:
            torch.randn, torch.randn_like, torch.Tensor.normal_ = o_randn, o_like, o_normal


def load_named(module, tag, overrides=None):
    sd = module.state_dict()
    filled = iw.fill_like(tag, sd)
    if overrides:
        for k, fn in overrides.items():
            for key in filled:
                if key.endswith(k):
                    filled[key] = fn(key, filled[key])
    module.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    module.eval()
    return {k: list(v.shape) for k, v in filled.items()}


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


TOY_UNET = dict(image_size=8, in_channels=64, model_channels=32, out_channels=64, attention_resolutions=[2],
                num_res_blocks=1, channel_mult=[1, 2], num_head_channels=32)
# second topology: three levels, attention at every level incl. ds=1, two res blocks, non-pow2 width (like 224=7*32)
TOY_UNET_B = dict(image_size=16, in_channels=16, model_channels=96, out_channels=16, attention_resolutions=[1, 2, 4],
                  num_res_blocks=2, channel_mult=[1, 2, 3], num_head_channels=32)
TOY_DD = dict(double_z=True, z_channels=32, resolution=32, in_channels=3, out_ch=64, ch=32, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_resolutions=[], hdbf_resolutions=[16, 8], dropout=0.0, attn_type="vanilla")
TOY_EMBED = 64
MLP_CFG = dict(in_ch=2, out_ch=3, ch=256, latent_dim=64)
# occupancy (config 4) toys: tri-plane decoder on a 4x4 latent -> planes 4^2, 8^2, 16^2; MLP3D at its real size
TOY_DD3 = dict(double_z=True, z_channels=32, resolution=16, in_channels=32, out_ch=64, ch=32, ch_mult=[1, 2, 4],
               num_res_blocks=2, attn_resolutions=[], hdbf_resolutions=[8, 4], inter_attn_resolutions=[16, 8, 4],
               dropout=0.0, attn_type="vanilla")
TOY_EMBED3 = 16
MLP3D_CFG = dict(in_ch=3, out_ch=1, ch=256, latent_dim=64)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    from diffusion.ddpm import DDPM
    from models.ldm.modules.diffusionmodules.openaimodel import (UNetModel, ResBlock, AttentionBlock,
                                                                   QKVAttentionLegacy, Downsample, Upsample)
    from models.ldm.modules.diffusionmodules.util import timestep_embedding, normalization
    from models.d2c_vae.autoencoder_unet import Autoencoder
    from models.d2c_vae.mlp import MLP
    from utils.general_utils import convert_to_coord_format_2d, get_scale_injection, singleplane_positional_encoding

    manifests = {}

    class Fixed(nn.Module):  # denoiser stub with the reference's call contract (ddpm.py:258,305)
        def __init__(self, out):
            super().__init__()
            self.out = out

        def forward(self, x, t, cond=None, x_self_cond=None, enable_mask=False):
            return self.out

    # ---------------- S1: schedules --------------------------------------------------------------
    s1 = {}
    for T in (1000, 250, 10):
        d = DDPM(model=Fixed(None), timesteps=T, sampling_timesteps=T, channels=4)
        for k, v in d.state_dict().items():
            if k != "mixing_logit":
                s1[f"T{T}.{k}"] = np32(v)
    pairs = {}
    for T, S in [(1000, s) for s in (1, 2, 7, 10, 30, 50, 100, 200, 250, 333, 500, 999)] + [(250, 50), (250, 249), (100, 99)]:
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        pairs[f"{T}_{S}"] = list(zip(times[:-1], times[1:]))
    np.savez(os.path.join(OUT, "S1_schedule.npz"), **s1)
    with open(os.path.join(OUT, "S1_time_pairs.json"), "w") as f:
        json.dump(pairs, f)

    # ---------------- K1: single step arithmetic -------------------------------------------------
    k1 = {}
    shape = (2, 8, 6, 6)
    x = iw.normal("K1", "x", shape)
    mo = iw.normal("K1", "model_out", shape)
    nz = iw.normal("K1", "noise", shape)
    logit = iw.normal("K1", "mixing_logit", (1, 8, 1, 1), 0.0, 1.5)
    k1.update(x=x, model_out=mo, noise=nz, mixing_logit=logit)
    for T in (1000, 250):
        d = DDPM(model=Fixed(torch.from_numpy(mo)), timesteps=T, sampling_timesteps=T, channels=8, clip_denoised=False)
        d.mixing_logit.data.copy_(torch.from_numpy(logit))
        for t in (0, 1, T // 2, T - 2, T - 1):
            with DrawQueue([nz]).patched():
                out = d.p_sample(torch.from_numpy(x), None, torch.full((2,), t, dtype=torch.long), clip_denoised=False)
            k1[f"anc.T{T}.t{t}"] = np32(out)
    # DDIM single steps via ddim_sample with a 1-entry schedule is awkward; drive model_predictions + the formula
    # through the reference's own ddim_sample by choosing S so the wanted pair is the first pair, and stopping after
    # one step is not possible -> capture full 3-step chains instead (first pair pinned by times below).
    for eta in (0.0, 1.0, 0.5):
        d = DDPM(model=Fixed(torch.from_numpy(mo)), timesteps=1000, sampling_timesteps=3, ddim_sampling_eta=eta, channels=8)
        d.mixing_logit.data.copy_(torch.from_numpy(logit))
        n1 = iw.normal("K1", "ddim_n1", shape)
        n2 = iw.normal("K1", "ddim_n2", shape)
        with DrawQueue([n1, n2]).patched():
            out = d.sample(shape=list(shape), noise=torch.from_numpy(x))
        k1[f"ddim3.eta{eta}"] = np32(out)
        k1["ddim_n1"], k1["ddim_n2"] = n1, n2
    # clip_denoised variant (ancestral)
    d = DDPM(model=Fixed(torch.from_numpy(mo)), timesteps=1000, sampling_timesteps=1000, channels=8)
    d.mixing_logit.data.copy_(torch.from_numpy(logit))
    with DrawQueue([nz]).patched():
        k1["anc_clip.T1000.t500"] = np32(d.p_sample(torch.from_numpy(x), None, torch.full((2,), 500, dtype=torch.long),
                                                    clip_denoised=True))
    np.savez(os.path.join(OUT, "K1_step.npz"), **k1)

    # ---------------- U1: UNet pieces ------------------------------------------------------------
    u1 = {}
    for dim in (224, 32):
        tt = torch.tensor([0, 1, 500, 999])
        u1[f"temb{dim}"] = np32(timestep_embedding(tt, dim))
    xin = iw.normal("U1", "x", (2, 64, 8, 8), 0.3, 1.7)
    u1["x"] = xin
    gn = normalization(64)
    manifests["U1.gn"] = load_named(gn, "U1.gn")
    u1["gn_silu"] = np32(F.silu(gn(torch.from_numpy(xin))))
    u1["gn"] = np32(gn(torch.from_numpy(xin)))
    qkv = iw.normal("U1", "qkv", (2, 2 * 3 * 32, 64))
    u1["qkv"] = qkv
    u1["attn_legacy"] = np32(QKVAttentionLegacy(2)(torch.from_numpy(qkv)))
    emb = iw.normal("U1", "emb", (2, 128))
    u1["emb"] = emb
    rb = ResBlock(64, 128, 0.0, out_channels=96, dims=2)
    manifests["U1.res"] = load_named(rb, "U1.res")
    u1["res"] = np32(rb(torch.from_numpy(xin), torch.from_numpy(emb)))
    rb2 = ResBlock(64, 128, 0.0, out_channels=64, dims=2)
    manifests["U1.res_same"] = load_named(rb2, "U1.res_same")
    u1["res_same"] = np32(rb2(torch.from_numpy(xin), torch.from_numpy(emb)))
    ab = AttentionBlock(64, num_heads=2, num_head_channels=32)
    manifests["U1.attn"] = load_named(ab, "U1.attn")
    u1["attn"] = np32(ab(torch.from_numpy(xin)))
    dn = Downsample(64, True, dims=2, out_channels=64)
    manifests["U1.down"] = load_named(dn, "U1.down")
    u1["down"] = np32(dn(torch.from_numpy(xin)))
    up = Upsample(64, True, dims=2, out_channels=64)
    manifests["U1.up"] = load_named(up, "U1.up")
    u1["up"] = np32(up(torch.from_numpy(xin)))
    np.savez(os.path.join(OUT, "U1_unet_ops.npz"), **u1)

    # ---------------- U2: toy UNet forwards ------------------------------------------------------
    u2 = {}
    for name, cfg, hw, tvals in (("a", TOY_UNET, 8, [999, 3]), ("b", TOY_UNET_B, 16, [0, 640])):
        unet = UNetModel(**cfg)
        manifests[f"U2{name}.unet"] = load_named(unet, f"U2{name}.unet")
        x = iw.normal(f"U2{name}", "x", (2, cfg["in_channels"], hw, hw))
        t = torch.tensor(tvals, dtype=torch.long)
        u2[f"{name}.x"], u2[f"{name}.t"] = x, t.numpy()
        u2[f"{name}.eps"] = np32(unet(torch.from_numpy(x), t))
    np.savez(os.path.join(OUT, "U2_unet.npz"), **u2)

    # ---------------- D1: toy decoder ------------------------------------------------------------
    ae = Autoencoder(ddconfig=TOY_DD, embed_dim=TOY_EMBED)
    full = load_named(ae, "D1.vae")
    manifests["D1.vae"] = {k: v for k, v in full.items() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    z = iw.normal("D1", "z", (2, TOY_EMBED, 8, 8))
    planes = ae.decode(torch.from_numpy(z))
    d1 = {"z": z}
    for i, pl in enumerate(planes):
        d1[f"plane{i}"] = np32(pl)
    np.savez(os.path.join(OUT, "D1_decoder.npz"), **d1)

    # ---------------- M1: image coordinate MLP ---------------------------------------------------
    mlp = MLP(**MLP_CFG)
    manifests["M1.mlp"] = load_named(mlp, "M1.mlp")
    m1 = {}
    for R in (32, 48, 8):
        e = (R - 1) / R
        coords = convert_to_coord_format_2d(1, R, R, hstart=-e, hend=e, wstart=-e, wend=e)
        si = get_scale_injection(R)
        m1[f"coords{R}"] = np32(coords)
        m1[f"si{R}"] = np.float64(si)
        with DrawQueue([np.zeros((2, 1, R, R), np.float32)] * 12).patched():
            m1[f"rgb{R}"] = np32(mlp(coords, hdbf=planes, si=si).detach())
    # plane sampling on its own (gu:122-123), incl. out-of-range coordinates -> border clamp
    gc = iw.normal("M1", "grid", (2, 5, 7, 2), 0.0, 0.8)
    m1["grid"] = gc
    for i, pl in enumerate(planes):
        m1[f"sample{i}"] = np32(singleplane_positional_encoding(pl, torch.from_numpy(gc)))
    # NoiseInjection with non-zero weights: draws become explicit inputs
    mlp2 = MLP(**MLP_CFG)
    manifests["M1n.mlp"] = load_named(mlp2, "M1n.mlp", overrides={
        "noise.weight": lambda key, v: iw.normal("M1n.mlp", key, v.shape, 0.0, 0.1)})
    R = 32
    e = (R - 1) / R
    coords = convert_to_coord_format_2d(1, R, R, hstart=-e, hend=e, wstart=-e, wend=e)
    draws = [iw.normal("M1n", f"draw{i}", (2, 1, R, R)) for i in range(12)]
    with DrawQueue(draws).patched():
        m1["rgb32_noise"] = np32(mlp2(coords, hdbf=planes, si=get_scale_injection(R)).detach())
    np.savez(os.path.join(OUT, "M1_mlp.npz"), **m1)

    # ---------------- G1: toy generate() chains --------------------------------------------------
    unet = UNetModel(**TOY_UNET)
    load_named(unet, "U2a.unet")
    g1 = {}
    B = 2
    shape = [B, 64, 8, 8]
    xT = iw.normal("G1", "x_T", shape)
    g1["x_T"] = xT
    R = 32
    e = (R - 1) / R
    coords = convert_to_coord_format_2d(1, R, R, hstart=-e, hend=e, wstart=-e, wend=e)
    for name, T, S, eta in (("ddim_eta0", 1000, 10, 0.0), ("ddim_eta1", 1000, 10, 1.0), ("anc", 10, 10, 1.0),
                            ("anc250", 250, 250, 1.0)):
        d = DDPM(model=unet, timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=64, image_size=8)
        manifests.setdefault("G1.ddpm_extra", {"mixing_logit": [1, 64, 1, 1]})
        d.mixing_logit.data.copy_(torch.from_numpy(iw.fill_state("G1.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"]))
        ndraw = S - 1 if S < T else T
        draws = [iw.normal("G1", f"{name}.n{i}", shape) for i in range(ndraw)]
        with DrawQueue(draws).patched():
            zz = d.sample(shape=shape, noise=torch.from_numpy(xT))
        g1[f"{name}.z"] = np32(zz)
        if name != "anc250":
            pl = ae.decode(zz)
            with DrawQueue([np.zeros((B, 1, R, R), np.float32)] * 12).patched():
                img = mlp(coords, hdbf=pl, si=get_scale_injection(R)).clamp(-1.0, 1.0)
            g1[f"{name}.img"] = np32(img)
    np.savez(os.path.join(OUT, "G1_generate.npz"), **g1)

    # ---------------- O1: occupancy path (Autoencoder3D.decode, MLP3D, query grid) ----------------
    from models.d2c_vae.autoencoder_unet import Autoencoder3D
    from models.d2c_vae.mlp import MLP3D
    ae3 = Autoencoder3D(ddconfig=TOY_DD3, embed_dim=TOY_EMBED3)
    full3 = load_named(ae3, "O1.vae3d")
    manifests["O1.vae3d"] = {k: v for k, v in full3.items() if k.startswith("decoder.") or k.startswith("post_quant_conv")}
    z3 = iw.normal("O1", "z", (2, 3 * TOY_EMBED3, 4, 4))
    xy, yz, xz = ae3.decode(torch.from_numpy(z3))
    o1 = {"z": z3}
    for name, lst in (("xy", xy), ("yz", yz), ("xz", xz)):
        for i, pl in enumerate(lst):
            o1[f"{name}{i}"] = np32(pl)
    mlp3 = MLP3D(**MLP3D_CFG)
    manifests["O1.mlp3d"] = load_named(mlp3, "O1.mlp3d")
    pts = iw.normal("O1", "points", (2, 4096, 3), 0.0, 0.35)      # |p| up to ~1.5: exercises both clamp branches
    pts[0, :8] = np.array([[0.55, -0.55, 0.0], [0.6, 0.6, 0.6], [-0.7, 0.2, 0.1], [0.5500055, 0, 0], [-0.5500055, 0, 0],
                           [0.0, 0.0, 0.0], [0.549, 0.549, -0.549], [10.0, -10.0, 3.0]], dtype=np.float32)
    o1["points"] = pts
    o1["logits"] = np32(mlp3(torch.from_numpy(pts), (xy, yz, xz)).logits)
    # the dense query grid of generate_mesh_fromdiffusion (generation.py:87-94) at nx = 6, through eval_points' chunking
    nx = 6
    lin = lambda n: torch.linspace(-0.5, 0.5, n)
    grid = 1.1 * torch.stack([lin(nx).view(-1, 1, 1).expand(nx, nx, nx).reshape(-1), lin(nx).view(1, -1, 1).expand(nx, nx, nx).reshape(-1),
                              lin(nx).view(1, 1, -1).expand(nx, nx, nx).reshape(-1)], dim=1)
    o1["grid6"] = np32(grid)
    one = tuple([pl[:1] for pl in lst] for lst in (xy, yz, xz))
    o1["grid6_logits"] = np32(mlp3(grid[None], one).logits[0])
    np.savez(os.path.join(OUT, "O1_occupancy.npz"), **o1)

    # ---------------- C1: coordinate grids -------------------------------------------------------
    c1 = {}
    for R in (4, 64, 256):
        e = (R - 1) / R
        c1[f"grid{R}"] = np32(convert_to_coord_format_2d(1, R, R, hstart=-e, hend=e, wstart=-e, wend=e))
    c1["grid5_default"] = np32(convert_to_coord_format_2d(1, 5, 5))
    np.savez(os.path.join(OUT, "C1_coords.npz"), **c1)

    # ---------------- manifests: full-size state-dict key/shape lists (AFHQ) ----------------------
    with torch.device("meta"):
        import yaml
        cfg = yaml.safe_load(open(os.path.join(REF, "configs/ldm/afhq.yaml")))["model"]
        un = UNetModel(**cfg["params"]["unetconfig"])
        manifests["afhq.unet"] = {k: list(v.shape) for k, v in un.state_dict().items()}
        aef = Autoencoder(ddconfig=cfg["params"]["ddconfig"], embed_dim=cfg["embed_dim"])
        manifests["afhq.vae_decoder"] = {k: list(v.shape) for k, v in aef.state_dict().items()
                                         if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    with open(os.path.join(OUT, "manifests.json"), "w") as f:
        json.dump(manifests, f)
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump({"TOY_UNET": TOY_UNET, "TOY_UNET_B": TOY_UNET_B, "TOY_DD": TOY_DD, "TOY_EMBED": TOY_EMBED,
                   "MLP_CFG": MLP_CFG, "TOY_DD3": TOY_DD3, "TOY_EMBED3": TOY_EMBED3, "MLP3D_CFG": MLP3D_CFG}, f)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, f"({tot / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
