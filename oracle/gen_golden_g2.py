#!/usr/bin/env python3
"""Capture G2 (BASELINE config 1) and G3 (same-seed draw order) from the REAL reference at /root/reference.

G2 - SURVEY.md 8(c)/8(d) "config 1": the reference's own classes on CPU fp32 with the AFHQ `unetconfig / ddconfig /
mlpconfig` verbatim (full-width nets), latent [1, 64, 16, 16], a 10-step DDIM chain (timesteps=1000,
sampling_timesteps=10, eta=0) and a 10-step ancestral chain (timesteps=10=sampling_timesteps), decode, MLP at R=64
(si = 4), clamp - i.e. `LDMTrainer.generate` (tools/ldm/image.py:225-249) with `torch.manual_seed(777)` right before it
and NOTHING patched: x_T, the per-step `randn` and the 12 NoiseInjection `normal_()` draws come from torch's CPU
generator in the reference's own order (ddpm.py:278,288,363; blocks.py:293-295).
G3 - the same on the toy nets with NON-ZERO `noise.weight`, so the 12 NoiseInjection draws (and therefore the whole
draw order x_T -> steps -> injections) show in the output.

Runs ONLY in the build container; writes arrays only (tests/golden/G2_config1.npz, G3_same_seed.npz).  Weights are
regenerated on both sides from ddmi_amd.initweights.  About 25 s on 8 cores.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.dont_write_bytecode = True

from ddmi_amd import initweights as iw  # noqa: E402
import gen_golden as GG  # noqa: E402  (install_stubs, load_named, toy configs)

SEED = 777


def run_generate(ddpm, ae, mlp, shape, R, convert, get_si):
    """tools/ldm/image.py:225-249 minus the accelerate / torchvision wrapping."""
    e = (R - 1) / R
    coords = convert(1, R, R, hstart=-e, hend=e, wstart=-e, wend=e)
    si = get_si(R)
    torch.manual_seed(SEED)
    z = ddpm.sample(shape=shape)
    planes = ae.decode(z)
    raw = mlp(coords, hdbf=planes, si=si)
    return z, planes, raw, raw.clamp(min=-1.0, max=1.0)


def main():
    GG.install_stubs()
    torch.set_grad_enabled(False)
    import yaml
    from diffusion.ddpm import DDPM
    from models.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from models.d2c_vae.autoencoder_unet import Autoencoder
    from models.d2c_vae.mlp import MLP
    from utils.general_utils import convert_to_coord_format_2d, get_scale_injection
    np32 = GG.np32

    # ---------------- G2: config 1 -------------------------------------------------------------------------------
    cfg = yaml.safe_load(open(os.path.join(GG.REF, "configs/ldm/afhq.yaml")))["model"]
    p = cfg["params"]
    unet = UNetModel(**p["unetconfig"])
    GG.load_named(unet, "afhq.unet")
    ae = Autoencoder(ddconfig=p["ddconfig"], embed_dim=cfg["embed_dim"])
    GG.load_named(ae, "afhq.vae")
    mlp = MLP(**p["mlpconfig"])
    GG.load_named(mlp, "afhq.mlp")
    logit = iw.fill_state("afhq.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"]
    g2 = {"seed": np.int64(SEED)}
    shape = [1, 64, 16, 16]
    for name, T, S, eta in (("ddim", 1000, 10, 0.0), ("anc", 10, 10, 1.0)):
        d = DDPM(model=unet, timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=64, image_size=16)
        d.mixing_logit.data.copy_(torch.from_numpy(logit))
        z, planes, raw, img = run_generate(d, ae, mlp, shape, 64, convert_to_coord_format_2d, get_scale_injection)
        g2[f"{name}.z"] = np32(z)
        for i, pl in enumerate(planes):            # 4 KB slices + a checksum of each plane (full tensors regenerable)
            a = np32(pl)
            g2[f"{name}.plane{i}.shape"] = np.array(a.shape)
            g2[f"{name}.plane{i}.slice"] = a[0, :4, :16, :16].copy()
            g2[f"{name}.plane{i}.sum"] = np.float64(a.astype(np.float64).sum())
            g2[f"{name}.plane{i}.abssum"] = np.float64(np.abs(a.astype(np.float64)).sum())
        g2[f"{name}.raw"] = np32(raw)
        g2[f"{name}.img"] = np32(img)
    # the first draws of the seeded CPU generator, so a reader can check the stream without the reference
    torch.manual_seed(SEED)
    g2["x_T"] = np32(torch.randn(shape))
    np.savez(os.path.join(GG.OUT, "G2_config1.npz"), **g2)

    # ---------------- G3: same-seed draw order on the toy nets, noise.weight != 0 ---------------------------------
    unet = UNetModel(**GG.TOY_UNET)
    GG.load_named(unet, "U2a.unet")
    ae = Autoencoder(ddconfig=GG.TOY_DD, embed_dim=GG.TOY_EMBED)
    GG.load_named(ae, "D1.vae")
    mlp = MLP(**GG.MLP_CFG)
    GG.load_named(mlp, "M1n.mlp", overrides={"noise.weight": lambda key, v: iw.normal("M1n.mlp", key, v.shape, 0.0, 0.1)})
    logit = iw.fill_state("G1.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"]
    g3 = {"seed": np.int64(SEED)}
    shape = [2, 64, 8, 8]
    for name, T, S, eta in (("ddim_eta0", 1000, 10, 0.0), ("ddim_eta1", 1000, 10, 1.0), ("anc", 10, 10, 1.0)):
        d = DDPM(model=unet, timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=64, image_size=8)
        d.mixing_logit.data.copy_(torch.from_numpy(logit))
        z, planes, raw, img = run_generate(d, ae, mlp, shape, 32, convert_to_coord_format_2d, get_scale_injection)
        g3[f"{name}.z"], g3[f"{name}.raw"], g3[f"{name}.img"] = np32(z), np32(raw), np32(img)
    np.savez(os.path.join(GG.OUT, "G3_same_seed.npz"), **g3)
    for f in ("G2_config1.npz", "G3_same_seed.npz"):
        print(f, os.path.getsize(os.path.join(GG.OUT, f)), "bytes")


if __name__ == "__main__":
    main()
