#!/usr/bin/env python3
"""Golden vectors for the VIDEO path, captured from the real reference.

Separate from gen_golden.py because it needs one more import stand-in: `xformers` (absent here, unpinned in the
reference's README.md:33).  `make_attn('vanilla-1d-multihead-expand')` raises without it (autoencoder_unet.py:596-600),
so the video classes cannot be built at all otherwise.  The stand-in restates the one function the reference calls,
`xformers.ops.memory_efficient_attention(q, k, v, attn_bias=None)` on [batch*heads, tokens, dim] tensors, by its
published definition softmax(q k^T / sqrt(dim)) v - with plain torch ops, independent of oracle/ddmi_oracle.py.
Everything else (module graphs, reshapes, head splits, plane bookkeeping, DDPM video branch) is the reference's code.

Writes tests/golden/V*.npz + manifests_video.json + configs_video.json.   Usage: python oracle/gen_golden_video.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import gen_golden as G  # noqa: E402  (stubs for torchvision / models.d2c_vae.op, DrawQueue, load_named)
from ddmi_amd import initweights as iw  # noqa: E402

OUT = G.OUT

TOY_UNET_V = dict(in_channels=8, model_channels=32, out_channels=8, num_res_blocks=1, attention_resolutions=[2],
                  channel_mult=[1, 2], num_head_channels=16, size1=8, size2=8, size3=4)
TOY_DDV = dict(double_z=True, timesformer_channels=32, splits=1, patch_size=8, resolution=64, z_channels=8, in_channels=3,
               out_ch=64, ch=32, ch_mult=[1, 1, 2, 2], num_res_blocks=1, attn_resolutions=[], hdbf_resolutions=[32, 16],
               inter_attn_resolutions=[8, 16, 32, 64], dropout=0.0, attn_type="vanilla-multihead")
TOY_EMBED_V, TOY_FRAMES = 8, 4
MLPV_CFG = dict(in_ch=2, out_ch=3, ch=256, latent_dim=64)


def install_xformers():
    xf, xfo = types.ModuleType("xformers"), types.ModuleType("xformers.ops")

    def memory_efficient_attention(q, k, v, attn_bias=None, op=None):
        assert attn_bias is None and q.dim() == 3
        s = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (q.shape[-1] ** -0.5), dim=-1)
        return torch.bmm(s, v)

    xfo.memory_efficient_attention = memory_efficient_attention
    xf.ops = xfo
    sys.modules.update({"xformers": xf, "xformers.ops": xfo})


def main():
    os.makedirs(OUT, exist_ok=True)
    G.install_stubs()
    install_xformers()
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    from diffusion.ddpm import DDPM
    from models.ldm.modules.diffusionmodules.openaimodel import UNetModel_Triplane
    from models.d2c_vae.autoencoder_vit import VITAutoencoder
    from models.d2c_vae.autoencoder_unet import MemoryEfficientAttnBlock1D, MemoryEfficientAttnBlock1D_expand
    from models.d2c_vae.mlp import MLPVideo
    from utils.general_utils import convert_to_coord_format_3d

    manifests = {}
    np32 = G.np32

    # ---------------- V0: the two 1-D attention blocks on their own -------------------------------------
    v0 = {}
    a16 = MemoryEfficientAttnBlock1D(64, num_heads=16)
    manifests["V0.attn16"] = G.load_named(a16, "V0.attn16")
    a8e = MemoryEfficientAttnBlock1D_expand(32, num_heads=8)
    manifests["V0.attn8e"] = G.load_named(a8e, "V0.attn8e")
    x16 = iw.normal("V0", "x16", (2, 64, 40))
    x8e = iw.normal("V0", "x8e", (2, 32, 72))
    v0.update(x16=x16, y16=np32(a16(torch.from_numpy(x16))), x8e=x8e, y8e=np32(a8e(torch.from_numpy(x8e))))
    np.savez(os.path.join(OUT, "V0_attn1d.npz"), **v0)

    # ---------------- V1: tri-plane UNet ---------------------------------------------------------------
    unet = UNetModel_Triplane(**TOY_UNET_V)
    manifests["V1.unet3"] = G.load_named(unet, "V1.unet3")
    L = TOY_UNET_V["size1"] * TOY_UNET_V["size2"] + 2 * TOY_UNET_V["size1"] * TOY_UNET_V["size3"]
    x = iw.normal("V1", "x", (2, 8, L))
    v1 = {"x": x}
    for name, ts in (("t7", [7, 7]), ("t3_11", [3, 11])):
        v1[name] = np32(unet(torch.from_numpy(x), torch.tensor(ts, dtype=torch.long)))
    np.savez(os.path.join(OUT, "V1_unet3.npz"), **v1)

    # ---------------- V2: VITAutoencoder.decode ---------------------------------------------------------
    vit = VITAutoencoder(ddconfig=TOY_DDV, embed_dim=TOY_EMBED_V, frames=TOY_FRAMES)
    full = G.load_named(vit, "V2.vit")
    manifests["V2.vit"] = {k: v for k, v in full.items() if k.startswith("decoder.") or k.startswith("post_")}
    z = iw.normal("V2", "z", (1, TOY_EMBED_V, L))
    xy, yt, xt = vit.decode(torch.from_numpy(z))
    v2 = {"z": z}
    for name, lst in (("xy", xy), ("yt", yt), ("xt", xt)):
        for i, pl in enumerate(lst):
            v2[f"{name}{i}"] = np32(pl)
    np.savez(os.path.join(OUT, "V2_vit_decode.npz"), **v2)

    # ---------------- V3: MLPVideo on those planes ------------------------------------------------------
    mlp = MLPVideo(**MLPV_CFG)
    manifests["V3.mlpv"] = G.load_named(mlp, "V3.mlpv")
    R, Tn = 64, TOY_FRAMES
    e = (R - 1) / R
    coords = convert_to_coord_format_3d(1, R, R, Tn, hstart=-e, hend=e, wstart=-e, wend=255 / R, tstart=-3 / 4, tend=3 / 4)
    vid = mlp(coords, (xy, yt, xt))
    np.savez(os.path.join(OUT, "V3_mlpvideo.npz"), video=np32(vid),
             **{f"coords_{k}": np32(v) for k, v in coords.items()})

    # ---------------- V4: whole video chain (tools/ldm/video.py:214-230) --------------------------------
    v4 = {}
    shape = [1, 8, L]
    xT = iw.normal("V4", "x_T", shape)
    v4["x_T"] = xT
    for name, T, S, eta in (("ddim_eta0", 20, 4, 0.0), ("anc", 5, 5, 1.0)):
        d = DDPM(model=unet, timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=8, domain="video")
        assert tuple(d.mixing_logit.shape) == (1, 8, 1)
        manifests.setdefault("V4.ddpm_extra", {"mixing_logit": [1, 8, 1]})
        d.mixing_logit.data.copy_(torch.from_numpy(iw.fill_state("V4.ddpm", [("mixing_logit", (1, 8, 1))])["mixing_logit"]))
        ndraw = S - 1 if S < T else T
        draws = [iw.normal("V4", f"{name}.n{i}", shape) for i in range(ndraw)]
        with G.DrawQueue(draws).patched():
            zz = d.sample(shape=shape, noise=torch.from_numpy(xT))
        v4[f"{name}.z"] = np32(zz)
        out = mlp(coords, vit.decode(zz)).clamp(-1.0, 1.0)
        v4[f"{name}.video"] = np32(out)
    np.savez(os.path.join(OUT, "V4_generate_video.npz"), **v4)

    with open(os.path.join(OUT, "manifests_video.json"), "w") as f:
        json.dump(manifests, f)
    with open(os.path.join(OUT, "configs_video.json"), "w") as f:
        json.dump(dict(TOY_UNET_V=TOY_UNET_V, TOY_DDV=TOY_DDV, TOY_EMBED_V=TOY_EMBED_V, TOY_FRAMES=TOY_FRAMES,
                       MLPV_CFG=MLPV_CFG), f)
    sizes = {fn: os.path.getsize(os.path.join(OUT, fn)) for fn in sorted(os.listdir(OUT)) if fn.startswith("V")}
    print(sizes)


if __name__ == "__main__":
    main()
