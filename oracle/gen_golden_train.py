#!/usr/bin/env python3
"""Capture T1 - one training step of the REAL reference (mlvlab/DDMI at /root/reference) on the toy UNet:
`DDPM.p_losses` (diffusion/ddpm.py:409-447) -> backward -> torch.optim.AdamW(lr=1e-4, weight_decay=0) step
(tools/ldm/image.py:62, 158-172), with x_start / t / noise given explicitly.

Stored (arrays only; weights are regenerated from ddmi_amd.initweights on both sides): loss, the full gradient of a
handful of tensors that exercise every backward op (first conv, a ResBlock's convs / GroupNorm / emb projection, the
Downsample and Upsample convs, an AttentionBlock's norm / qkv / proj, time_embed, the output conv, mixing_logit), the
(sum, l2) of EVERY gradient, and the same tensors after the AdamW step.  Runs only in the build container; ~10 s."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.dont_write_bytecode = True

from ddmi_amd import initweights as iw  # noqa: E402
import gen_golden as GG  # noqa: E402

FULL = ["model.input_blocks.0.0.weight", "model.input_blocks.0.0.bias", "model.input_blocks.1.0.in_layers.0.weight",
        "model.input_blocks.1.0.in_layers.2.weight", "model.input_blocks.1.0.emb_layers.1.weight", "model.input_blocks.1.0.out_layers.0.bias",
        "model.input_blocks.1.0.out_layers.3.weight", "model.input_blocks.2.0.op.weight", "model.input_blocks.3.0.skip_connection.weight",
        "model.input_blocks.3.1.norm.weight", "model.input_blocks.3.1.qkv.weight", "model.input_blocks.3.1.qkv.bias",
        "model.input_blocks.3.1.proj_out.weight", "model.middle_block.1.qkv.weight", "model.output_blocks.1.2.conv.weight",
        "model.output_blocks.3.0.skip_connection.weight", "model.time_embed.0.weight", "model.time_embed.2.bias", "model.out.0.weight",
        "model.out.2.weight", "model.out.2.bias", "mixing_logit"]


def main():
    GG.install_stubs()
    from diffusion.ddpm import DDPM
    from models.ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.manual_seed(0)
    unet = UNetModel(**GG.TOY_UNET)
    GG.load_named(unet, "T1.unet")
    ddpm = DDPM(model=unet, timesteps=1000, sampling_timesteps=1000, channels=64, image_size=8)
    ddpm.mixing_logit.data.copy_(torch.from_numpy(iw.normal("T1", "mixing_logit", (1, 64, 1, 1), 0.0, 0.5)))
    ddpm.train()
    shape = (2, 64, 8, 8)
    x0 = torch.from_numpy(iw.normal("T1", "x_start", shape))
    noise = torch.from_numpy(iw.normal("T1", "noise", shape))
    t = torch.tensor([3, 900], dtype=torch.long)
    opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4, weight_decay=0.0)   # tools/ldm/image.py:62
    loss = ddpm.p_losses(x0, None, t, noise=noise)
    loss.backward()
    out = {"t": t.numpy(), "loss": np.float64(loss.item())}
    names = dict(ddpm.named_parameters())
    assert all(k in names for k in FULL), [k for k in FULL if k not in names]
    keys, sums, l2s = [], [], []
    for k, p in names.items():
        g = p.grad.detach().double()
        keys.append(k)
        sums.append(float(g.sum()))
        l2s.append(float(g.pow(2).sum().sqrt()))
        if k in FULL:
            out["grad." + k] = p.grad.detach().numpy().astype(np.float32)
    out["all_keys"] = np.array(keys)
    out["all_sum"] = np.array(sums)
    out["all_l2"] = np.array(l2s)
    opt.step()
    for k in FULL:
        out["adamw." + k] = names[k].detach().numpy().astype(np.float32)
    np.savez(os.path.join(GG.OUT, "T1_train_step.npz"), **out)
    print("T1_train_step.npz", os.path.getsize(os.path.join(GG.OUT, "T1_train_step.npz")), "bytes; loss", loss.item(), "params", len(keys))


if __name__ == "__main__":
    main()
