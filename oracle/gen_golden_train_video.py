#!/usr/bin/env python3
"""Capture T2 - one VIDEO-domain training step of the REAL reference (mlvlab/DDMI at /root/reference) on a toy tri-plane UNet:
`DDPM(domain='video').p_losses` (diffusion/ddpm.py:409-447; mixing_logit [1, C, 1], :152-159) over `UNetModel_Triplane`
(openaimodel.py:800-1221) on a latent z [B, C, s1*s2 + s1*s3 + s2*s3] -> backward -> torch.optim.AdamW(lr=1e-4, weight_decay=0) step,
as tools/ldm/video.py:150-167 runs it, with x_start / t / noise given explicitly.

Stored (arrays only; weights are regenerated from ddmi_amd.initweights on both sides): loss, the full gradient of a handful of tensors
that exercise every backward op of the tri-plane network (the shared-weight convolutions applied to three planes, a ResBlock's
GroupNorm / emb projection, the per-plane AttentionBlock, the 16-head inter-plane attention's norm / q / k / v / proj_out after an input
block, the middle block and an output block, the Downsample / Upsample convolutions, time_embed, the output conv, mixing_logit), the
(sum, l2) of EVERY gradient, and the same tensors after the AdamW step.  `xformers` is stood in for as in gen_golden_video.py.
Runs only in the build container; ~10 s."""
import json
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
import gen_golden_video as GV  # noqa: E402

# 64 / 128 channels: the inter-plane attention's 16 heads get 4 / 8 channels, the per-plane AttentionBlock (num_head_channels 16) 8 heads
TOY_UNET_T2 = dict(in_channels=8, model_channels=64, out_channels=8, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2],
                   num_head_channels=16, size1=8, size2=8, size3=4)

FULL = ["model.input_blocks.0.0.weight", "model.input_blocks.1.0.in_layers.0.weight", "model.input_blocks.1.0.in_layers.2.weight",
        "model.input_blocks.1.0.emb_layers.1.weight", "model.input_blocks.1.0.out_layers.3.bias", "model.input_blocks.2.0.op.bias",
        "model.input_blocks.3.0.skip_connection.weight", "model.input_blocks.3.1.norm.weight", "model.input_blocks.3.1.qkv.weight",
        "model.input_blocks.3.1.proj_out.bias", "model.input_attns.1.norm.weight", "model.input_attns.1.q.weight", "model.input_attns.1.k.bias",
        "model.input_attns.1.v.weight", "model.input_attns.1.proj_out.weight", "model.input_attns.3.q.weight", "model.mid_attn.norm.bias",
        "model.mid_attn.k.weight", "model.mid_attn.proj_out.weight", "model.middle_block.1.qkv.bias", "model.output_blocks.1.2.conv.bias",
        "model.output_blocks.3.0.skip_connection.weight", "model.output_attns.0.v.weight", "model.output_attns.3.q.bias",
        "model.output_attns.3.proj_out.weight", "model.time_embed.0.weight", "model.time_embed.2.bias", "model.out.0.weight",
        "model.out.2.weight", "model.out.2.bias", "mixing_logit"]


def main():
    GG.install_stubs()
    GV.install_xformers()
    from diffusion.ddpm import DDPM
    from models.ldm.modules.diffusionmodules.openaimodel import UNetModel_Triplane
    torch.manual_seed(0)
    unet = UNetModel_Triplane(**TOY_UNET_T2)
    manifest = GG.load_named(unet, "T2.unet3")
    ddpm = DDPM(model=unet, timesteps=1000, sampling_timesteps=1000, channels=8, image_size=8, domain="video")
    assert tuple(ddpm.mixing_logit.shape) == (1, 8, 1)
    ddpm.mixing_logit.data.copy_(torch.from_numpy(iw.normal("T2", "mixing_logit", (1, 8, 1), 0.0, 0.5)))
    ddpm.train()
    L = 8 * 8 + 2 * 8 * 4
    shape = (2, 8, L)
    x0 = torch.from_numpy(iw.normal("T2", "x_start", shape))
    noise = torch.from_numpy(iw.normal("T2", "noise", shape))
    t = torch.tensor([5, 870], dtype=torch.long)
    opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4, weight_decay=0.0)   # tools/ldm/video.py:62
    loss = ddpm.p_losses(x0, None, t, noise=noise)
    if isinstance(loss, tuple):
        loss = loss[0]
    loss.backward()
    out = {"t": t.numpy(), "loss": np.float64(loss.item())}
    names = dict(ddpm.named_parameters())
    assert all(k in names for k in FULL), [k for k in FULL if k not in names]
    keys, sums, l2s = [], [], []
    for k, p in names.items():
        assert p.grad is not None, k
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
    np.savez(os.path.join(GG.OUT, "T2_train_step_video.npz"), **out)
    with open(os.path.join(GG.OUT, "T2_config.json"), "w") as f:
        json.dump({"TOY_UNET_T2": TOY_UNET_T2, "manifest": {k: list(v) for k, v in manifest.items()} if isinstance(manifest, dict) else manifest}, f)
    print("T2_train_step_video.npz", os.path.getsize(os.path.join(GG.OUT, "T2_train_step_video.npz")), "bytes; loss", loss.item(), "params", len(keys))


if __name__ == "__main__":
    main()
