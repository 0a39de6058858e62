"""Shared helpers for the parity tests (not collected by pytest)."""
import numpy as np
import torch

from ddmi_amd import initweights as iw

AFHQ_UNET = dict(image_size=64, in_channels=64, model_channels=224, out_channels=64, attention_resolutions=[8, 4, 2],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 6], num_head_channels=32)
AFHQ_DD = dict(double_z=True, z_channels=128, resolution=256, in_channels=3, out_ch=64, ch=128, ch_mult=[1, 2, 4],
               num_res_blocks=2, attn_resolutions=[], hdbf_resolutions=[128, 64], dropout=0.0, attn_type="vanilla")
AFHQ_EMBED = 64
MLP_CFG = dict(in_ch=2, out_ch=3, ch=256, latent_dim=64)

# tolerances (max-abs error / max-abs reference), stated once:
TOL_FP32 = 1e-3   # north star: "within 1e-3 relative fp32 tolerance" - parity numerics mode
TOL_BF16 = 4e-2   # perf mode: bf16 operands (8 mantissa bits) through tens of layers; checked per network


def named_state(tag, manifest):
    return iw.fill_state(tag, [(k, tuple(s)) for k, s in manifest])


def to_torch(state):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}


def rel_err(a, b):
    a = a.detach().float().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b = b.detach().float().cpu().numpy() if hasattr(b, "detach") else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / (np.abs(b).max() + 1e-30))

# video (config 5, configs/ldm/skytimelapse.yaml + configs/d2c-vae/skytimelapse.yaml)
SKY_UNET = dict(in_channels=64, model_channels=256, out_channels=64, attention_resolutions=[8, 4, 2], num_res_blocks=2,
                channel_mult=[1, 2, 4, 6], num_head_channels=64, size1=32, size2=32, size3=16)
SKY_DD = dict(double_z=True, timesformer_channels=384, splits=1, patch_size=8, resolution=256, z_channels=64, in_channels=3,
              out_ch=64, ch=64, ch_mult=[1, 2, 4, 8], num_res_blocks=2, attn_resolutions=[], hdbf_resolutions=[128, 64],
              inter_attn_resolutions=[32, 64, 128, 256], dropout=0.0, attn_type="vanilla-multihead")
SKY_EMBED, SKY_FRAMES = 64, 16
