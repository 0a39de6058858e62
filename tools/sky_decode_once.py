#!/usr/bin/env python3
"""One SkyTimelapse clip through VITAutoencoder.decode (+ MLPVideo) twice - warm-up, then the measured pass - with the attention launches'
shapes printed (DDMI_PROF_SHAPES scopes): the program the per-shape rocprofv3 passes of tools/gpu_r6_sky_attn.sh run."""
import os
import sys

os.environ["DDMI_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ddmi_amd import _lib, initweights as iw  # noqa: E402
from ddmi_amd.config import load_ldm_args  # noqa: E402
from ddmi_amd.video import VITAutoencoder  # noqa: E402

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = load_ldm_args(os.path.join(root, "configs", "ldm", "skytimelapse.yaml"))
    frames = int(a.data_config.get("frames", 16))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    vae = VITAutoencoder(dict(a.ddconfig), a.embed_dim, frames, numerics="bf16", max_batch=B)
    vae.load_state_dict(iw.fill_state("sky.vit", vae.manifest()))
    z = torch.randn((B, a.embed_dim, vae.seq_len), device="cuda:0")
    vae.decode(z)
    torch.cuda.synchronize()
    _lib.profile_enable(1)
    vae.decode(z)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(0)
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        if k.startswith("attention"):
            print(f"SHAPE {k} launches {v['launches']} us_per_launch {1e3 * v['total_ms'] / v['launches']:.1f} algorithmic_bytes {v['bytes'] / v['launches']:.0f} "
                  f"gflop {v['flops'] / v['launches'] / 1e9:.2f}")
