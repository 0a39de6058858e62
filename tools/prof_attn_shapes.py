#!/usr/bin/env python3
"""Attention launches of config 5 (sky) by SHAPE: one sampling step + decode + MLPVideo at the bench leg's batch with every launch timed
(DDMI_PROF_SHAPES=1 tags the profiler scope of `launch_attention` with B / heads / d / N), sorted by total time.
`prof_attn_shapes.py B nfe unet` times the sampling loop alone."""
import os
import sys

os.environ["DDMI_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ddmi_amd import _lib  # noqa: E402
from ddmi_amd.config import load_ldm_args  # noqa: E402
from ddmi_amd.pipeline import VideoPipeline  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = load_ldm_args(os.path.join(root, "configs", "ldm", "skytimelapse.yaml"))
    ddpm = dict(a.ddpmconfig, timesteps=nfe, sampling_timesteps=nfe, ddim_sampling_eta=1.0)
    pipe = VideoPipeline(a.unetconfig, a.ddconfig, a.mlpconfig, ddpm, a.embed_dim, frames=int(a.data_config.get("frames", 16)),
                         test_resolution=256, numerics="bf16", max_batch=B, device="cuda:0")
    pipe.vae.max_batch = 4
    pipe.load_named_seed("sky")
    pipe.finalize()
    pipe.generate(list(range(B)), base_seed=0)
    torch.cuda.synchronize()
    _lib.profile_enable(1)
    if len(sys.argv) > 3 and sys.argv[3] == "unet":   # the sampling loop only (no decode / MLPVideo)
        from ddmi_amd.pipeline import sample_seed
        seeds = torch.tensor([sample_seed(1, i) for i in range(B)], dtype=torch.int64, device="cuda:0")
        pipe.ddpm.sample([B, a.embed_dim, pipe.unet.seq_len], seeds=seeds, device="cuda:0")
    else:
        pipe.generate(list(range(B)), base_seed=1)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(0)
    tot = sum(v["total_ms"] for v in prof.values())
    print(f"B={B} nfe={nfe}: {tot:.1f} ms of kernel time ({nfe} UNet evaluations + {B} decodes)")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        if v["total_ms"] / tot < (0.004 if len(sys.argv) > 3 else 0.002) and not k.startswith("attention"):
            continue
        us = 1e3 * v["total_ms"] / v["launches"]
        tf = v["flops"] / v["total_ms"] / 1e9 if v.get("flops") else 0.0
        print(f"{100 * v['total_ms'] / tot:6.2f} %  {v['launches']:6d} x {us:10.1f} us  {tf:8.1f} TFLOP/s  {k}")
