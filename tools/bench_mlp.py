#!/usr/bin/env python3
"""Timing of the three fused plane-sampling + MLP kernels at their BASELINE sizes (image 256^2, occupancy 128^3, video
16x256^2) on random planes; every launch timed with the library's HIP-event profiler."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ddmi_amd import _lib  # noqa: E402
from ddmi_amd.coords import generation_coords  # noqa: E402
from ddmi_amd.mlp import MLP  # noqa: E402
from helpers import MLP_CFG, named_state  # noqa: E402


def timed(label, fn, iters, key):
    fn()
    torch.cuda.synchronize()
    _lib.profile_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    p = _lib.profile_collect()
    _lib.profile_enable(0)
    d = p[key]
    ms = d["total_ms"] / d["launches"]
    print(f"{label}: {ms:.3f} ms per launch, {d['flops'] / d['launches'] / ms / 1e9:.1f} TFLOP/s "
          f"({d['flops'] / d['launches'] / ms / 1e9 / 2500:.3f} of the bf16 MFMA peak)", flush=True)
    return ms


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="2d,3d,v")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = "cuda:0"
    which = args.which.split(",")
    if "2d" in which:
        B = args.batch
        mlp = MLP(**MLP_CFG, numerics="bf16")
        mlp.load_state_dict(named_state("afhq.mlp", mlp.manifest()))
        planes = [torch.randn((B, 64, p, p), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for p in (64, 128, 256)]
        coords = generation_coords(256)
        timed(f"mlp2d B={B} R=256", lambda: mlp(coords, hdbf=planes, si=1.0), args.iters, "mlp2d_fused")
    if "3d" in which:
        from ddmi_amd.occupancy import MLP3D, occupancy_grid_logits
        m3 = MLP3D(numerics="bf16")
        m3.load_state_dict(named_state("shapenet.mlp", m3.manifest()))
        pl = tuple([torch.randn((1, 64, p, p), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for p in (16, 32, 64)]
                   for _ in range(3))
        timed("mlp3d 128^3 grid", lambda: occupancy_grid_logits(m3, pl, 128), args.iters, "mlp3d_fused")
    if "v" in which:
        from ddmi_amd.video import MLPVideo, video_generation_coords
        mv = MLPVideo(numerics="bf16")
        mv.load_state_dict(named_state("sky.mlpv", mv.manifest()))
        cl = lambda t: t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        xy = [cl(torch.randn((1, 64, p, p), device=dev)) for p in (64, 128, 256)]
        yt = [cl(torch.randn((1, 64, 16, p), device=dev)) for p in (64, 128, 256)]
        xt = [cl(torch.randn((1, 64, 16, p), device=dev)) for p in (64, 128, 256)]
        coords = video_generation_coords(256, 16)
        timed("mlpv 16x256x256", lambda: mv(coords, (xy, yt, xt)), args.iters, "mlpv_fused")
