#!/usr/bin/env python3
"""Timing of the video path's three stages at SkyTimelapse sizes (config 5 of SURVEY.md 8d): one tri-plane UNet
evaluation, VITAutoencoder.decode, MLPVideo over 16 x 256 x 256.  Random-init named-seed weights, synthetic latents."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ddmi_amd import _lib, initweights as iw  # noqa: E402
from ddmi_amd.video import MLPVideo, UNetModel_Triplane, VITAutoencoder, video_generation_coords  # noqa: E402
from helpers import SKY_DD, SKY_EMBED, SKY_FRAMES, SKY_UNET, named_state  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 4, 8])
    ap.add_argument("--numerics", default="bf16")
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    bmax = max(args.batch)
    unet = UNetModel_Triplane(numerics=args.numerics, max_batch=bmax, **SKY_UNET)
    unet.load_state_dict(named_state("sky.unet3", unet.manifest()))
    unet.set_uniform_t(True)
    vit = VITAutoencoder(SKY_DD, SKY_EMBED, SKY_FRAMES, numerics=args.numerics, max_batch=1)
    vit.load_state_dict(named_state("sky.vit", vit.manifest()))
    mlp = MLPVideo(numerics=args.numerics)
    mlp.load_state_dict(named_state("sky.mlpv", mlp.manifest()))
    coords = video_generation_coords(256, 16)
    for B in args.batch:
        x = torch.randn((B, 64, 2048), device=dev)
        t = torch.full((B,), 437, dtype=torch.int64, device=dev)
        ms = timed(lambda: unet(x, t), 5)
        print(f"unet3 B={B}: {ms:.2f} ms/eval  ({230.3 * B / ms:.1f} TFLOP/s at 230.3 GFLOP per sample-step)")
    z = torch.randn((1, 64, 2048), device=dev)
    ms_d = timed(lambda: vit.decode(z), 3)
    hdbf = vit.decode(z)
    ms_m = timed(lambda: mlp(coords, hdbf), 3)
    print(f"decode B=1: {ms_d:.1f} ms ({13.75 / ms_d * 1e3:.0f} TFLOP/s);  mlpvideo B=1: {ms_m:.2f} ms ({1.80 / ms_m * 1e3:.0f} TFLOP/s)")
    if args.profile:
        for label, fn in (("unet3 B=%d" % B, lambda: unet(x, t)), ("decode", lambda: vit.decode(z)), ("mlpvideo", lambda: mlp(coords, hdbf))):
            _lib.profile_enable(1)
            fn()
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(0)
            tot = sum(v["total_ms"] for v in prof.values())
            nl = sum(v["launches"] for v in prof.values())
            print(f"-- {label}: {tot:.2f} ms of kernel time in {nl} launches")
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:10]:
                tf = v["flops"] / (v["total_ms"] * 1e-3) / 1e12 if v.get("flops") else 0.0
                print(f"  {k:34s} {v['total_ms']:9.3f} ms  {v['launches']:5d} launches  {100 * v['total_ms'] / tot:5.1f} %  {tf:7.1f} TFLOP/s")
