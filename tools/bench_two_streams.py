#!/usr/bin/env python3
"""AFHQ UNet evaluations: one stream at B = 512 against two streams at B = 256 each (two handles, one host thread alternating):
do the tails / small kernels of one half-batch fill under the other's?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib, initweights  # noqa: E402
from ddmi_amd.unet import UNetModel  # noqa: E402
import bench  # noqa: E402

def make(B):
    u = UNetModel(**bench.AFHQ["unetconfig"], numerics="bf16", max_batch=B)
    u.load_state_dict(initweights.fill_state("afhq.unet", u.manifest()))
    u.finalize(0)
    u.set_uniform_t(True)
    return u

if __name__ == "__main__":
    _lib.init_device(0)
    dev = "cuda:0"
    N = 10
    u512 = make(512)
    x = torch.randn((512, 64, 64, 64), device=dev)
    t = torch.full((512,), 3, device=dev, dtype=torch.int64)
    for _ in range(2):
        u512(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        u512(x, t)
    torch.cuda.synchronize()
    print(f"one stream,  B=512     : {(time.perf_counter() - t0) / N * 1e3:8.2f} ms per 512 samples", flush=True)
    del u512
    ua, ub = make(256), make(256)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    xa, xb, ta = x[:256].contiguous(), x[256:].contiguous(), t[:256].contiguous()
    for _ in range(2):
        with torch.cuda.stream(sa):
            ua(xa, ta)
        with torch.cuda.stream(sb):
            ub(xb, ta)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        with torch.cuda.stream(sa):
            ua(xa, ta)
        with torch.cuda.stream(sb):
            ub(xb, ta)
    torch.cuda.synchronize()
    print(f"two streams, B=256 each: {(time.perf_counter() - t0) / N * 1e3:8.2f} ms per 512 samples", flush=True)
    t0 = time.perf_counter()
    for _ in range(N):
        ua(xa, ta)
        ub(xb, ta)
    torch.cuda.synchronize()
    print(f"one stream,  2 x B=256 : {(time.perf_counter() - t0) / N * 1e3:8.2f} ms per 512 samples", flush=True)
