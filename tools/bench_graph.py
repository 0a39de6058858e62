#!/usr/bin/env python3
"""Small-batch latency of one UNet evaluation with and without hipGraph replay (ddmi_unet_set_graph), image (AFHQ) and
video (SkyTimelapse) UNets, bf16.  Also checks that both paths give identical outputs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ddmi_amd.unet import UNetModel  # noqa: E402
from ddmi_amd.video import UNetModel_Triplane  # noqa: E402
from helpers import AFHQ_UNET, SKY_UNET, named_state  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3


if __name__ == "__main__":
    dev = "cuda:0"
    side = torch.cuda.Stream()  # stream capture is not allowed on the legacy default stream
    for name, cls, cfg, shape in (("afhq", UNetModel, AFHQ_UNET, (64, 64, 64)), ("sky", UNetModel_Triplane, SKY_UNET, (64, 2048))):
        bmax = 8
        m = cls(numerics="bf16", max_batch=bmax, **cfg)
        m.load_state_dict(named_state(name + ".g", m.manifest()))
        m.set_uniform_t(True)
        for B in (1, 2, 4, 8):
            x = torch.randn((B,) + shape, device=dev)
            t = torch.full((B,), 321, dtype=torch.int64, device=dev)
            out = torch.empty_like(x)
            with torch.cuda.stream(side):
                m.set_graph(False)
                ref = m(x, t).clone()
                ms_plain = timed(lambda: m(x, t, out=out))
                m.set_graph(True)
                m(x, t, out=out)  # capture
                m(x, t, out=out)  # replay
                side.synchronize()
                same = bool(torch.equal(out, ref))
                ms_graph = timed(lambda: m(x, t, out=out))
            print(f"{name} B={B}: launches {ms_plain:.2f} ms, graph {ms_graph:.2f} ms ({ms_plain / ms_graph:.2f}x), identical={same}")
        del m
        torch.cuda.empty_cache()
