#!/usr/bin/env python3
"""Diagnostics: the full AFHQ UNet at batch B with the producer-side GroupNorm on / off: time per forward, fused launches, timeout flag."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from ddmi_amd import _lib, initweights as iw  # noqa: E402
from ddmi_amd.unet import UNetModel  # noqa: E402
from helpers import AFHQ_UNET  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib.init_device(0)
u = UNetModel(**AFHQ_UNET, numerics="bf16", max_batch=B)
u.load_state_dict(iw.fill_state("bench.unet", u.manifest()))
x = torch.randn(B, 64, 64, 64, device="cuda")
t = torch.full((B,), 100, dtype=torch.int64, device="cuda")
print("finalize ...", flush=True)
t0 = time.time()
u.finalize(0)
print(f"finalized in {time.time() - t0:.1f} s", flush=True)
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.time()
    y = u(x, t)
    torch.cuda.synchronize()
    dt = time.time() - t0
    try:
        fused = u.async_status()
        st = "ok"
    except Exception as e:  # noqa: BLE001
        fused, st = -1, str(e)[:120]
    print(f"B={B} forward {i}: {dt * 1e3:.1f} ms, fused GroupNorm launches {fused}, status {st}, finite {bool(torch.isfinite(y).all())}", flush=True)
