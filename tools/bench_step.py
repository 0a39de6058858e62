#!/usr/bin/env python3
"""The fused sampler-step kernel at the bench shape (512 x 64 x 64 x 64 fp32): time per launch and achieved HBM rate for
its algorithmic traffic (12 B per element with in-kernel Philox noise: x, eps in; x out; 16 B with a noise tensor)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402
from ddmi_amd.ddpm import DDPM  # noqa: E402

if __name__ == "__main__":
    dev, B = "cuda:0", int(sys.argv[1]) if len(sys.argv) > 1 else 512
    _lib.init_device(0)
    d = DDPM(model=None, timesteps=250, sampling_timesteps=250, channels=64)
    x, mo, out = (torch.randn((B, 64, 64, 64), device=dev) for _ in range(3))
    nz = torch.randn_like(x)
    seeds = torch.arange(B, dtype=torch.int64, device=dev) + (777 << 32)
    n = x.numel()
    for label, kw, bpe in (("philox noise in registers", dict(seeds=seeds, noise=None), 12), ("noise tensor", dict(seeds=None, noise=nz), 16)):
        for _ in range(3):
            d._step(x, mo, kw["noise"], d.ancestral_coefs(100), _lib.STEP_ANCESTRAL, out, False, kw["seeds"], 7)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            d._step(x, mo, kw["noise"], d.ancestral_coefs(100), _lib.STEP_ANCESTRAL, out, False, kw["seeds"], i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"ddpm_step B={B} ({label}): {us:.1f} us per launch, {bpe * n / us / 1e6:.2f} TB/s of algorithmic traffic "
              f"({bpe * n / us / 1e6 / 8.0:.3f} of the 8 TB/s HBM peak)")
