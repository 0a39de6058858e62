#!/usr/bin/env python3
"""Kernel time shares of one AFHQ UNet evaluation in the bf16x3 parity mode (and, for scale, the bf16 mode) at a given batch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ddmi_amd import _lib  # noqa: E402
from ddmi_amd.unet import UNetModel  # noqa: E402
from helpers import AFHQ_UNET, named_state  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for numerics in (sys.argv[2:] or ["bf16x3", "bf16"]):
        m = UNetModel(**AFHQ_UNET, numerics=numerics, max_batch=B)
        m.load_state_dict(named_state("afhq.unet", m.manifest()))
        m.finalize(0) if hasattr(m, "finalize") else None
        m.set_uniform_t(True)
        x = torch.randn((B, 64, 64, 64), device="cuda:0")
        t = torch.full((B,), 100, dtype=torch.int64, device="cuda:0")
        for _ in range(2):
            m(x, t)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            m(x, t)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        _lib.profile_enable(1)
        m(x, t)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(0)
        tot = sum(v["total_ms"] for v in prof.values())
        print(f"{numerics} B={B}: {ms:.2f} ms per evaluation ({B / ms * 1e3 / 250:.2f} samples/s at 250 steps); profiled kernel time {tot:.2f} ms")
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
            print(f"   {k:44s} {v['total_ms']:8.2f} ms  {v['launches']:4d} launches")
        del m
        torch.cuda.empty_cache()
