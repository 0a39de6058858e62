#!/usr/bin/env python3
"""Which host ops of one eager training step issue device-to-device copies (torch.profiler, B = 8)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib, train as T  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    _lib.init_device(0)
    from ddmi_amd import initweights as iw
    step = T.LDMTrainStep(bench.AFHQ["unetconfig"], dict(timesteps=1000, channels=64), numerics="bf16", lr=1e-4, device="cuda:0", use_graph=False)
    st = {"model." + k: v for k, v in iw.fill_state("afhq.unet", [(k, tuple(p.shape)) for k, p in step.model.state_dict().items()]).items()}
    step.load_state_dict(st, strict=False).finalize()
    z = torch.randn((8, 64, 64, 64), device="cuda:0")
    for _ in range(2):
        step.train_step(z)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        step.train_step(z)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    for e in sorted(ev, key=lambda e: -e.count)[:60]:
        if "emcpy" in e.key or "copy" in e.key.lower() or "clone" in e.key.lower() or "contiguous" in e.key.lower():
            print(f"{e.count:6d}  {e.key[:100]}")
