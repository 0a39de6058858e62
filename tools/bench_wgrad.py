#!/usr/bin/env python3
"""conv backward at the AFHQ UNet's training shapes (B x H x H x Ci -> Co, 3x3): per-kernel time and TFLOP/s of the wgrad
(TN implicit GEMM, or the im2col + GEMM formulation with DDMI_WGRAD_GEMM=1) and dgrad launches, from the library's own
HIP-event scopes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402
from ddmi_amd import train as T  # noqa: E402

SHAPES = [(64, 224, 224, 3), (64, 448, 224, 3), (64, 672, 224, 3), (32, 448, 448, 3), (32, 896, 448, 3), (32, 1344, 448, 3),
          (16, 896, 896, 3), (16, 1792, 896, 3), (16, 2240, 896, 3), (8, 1344, 1344, 3), (8, 2688, 1344, 3), (32, 224, 448, 1), (16, 448, 896, 1)]

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = "cuda:0"
    _lib.init_device(0)
    tc = T.TrainContext("bf16", dev)
    for hw, ci, co, k in SHAPES:
        x = torch.randn((B, hw, hw, ci), device=dev).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn((co, ci, k, k), device=dev) / (ci * k * k) ** 0.5).requires_grad_(True)
        b = torch.zeros((co,), device=dev, requires_grad=True)
        dy = torch.randn((B, hw, hw, co), device=dev).to(torch.bfloat16)
        z = torch.zeros((1, 64), device=dev)
        one = torch.ones((1,), device=dev)
        for it in range(5):
            if it == 2:
                torch.cuda.synchronize()
                _lib.profile_enable(1)
            # a training step (and with it the profiler's sampling window) starts at q_sample
            _lib.check(_lib.lib().ddmi_train_q_sample(z.data_ptr(), z.data_ptr(), one.data_ptr(), one.data_ptr(), z.clone().data_ptr(), 1, 64,
                                                      _lib.stream_ptr()))
            y = T.ConvFn.apply(tc, x, w, b, None, None, 1, False)
            y.backward(dy)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(0)
        fl = 2.0 * B * hw * hw * ci * co * k * k
        row = [f"{hw:3d}x{hw:<3d} {ci:5d}->{co:<5d} k{k}  ({fl / 1e9:7.1f} GFLOP)"]
        for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
            us = 1e3 * v["total_ms"] / max(1, v["launches"])
            tf = v.get("flops", 0.0) / max(1e-9, v["total_ms"]) / 1e9
            row.append(f"{name}: {us:7.1f} us x{v['launches'] // 3} {tf:6.0f} TF")
        print(" | ".join(row[:1] + row[1:6]), flush=True)
