#!/usr/bin/env python3
"""The tri-plane UNet's token attention (16 heads over the xy | xt | yt tokens of a sample) at its four levels: kernel time from the
library's HIP-event scope, scores per ns and TFLOP/s (4 N^2 d per head).  `bench_attn_mh.py B prescaled` runs the bf16 networks' arrangement
(q pre-scaled: the lazy loop, and the resident kernel at the d = 16 / 32 levels unless DDMI_ATTN_RES=0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402

LEVELS = [(16, 16, 2048, 1024, 512), (16, 32, 2048, 1024, 512), (16, 32, 512, 256, 128), (16, 64, 512, 256, 128), (16, 64, 128, 64, 32),
          (16, 96, 32, 16, 8)]   # heads, d, N, seg0, seg1 (configs/ldm/skytimelapse.yaml at 16 x 256^2: tools/prof_attn_shapes.py lists them)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    prescaled = len(sys.argv) > 2 and sys.argv[2] == "prescaled"
    dev = "cuda:0"
    _lib.init_device(0)
    L = _lib.lib()
    z = torch.zeros((1, 64), device=dev)
    one = torch.ones((1,), device=dev)
    levels = LEVELS
    if len(sys.argv) > 3 and sys.argv[3] == "decoder":   # VideoDecoder_light's inter-plane attention at decode batch 4 (B is the decode batch)
        levels = [(8, 64, 73728, 32768, 20480), (8, 128, 20480, 8192, 6144), (8, 256, 6144, 2048, 2048), (8, 512, 2048, 1024, 512)]
    for heads, d, N, s0, s1 in levels:
        C = heads * d
        q, k, v = (torch.randn((B, C, N), device=dev) for _ in range(3))
        out = torch.empty((B, C, N), device=dev)
        for it in range(4):
            if it == 1:
                torch.cuda.synchronize()
                _lib.profile_enable(1)
            _lib.check(L.ddmi_train_q_sample(z.data_ptr(), z.data_ptr(), one.data_ptr(), one.data_ptr(), z.clone().data_ptr(), 1, 64,
                                             _lib.stream_ptr()))   # opens the profiler's sampling window
            if prescaled:
                _lib.check(L.ddmi_op_attention_mh_prescaled(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, d, N, s0, s1))
            else:
                _lib.check(L.ddmi_op_attention_mh(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, d, N, s0, s1,
                                                  _lib.numerics_code("bf16")))
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(0)
        p = prof.get("attention_mh")
        us = 1e3 * p["total_ms"] / p["launches"]
        scores = B * heads * N * N
        clk = scores / us / 1e3 / 1024 / 2.2   # scores per clock and SIMD at 2.2 GHz
        print(f"B={B} heads={heads} d={d:3d} N={N:5d}: {us:9.1f} us  {scores / us / 1e3:7.2f} scores/ns  {4.0 * scores * d / us / 1e6:7.1f} TFLOP/s  {clk:5.2f} scores/clk/SIMD", flush=True)
