#!/usr/bin/env python3
"""The AFHQ UNet's QKVAttentionLegacy at its three levels (14 / 28 / 42 heads of 32 channels over 1024 / 256 / 64 tokens) at the bench
batch, once per workgroup order (DDMI_ATTN_XCDMAP: 0 = XCD owns whole samples, 2 = plain order with the query tile fastest).  Run under
`rocprofv3 --kernel-trace --stats` for the kernel times and under `rocprofv3 --pmc FETCH_SIZE` for the L2 -> fabric reads per dispatch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    _lib.init_device(0)
    L = _lib.lib()
    for heads, T in ((14, 1024), (28, 256), (42, 64)):
        qkv = torch.randn((B, heads * 96, T), device="cuda:0")
        out = torch.empty((B, heads * 32, T), device="cuda:0")
        for mode in ("2", "0"):
            os.environ["DDMI_ATTN_XCDMAP"] = mode
            for _ in range(3):
                _lib.check(L.ddmi_op_attention_legacy_hd(qkv.data_ptr(), out.data_ptr(), B, heads, 32, T, _lib.numerics_code("bf16")))
            torch.cuda.synchronize()
            print(f"heads={heads} T={T} map={mode} done (timings: the profiler's kernel trace)", flush=True)
        del qkv, out
