#!/usr/bin/env python3
"""Diagnostics: the fused conv + GroupNorm op (ddmi_op_conv_gn) at growing batch - time per run and the hand-off's timeout flag."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddmi_amd import _lib  # noqa: E402

L = _lib.lib()
_lib.init_device(0)
Cin, H, Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 64, int(sys.argv[4]) if len(sys.argv) > 4 else 224
for B in [int(b) for b in sys.argv[1].split(",")]:
    x = torch.randn(B, Cin, H, H, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    emb = torch.randn(B, Cout, device="cuda")
    g = torch.ones(Cout, device="cuda")
    y = torch.empty(B, Cout, H, H, device="cuda")
    for fuse in [int(m) for m in os.environ.get("FGN_MODES", "0,2").split(",")]:
        used = C.c_int(-1)
        L.ddmi_debug_gn_fuse(fuse)
        torch.cuda.synchronize()
        t0 = time.time()
        rc = L.ddmi_op_conv_gn(x.data_ptr(), w.data_ptr(), b.data_ptr(), emb.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), B, Cin, H, H, Cout,
                               1e-5, 1, 1, int(os.environ.get('FGN_ITERS', '3')), C.byref(used))
        torch.cuda.synchronize()
        print(f"B={B} Cin={Cin} H={H} Cout={Cout} fuse={fuse}: rc={rc} fused_used={used.value} {time.time() - t0:.2f} s for plan + 3 runs "
              f"{L.ddmi_last_error().decode() if rc else ''}", flush=True)
    L.ddmi_debug_gn_fuse(-1)
