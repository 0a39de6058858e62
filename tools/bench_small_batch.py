#!/usr/bin/env python3
"""The small-batch regime of the AFHQ UNet (SURVEY 8(d) config 2 at B in {1, 8, 16}; the reference's own gen mode runs
test_batch_size 6-10): ms per UNet evaluation as plain launches and as a hipGraph replay, launches per evaluation, and the
HBM floor (the bf16 weights read once at 6 TB/s) beside it.

    python tools/bench_small_batch.py [--batches 1,8,16] [--json out.json]
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/bench_small_batch.py --batches 1 --plain-only --iters 10
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddmi_amd import _lib  # noqa: E402
from ddmi_amd.unet import UNetModel  # noqa: E402
from helpers import AFHQ_UNET, named_state  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8,16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--plain-only", action="store_true")
    ap.add_argument("--graph-only", action="store_true")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    dev = "cuda:0"
    batches = [int(b) for b in args.batches.split(",")]
    _lib.init_device(0)
    m = UNetModel(numerics="bf16", max_batch=max(batches), **AFHQ_UNET)
    m.load_state_dict(named_state("afhq.g", m.manifest()))
    m.finalize()
    m.set_uniform_t(True)
    mem = m.memory()
    floor_ms = mem["weights"] / 6.0e12 * 1e3
    side = torch.cuda.Stream()   # stream capture is refused on the legacy default stream
    rows = []
    for B in batches:
        x = torch.randn((B, 64, 64, 64), device=dev)
        t = torch.full((B,), 123, dtype=torch.int64, device=dev)
        out = torch.empty_like(x)
        row = {"B": B}
        with torch.cuda.stream(side):
            if not args.graph_only:
                m.set_graph(False)
                row["ms_plain"] = timed(lambda: m(x, t, out=out), args.iters)
            if not args.plain_only:
                m.set_graph(True)
                m(x, t, out=out)
                m(x, t, out=out)
                row["ms_graph"] = timed(lambda: m(x, t, out=out), args.iters)
                m.set_graph(False)
        best = min(v for k, v in row.items() if k.startswith("ms_"))
        row["samples_per_sec_250nfe"] = B / (250 * best * 1e-3)
        rows.append(row)
        print(json.dumps(row), flush=True)
    doc = {"weights_bytes": mem["weights"], "hbm_floor_ms": floor_ms, "rows": rows}
    print(json.dumps(doc))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(doc, f)


if __name__ == "__main__":
    main()
