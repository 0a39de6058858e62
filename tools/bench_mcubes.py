#!/usr/bin/env python3
"""Times the device marching cubes (extract_mesh: virtual border, count + emit) on synthetic smooth logit grids.

    python tools/bench_mcubes.py [--sizes 128,256] [--iters 20]

Algorithmic bytes per run: the grid is read once (4 B/voxel) and the mesh written once (24 B/vertex + 24 B/triangle);
the per-cell bookkeeping (1 + 8 B/cell written, read back) is the method's own traffic and is reported separately.
The host-side comparison (the CPU oracle and the reference's own marchingcubes.cpp) lives with the tests, which are the only
code allowed to touch `oracle/`: `python tests/time_mcubes_cpu.py`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def field(n, dev):
    import torch
    g = torch.linspace(-1, 1, n, device=dev)
    x, y, z = torch.meshgrid(g, g, g, indexing="ij")
    r = torch.sqrt((x - 0.05) ** 2 + (y + 0.02) ** 2 + (z - 0.03) ** 2)
    return (4.0 * (0.62 - r) + 0.3 * torch.sin(7 * x) * torch.cos(5 * y + z)).float()


def run(n, iters, dev="cuda:0"):
    import torch
    from ddmi_amd import mcubes
    occ = field(n, dev)
    v, f = mcubes.extract_mesh(occ, 0.2, 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        v, f = mcubes.extract_mesh(occ, 0.2, 0.1)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    cells = (n + 1) ** 3
    algo = 4 * n ** 3 + 24 * v.shape[0] + 24 * f.shape[0]
    book = cells * (1 + 1 + 8 + 1) + 8 * v.shape[0]
    line = {"grid": n, "vertices": v.shape[0], "triangles": f.shape[0], "gpu_ms": round(ms, 3),
            "algorithmic_MB": round(algo / 1e6, 2), "algorithmic_GBps": round(algo / ms / 1e6, 1),
            "bookkeeping_MB": round(book / 1e6, 2)}
    return line, occ, v, f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="128,256")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    for n in (int(s) for s in args.sizes.split(",")):
        print(json.dumps(run(n, args.iters)[0]), flush=True)


if __name__ == "__main__":
    main()
