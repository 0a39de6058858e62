#!/usr/bin/env python3
"""Times the device marching cubes (extract_mesh: virtual border, count + emit) on synthetic smooth logit grids and the
CPU oracle / reference build beside it.

    python tools/bench_mcubes.py [--sizes 128,256] [--iters 20]

Algorithmic bytes per run: the grid is read once (4 B/voxel) and the mesh written once (24 B/vertex + 24 B/triangle);
the per-cell bookkeeping (1 + 8 B/cell written, read back) is the method's own traffic and is reported separately.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="128,256")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    from ddmi_amd import mcubes
    from oracle import mcubes_oracle as M
    dev = "cuda:0"
    for n in (int(s) for s in args.sizes.split(",")):
        occ = field(n, dev)
        v, f = mcubes.extract_mesh(occ, 0.2, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            v, f = mcubes.extract_mesh(occ, 0.2, 0.1)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.iters
        cells = (n + 1) ** 3
        algo = 4 * n ** 3 + 24 * v.shape[0] + 24 * f.shape[0]
        book = cells * (1 + 1 + 8 + 1) + 8 * v.shape[0]
        line = {"grid": n, "vertices": v.shape[0], "triangles": f.shape[0], "gpu_ms": round(ms, 3),
                "algorithmic_MB": round(algo / 1e6, 2), "algorithmic_GBps": round(algo / ms / 1e6, 1),
                "bookkeeping_MB": round(book / 1e6, 2)}
        if not args.no_cpu:
            host = occ.cpu().numpy()
            t0 = time.perf_counter()
            rv, rf = M.extract_mesh(host, 0.2, 0.1)
            line["cpu_oracle_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
            if M.have_ref():
                t0 = time.perf_counter()
                M.extract_mesh(host, 0.2, 0.1, mc=M.marching_cubes_ref)
                line["cpu_reference_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
            t0 = time.perf_counter()
            occ.cpu()
            line["d2h_grid_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
            import numpy as np
            line["identical_to_oracle"] = bool(np.array_equal(v.cpu().numpy(), rv) and np.array_equal(f.cpu().numpy(), rf.astype("int64")))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
