#!/usr/bin/env python3
"""Host-side timing beside tools/bench_mcubes.py (test infrastructure: this is where `oracle/` may be used): the CPU oracle
(oracle/mcubes_oracle.c), the reference's own marchingcubes.cpp (oracle/_ref, when built) and the D2H copy of the logit
grid that the reference's path pays, on the same smooth fields; also checks that the device mesh is identical.

    python tests/time_mcubes_cpu.py [--sizes 128,256]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="128,256")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import bench_mcubes
    from oracle import mcubes_oracle as M
    for n in (int(s) for s in args.sizes.split(",")):
        line, occ, v, f = bench_mcubes.run(n, args.iters)
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
        line["identical_to_oracle"] = bool(np.array_equal(v.cpu().numpy(), rv) and np.array_equal(f.cpu().numpy(), rf.astype("int64")))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
