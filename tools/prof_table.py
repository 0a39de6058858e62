#!/usr/bin/env python3
"""Table of a profiler record written by `DDMI_PROF_DUMP=<dir> DDMI_PROF_SHAPES=1 python bench.py --workload ...`: share, launches, us per launch, TFLOP/s."""
import json
import sys

if __name__ == "__main__":
    prof = json.load(open(sys.argv[1]))
    cut = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
    tot = sum(v["total_ms"] for v in prof.values())
    print(f"{tot:.1f} ms of profiled kernel time")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        if v["total_ms"] / tot < cut:
            continue
        us = 1e3 * v["total_ms"] / v["launches"]
        tf = v["flops"] / v["total_ms"] / 1e9 if v.get("flops") else 0.0
        print(f"{100 * v['total_ms'] / tot:6.2f} %  {v['launches']:6d} x {us:10.1f} us  {tf:8.1f} TFLOP/s  {k}")
