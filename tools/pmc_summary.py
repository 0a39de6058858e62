#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) into per-kernel HBM bytes per launch.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- python3 bench.py --nfe 4 --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -- python3 bench.py ... (same command)
  python tools/pmc_summary.py out_f out_w "<command text>" > profiles/rN_pmc_traffic.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes, so it is doubled; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(dirname, counter):
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    for fn in files:
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                a = acc[row["Kernel_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


if __name__ == "__main__":
    fdir, wdir = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    out = {"command": cmd, "head": os.environ.get("DDMI_HEAD") or (sys.argv[4] if len(sys.argv) > 4 else None),
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE x1, KiB -> bytes",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
        nf, fk = fetch.get(k, [0, 0.0])
        nw, wk = write.get(k, [0, 0.0])
        n = max(nf, nw)
        out["kernels"][k] = {"dispatches": n, "fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1),
                             "hbm_bytes_per_launch": int((2.0 * fk / max(nf, 1) + wk / max(nw, 1)) * 1024)}
    json.dump(out, sys.stdout, indent=1)
