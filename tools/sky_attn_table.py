#!/usr/bin/env python3
"""Joins the three rocprofv3 passes of tools/gpu_r6_sky_attn.sh (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE over tools/sky_decode_once.py)
per ATTENTION DISPATCH of the measured decode (the second half of the process's attention dispatches: warm-up first), in launch order, and
names each dispatch's shape from the profiler-scope lines the program printed (same order of first appearance per kernel instantiation and
grid).  Corrections as tools/pmc_summary.py: FETCH_SIZE x2 on gfx950, KiB -> bytes."""
import csv
import glob
import json
import os
import re
import sys

PEAK = 2500.0


def rows_of(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not fs:
        raise SystemExit(f"no {pat} under {d}")
    return list(csv.DictReader(open(fs[0], newline="")))


if __name__ == "__main__":
    base = sys.argv[1]
    kt = [r for r in rows_of(os.path.join(base, "kt"), "*kernel_trace.csv") if "attention" in r["Kernel_Name"]]
    kt.sort(key=lambda r: int(r["Start_Timestamp"]))
    kt = kt[len(kt) // 2:]
    def pmc(d, name):
        rs = [r for r in rows_of(os.path.join(base, d), "*counter_collection.csv") if "attention" in r["Kernel_Name"] and r["Counter_Name"] == name]
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        return rs[len(rs) // 2:]
    pf, pw = pmc("pf", "FETCH_SIZE"), pmc("pw", "WRITE_SIZE")
    shapes = []
    for ln in open(os.path.join(base, "kt.log")):
        m = re.match(r"SHAPE (\S+) launches (\d+) us_per_launch ([\d.]+) algorithmic_bytes (\d+) gflop ([\d.]+)", ln)
        if m:
            shapes.append({"scope": m.group(1), "launches": int(m.group(2)), "us_hip_events": float(m.group(3)), "algorithmic_bytes": int(m.group(4)),
                           "gflop": float(m.group(5))})
    assert len(kt) == len(pf) == len(pw), (len(kt), len(pf), len(pw))
    disp = []
    for a, f, w in zip(kt, pf, pw):
        us = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
        disp.append({"kernel": a["Kernel_Name"].split("(")[0], "grid": int(a.get("Grid_Size_X", a.get("Grid_Size", 0)) or 0), "us": round(us, 1),
                     "hbm_bytes": int((2.0 * float(f["Counter_Value"]) + float(w["Counter_Value"])) * 1024),
                     "fetch_bytes": int(2.0 * float(f["Counter_Value"]) * 1024), "write_bytes": int(float(w["Counter_Value"]) * 1024)})
    # shape of a dispatch: the scope whose per-launch HIP-event time is closest (the decoder's attention shapes differ by >= 2x in time)
    for d in disp:
        s = min(shapes, key=lambda s: abs(s["us_hip_events"] - d["us"]) / max(s["us_hip_events"], 1.0)) if shapes else None
        if s:
            d["shape"] = s["scope"]
            d["algorithmic_bytes"] = s["algorithmic_bytes"]
            d["traffic_over_algorithmic"] = round(d["hbm_bytes"] / max(1, s["algorithmic_bytes"]), 2)
            d["tflops"] = round(s["gflop"] / d["us"] * 1e3, 1)   # GFLOP / us = PFLOP/s
            d["frac_of_2500"] = round(d["tflops"] / PEAK, 4)
    json.dump({"what": "VITAutoencoder.decode of ONE SkyTimelapse clip (bf16): every attention dispatch of the measured pass, in launch order",
               "head": os.environ.get("DDMI_HEAD"), "correction": "FETCH_SIZE x2 (gfx950), KiB -> bytes; three separate rocprofv3 passes",
               "scopes": shapes, "dispatches": disp}, sys.stdout, indent=1)
