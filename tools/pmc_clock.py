#!/usr/bin/env python3
"""Effective shader clock per kernel from one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass (CSV output):
clock = sum(GRBM_GUI_ACTIVE) / 8 XCDs / sum(dispatch duration)  (MI355X_MICROARCH.md, 'DVFS give-back': rocprofv3 reports the
sum over the 8 XCDs; reads high on dispatches shorter than ~0.3 ms).  The dense-MFMA peak scales with it: 2.5 PFLOP/s is quoted
at 2.4 GHz.   usage: tools/pmc_clock.py <rocprof dir> "<command text>" > profiles/rN_effective_clock.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

if __name__ == "__main__":
    d, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    dur = {}
    for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = defaultdict(lambda: [0, 0.0, 0])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
                    continue
                if "Start_Timestamp" in r and r["Start_Timestamp"]:
                    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                elif r["Dispatch_Id"] in dur:
                    ns = dur[r["Dispatch_Id"]][1]
                else:
                    continue
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] += ns
    out = {"command": cmd, "head": os.environ.get("DDMI_HEAD"), "method": "sum(GRBM_GUI_ACTIVE) / 8 / sum(dispatch ns); nominal 2.4 GHz <-> 2.5 PFLOP/s dense bf16", "kernels": {}}
    for k, (n, cyc, ns) in sorted(acc.items(), key=lambda kv: -kv[1][2])[:24]:
        if ns <= 0:
            continue
        ghz = cyc / 8.0 / ns
        out["kernels"][k] = {"dispatches": n, "avg_us": round(ns / n / 1e3, 1), "effective_ghz": round(ghz, 3),
                             "mfma_peak_at_clock_tflops": round(2500.0 * ghz / 2.4, 0)}
    json.dump(out, sys.stdout, indent=1)
