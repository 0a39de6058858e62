#!/usr/bin/env python3
"""Cuts the LAST UNet evaluation out of a rocprofv3 kernel_trace.csv (it starts at the last timestep_embedding_kernel) and prints one
line per launch: start offset, duration, gap to the previous launch's end, grid / workgroup size, LDS, kernel name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "timestep_embedding" in r["Kernel_Name"])
ev = rows[last:]
t0 = int(ev[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in ev:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  wgs {grid // max(1, wg):6d} x {wg:4d}  lds {r.get('LDS_Block_Size', '?'):>6}  {r['Kernel_Name'][:90]}")
    busy += e - s
    prev_end = e
print(f"# {len(ev)} launches, busy {busy / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
