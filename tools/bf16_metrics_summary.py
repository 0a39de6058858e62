#!/usr/bin/env python3
"""Worst case per `assert_bf16_close` label of one `DDMI_METRICS_LOG=<file> pytest -m gpu` run -> the committed profiles/*_bf16_metrics_all.jsonl:
   python tools/bf16_metrics_summary.py <raw log> <commit> > profiles/r4_bf16_metrics_all.jsonl"""
import json
import sys

LIMITS = {"linf": 0.04, "rel_l2": 0.02, "min_channel_cos": 0.999}

if __name__ == "__main__":
    if sys.argv[1] == "--baselines":   # -> tests/golden/bf16_metric_baselines.json: {label: {linf, rel_l2}} (worst per label), tests/helpers.py
        worst = {}
        for r in (json.loads(line) for line in open(sys.argv[2]) if line.strip()):
            w = worst.setdefault(r["what"], {"linf": 0.0, "rel_l2": 0.0})
            w["linf"] = round(max(w["linf"], r["linf"]), 6)
            w["rel_l2"] = round(max(w["rel_l2"], r["rel_l2"]), 6)
        print(json.dumps(worst, indent=0, sort_keys=True))
        sys.exit(0)
    rows = [json.loads(line) for line in open(sys.argv[1]) if line.strip()]
    head = sys.argv[2] if len(sys.argv) > 2 else "unknown"
    worst = {}
    for r in rows:
        w = worst.setdefault(r["what"], {"what": r["what"], "calls": 0, "linf": 0.0, "rel_l2": 0.0, "min_channel_cos": 1.0})
        w["calls"] += 1
        w["linf"] = max(w["linf"], r["linf"])
        w["rel_l2"] = max(w["rel_l2"], r["rel_l2"])
        w["min_channel_cos"] = min(w["min_channel_cos"], r["min_channel_cos"])
    print(json.dumps({"_": f"worst case per assert_bf16_close label over one full `pytest -m gpu` run (commit {head}; plan-deterministic tests: DDMI_TUNE_STRICT=1, "
                           "shipped plans / heuristic plans, no per-box autotune); limits: linf < 4e-2 of range, rel_l2 < 2.0e-2, min per-channel cosine > 0.999",
                      "limits": LIMITS}))
    for w in worst.values():
        frac = max(w["linf"] / LIMITS["linf"], w["rel_l2"] / LIMITS["rel_l2"], (1.0 - w["min_channel_cos"]) / (1.0 - LIMITS["min_channel_cos"]))
        print(json.dumps({"what": w["what"], "calls": w["calls"], "linf": round(w["linf"], 6), "rel_l2": round(w["rel_l2"], 6),
                          "min_channel_cos": round(w["min_channel_cos"], 6), "fraction_of_tightest_limit": round(frac, 3)}))
