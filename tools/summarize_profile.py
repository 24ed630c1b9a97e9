#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a small text/JSON report."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


rep = {}
for f in find("ktrace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for r in rows[:12]:
        print("  %-70s calls=%s avg_ns=%s total_ns=%s pct=%s" % (r.get("Name", "")[:70], r.get("Calls"),
              r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
    rep["kernel_stats"] = rows[:12]
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== counters (%s)" % os.path.relpath(f, out))
        for k, cs in agg.items():
            for c, vals in cs.items():
                print("  %-60s %-22s n=%d mean=%.6g" % (k[:60], c, len(vals), sum(vals) / len(vals)))
                rep.setdefault("counters", {}).setdefault(k, {})[c] = {"n": len(vals), "mean": sum(vals) / len(vals)}
json.dump(rep, open(os.path.join(out, "summary.json"), "w"), indent=1)
