#!/usr/bin/env python3
"""Per-kernel duration / gap statistics from a rocprofv3 kernel_trace.csv: trace_stats.py <csv> [name-substring]"""
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = [r for r in rows if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
g = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
print("kernels", len(rows), "duration ns: median", st.median(d), "min", min(d), "max", max(d))
if g:
    print("gap ns: median", st.median(g), "p90", sorted(g)[int(.9 * len(g))], "max", max(g))
print("grid/wg of first:", rows[0].get("Grid_Size"), rows[0].get("Workgroup_Size"), "LDS", rows[0].get("LDS_Block_Size"),
      "vgpr", rows[0].get("VGPR_Count"), "sgpr", rows[0].get("SGPR_Count"))
