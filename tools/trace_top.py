#!/usr/bin/env python
"""Longest kernels and largest idle gaps (with their neighbours) in a rocprofv3 kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
t00 = int(rows[0]["Start_Timestamp"])
longest = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:8]
print("longest kernels:")
for r in longest:
    print("  %-60s %10.1f us at t=%.1f ms grid %s" % (nm(r), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          (int(r["Start_Timestamp"]) - t00) / 1e6, r.get("Grid_Size", "")))
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), a, b))
gaps.sort(key=lambda g: -g[0])
print("largest gaps:")
for g, a, b in gaps[:12]:
    print("  %10.1f us at t=%.1f ms between %s -> %s" % (g / 1e3, (int(a["End_Timestamp"]) - t00) / 1e6, nm(a), nm(b)))
