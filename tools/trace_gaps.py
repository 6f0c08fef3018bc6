#!/usr/bin/env python
"""Per-kernel duration and the idle gap BEFORE each kernel, from a rocprofv3 --kernel-trace csv
(steady state: the middle half of the dispatches).  python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import OrderedDict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("at::native") or "at::native" in n[:40]:
        return "torch: " + n.split("<")[0].split("::")[-1]
    return n.split("(")[0][:70]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[n // 4: 3 * n // 4]
agg = OrderedDict()
prev_end = None
t_first, t_last = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))
    a = agg.setdefault(key, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    if prev_end is not None:
        a[2] += max(0, s - prev_end)
    prev_end = max(e, prev_end or 0)
tot_busy = sum(a[1] for a in agg.values())
tot_gap = sum(a[2] for a in agg.values())
print("window %.1f us, busy %.1f us (%.1f%%), gaps %.1f us" % ((t_last - t_first) / 1e3, tot_busy / 1e3,
      100.0 * tot_busy / (t_last - t_first), tot_gap / 1e3))
print("%-72s %10s %6s %8s %8s" % ("kernel", "grid", "calls", "avg_us", "gap_us"))
for (k, g), (c, d, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %10s %6d %8.2f %8.2f" % (k, g, c, d / c / 1e3, gp / c / 1e3))
