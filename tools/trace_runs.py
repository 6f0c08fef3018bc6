#!/usr/bin/env python
"""Timeline of short timed runs from a rocprofv3 --kernel-trace csv: the dispatches are split into bursts at idle gaps
longer than `--idle` us (a burst = one run() between two synchronizations); for the last few bursts of a given
length print, per iteration (split at the stage-in / first kernel of an iteration by launch count), the time from the
burst's first kernel, the busy time and the gaps.   python tools/trace_runs.py <kernel_trace.csv> [--idle 200]"""
import csv
import sys

idle_us = 200.0
if "--idle" in sys.argv:
    idle_us = float(sys.argv[sys.argv.index("--idle") + 1])
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bursts, cur, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > idle_us * 1e3 and cur:
        bursts.append(cur)
        cur = []
    cur.append((s, e, r["Kernel_Name"]))
    prev_end = max(e, prev_end or 0)
if cur:
    bursts.append(cur)
print("%d bursts; kernels per burst: %s" % (len(bursts), [len(b) for b in bursts][-12:]))
for b in bursts[-4:]:
    t0 = b[0][0]
    span = (b[-1][1] - t0) / 1e3
    busy = sum(e - s for s, e, _ in b) / 1e3
    print("burst: %d kernels, span %.1f us, busy %.1f us" % (len(b), span, busy))
    # gaps > 3 us inside the burst (graph boundaries, gate waits)
    pe = None
    for i, (s, e, n) in enumerate(b):
        if pe is not None and s - pe > 3000:
            print("   gap %.1f us before kernel %d (%s) at +%.1f us" % ((s - pe) / 1e3, i, n.split("(")[0][:50], (s - t0) / 1e3))
        pe = max(e, pe or 0)
    for i, (s_, e_, n_) in enumerate(b):                 # the stage-in launches: where, how long, what gap in front
        if "stage_in" in n_:
            gap = (s_ - max(x[1] for x in b[:i])) / 1e3 if i else 0.0
            print("   stage-in at kernel %d: +%.1f us, %.1f us long, gap in front %.1f us" % (i, (s_ - t0) / 1e3, (e_ - s_) / 1e3, gap))
    # slow kernels: duration per launch position vs the median of the same kernel name in the burst
    by = {}
    for s, e, n in b:
        by.setdefault(n, []).append((e - s) / 1e3)
    med = {n: sorted(v)[len(v) // 2] for n, v in by.items()}
    extra = [(i, (e - s) / 1e3 - med[n]) for i, (s, e, n) in enumerate(b)]
    head = sum(x for i, x in extra[:len(b) // 10])
    print("   first tenth of the kernels: %.1f us above their medians; whole burst %.1f us" % (head, sum(x for _, x in extra)))
    worst = sorted(extra, key=lambda t: -t[1])[:6]
    print("   largest excesses: %s" % [(i, round(x, 1), b[i][2].split("(")[0][-40:]) for i, x in worst])
