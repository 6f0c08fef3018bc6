#!/usr/bin/env python
"""Average SQ counters per kernel from a rocprofv3 --pmc ... --kernel-trace csv run.
python tools/pmc_sq_summary.py <dir> [out.json]   (the json: {"kernel|grid": {counter: average per dispatch}},
what bench.py reads roofline.mfma_busy_frac from)"""
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if name.startswith("at::") or "rocclr" in name:
        continue
    a = agg[(name, r["Grid_Size"])][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for (k, g), cs in agg.items():
    print("%s grid %s" % (k[:80], g))
    wc = cs.get("SQ_WAVE_CYCLES", [1, 0])
    wcv = wc[1] / max(wc[0], 1)
    for c, (n, tot) in sorted(cs.items()):
        v = tot / n
        print("   %-28s %14.0f  %s" % (c, v, ("%.1f%% of WAVE_CYCLES" % (100 * v / wcv)) if wcv and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""))
if len(sys.argv) > 2:
    json.dump({"%s|%s" % (k, g): {c: tot / n for c, (n, tot) in sorted(cs.items())} for (k, g), cs in agg.items()},
              open(sys.argv[2], "w"), indent=1)
