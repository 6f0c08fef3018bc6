#!/usr/bin/env python
"""Joins tools/fill_probe.bin's timing with the rocprofv3 --pmc passes of tools/fill_law.sh.

  fill_law_report.py --reduce <counter_collection.csv>   -> "kernel,counter,avg per dispatch,dispatches" (small; kept)
  fill_law_report.py <gpurun_out/fill_law>               -> markdown (profiles/r05_fill_law.md)
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"k_fill<(\d+), (\d+), (\d+), (\d+)>", name)
    return "%s/%sw/d%s/p%s" % m.groups() if m else None


def reduce(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        a = agg[(k, r["Grid_Size"], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid", "counter", "avg", "n"])
    for (k, g, c), (n, tot) in sorted(agg.items()):
        w.writerow([k, g, c, "%.6g" % (tot / n), n])


def report(d):
    timing = {}
    clock = None
    for line in open(os.path.join(d, "timing.txt")):
        m = re.match(r"device .* clock (\d+) MHz", line)
        if m:
            clock = float(m.group(1))
        m = re.match(r"(\S+)\s+wgs\s+(\d+) waves/wg\s+(\d+) depth (\d+)\s+([\d.]+) us\s+([\d.]+) GB/s per CU\s+([\d.]+) TB/s chip\s+([\d.]+) cycles", line)
        if m:
            name = m.group(1)
            key = "/".join(name.split("/")[:4])
            wgs = int(m.group(2))
            timing[(key, wgs)] = dict(name=name, wgs=wgs, waves=int(m.group(3)), depth=int(m.group(4)), us=float(m.group(5)),
                                      gbs=float(m.group(6)), tbs=float(m.group(7)), cyc=float(m.group(8)))
    ctr = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "pmc_*.csv"))):
        for r in csv.DictReader(open(f)):
            wgs = int(r["grid"]) // (int(r["kernel"].split("/")[1][:-1]) * 64)
            ctr[(r["kernel"], wgs)][r["counter"]] = float(r["avg"])
    print("# Fill law: bytes per second one CU pulls from L2, by waves per CU and loads in flight (tools/fill_probe.hip)\n")
    print("Shader clock reported by the runtime: %s MHz.  One row per kernel instantiation; timing from the probe's own "
          "HIP events (20 launches), counters from separate `rocprofv3 --pmc` passes (3 launches each, averaged per dispatch).\n" % clock)
    print("name = mode (0 VGPR `global_load_dwordx4`, 1 LDS-DMA `global_load_lds_dwordx4`) / waves per workgroup / loads in flight per wave / "
          "pattern (0: 1 KB contiguous per instruction from a shared 3 MB buffer; 1: 16 rows x 64 B at 3136 B stride; 2: 16 KB per workgroup, L1-resident)\n")
    hdr = ["config", "WGs", "GB/s per CU", "TB/s chip", "cycles / wave instr", "TA busy %", "TA addr-stalled-by-TC %", "TA data-stalled-by-TC %",
           "TD busy %", "TCP busy (GATE_EN2/EN1) %", "TCP pending-stall %", "TCP->TCC read latency (cycles)", "TCP latency (cycles)", "TCC hit %", "TCC busy %",
           "SQ wait-any % of wave cycles", "VMEM level (avg in flight per CU)"]
    print("| " + " | ".join(hdr) + " |")
    print("|" + "---|" * len(hdr))

    def pct(a, b):
        return "%.0f" % (100.0 * a / b) if a is not None and b else "-"
    for key, t in timing.items():
        c = ctr.get(key, {})
        g = c.get
        kernel_cycles = t["us"] * (clock or 2400.0)
        n_cu = min(t["wgs"], 256)
        # *_sum counters add up all instances (one TA/TD/TCP per CU; 16 TCC channels per XCD x 8)
        ta_busy = pct(g("TA_TA_BUSY_sum"), kernel_cycles * 256) if g("TA_TA_BUSY_sum") else "-"
        row = [t["name"], t["wgs"], "%.1f" % t["gbs"], "%.2f" % t["tbs"], "%.1f" % t["cyc"], ta_busy,
               pct(g("TA_ADDR_STALLED_BY_TC_CYCLES_sum"), g("TA_TA_BUSY_sum")), pct(g("TA_DATA_STALLED_BY_TC_CYCLES_sum"), g("TA_TA_BUSY_sum")),
               pct(g("TD_TD_BUSY_sum"), kernel_cycles * 256), pct(g("TCP_GATE_EN2_sum"), g("TCP_GATE_EN1_sum")),
               pct(g("TCP_PENDING_STALL_CYCLES_sum"), g("TCP_GATE_EN1_sum") or kernel_cycles * 256),
               "%.0f" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")) if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum") else "-",
               "%.0f" % (g("TCP_TCP_LATENCY_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")) if g("TCP_TCP_LATENCY_sum") and g("TCP_TOTAL_CACHE_ACCESSES_sum") else "-",
               pct(g("TCC_HIT_sum"), (g("TCC_HIT_sum") or 0) + (g("TCC_MISS_sum") or 0)),
               pct(g("TCC_BUSY_sum"), kernel_cycles * 128),
               pct(g("SQ_WAIT_INST_ANY"), g("SQ_WAVE_CYCLES")),
               "%.1f" % (g("SQ_INST_LEVEL_VMEM") / g("SQ_BUSY_CU_CYCLES")) if g("SQ_INST_LEVEL_VMEM") and g("SQ_BUSY_CU_CYCLES") else
               ("%.1f" % (g("SQ_INST_LEVEL_VMEM") / (kernel_cycles * n_cu)) if g("SQ_INST_LEVEL_VMEM") else "-")]
        print("| " + " | ".join(str(x) for x in row) + " |")
    print("\n## Raw counter averages per dispatch\n")
    names = sorted({c for v in ctr.values() for c in v})
    print("| config | WGs | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for key, t in timing.items():
        c = ctr.get(key, {})
        print("| %s | %d | " % (t["name"], t["wgs"]) + " | ".join("%.4g" % c[n] if n in c else "-" for n in names) + " |")


if __name__ == "__main__":
    if sys.argv[1] == "--reduce":
        reduce(sys.argv[2])
    else:
        report(sys.argv[1])
