"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run (gpurun_out/<dir>) into the
small markdown + csv kept under profiles/.  Usage: python profiles/make_summary.py <run_dir> <tag> [outdir]"""
import csv
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("at::native") or "at::native" in name[:40]:
        return "torch: " + name.split("<")[0].split("::")[-1] + "<...>"
    return name.split("(")[0]


def main(run_dir, tag, outdir=None):
    outdir = outdir or os.path.dirname(os.path.abspath(__file__))
    stats = [f for f in os.listdir(run_dir) if f.endswith("kernel_stats.csv")][0]
    rows = list(csv.DictReader(open(os.path.join(run_dir, stats))))
    trace = [f for f in os.listdir(run_dir) if f.endswith("kernel_trace.csv")]
    out_csv = os.path.join(outdir, tag + "_kernel_stats.csv")
    with open(out_csv, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.1f" % (int(r["TotalDurationNs"]) / 1e3),
                        "%.3f" % (float(r["AverageNs"]) / 1e3), r["Percentage"],
                        "%.2f" % (int(r["MinNs"]) / 1e3), "%.2f" % (int(r["MaxNs"]) / 1e3)])
    lines = ["# rocprofv3 --kernel-trace --stats summary: %s" % tag, "",
             "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in rows[:12]:
        lines.append("| `%s` | %s | %.2f | %.2f | %.2f | %s |" % (
            short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, int(r["MinNs"]) / 1e3,
            int(r["MaxNs"]) / 1e3, r["Percentage"]))
    if trace:
        # per launch-shape averages of the GEMM kernel (grid size identifies the layer)
        t = list(csv.DictReader(open(os.path.join(run_dir, trace[0]))))
        agg = {}
        for r in t:
            if "gemm" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                   r.get("Grid_Size_Y", ""))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg.setdefault(key, [0, 0])
            a[0] += 1
            a[1] += d
        lines += ["", "GEMM launches by grid (threads x, y):", "",
                  "| kernel | grid | calls | avg us |", "|---|---|---|---|"]
        for k, (n, tot) in sorted(agg.items()):
            lines.append("| `%s` | %s x %s | %d | %.2f |" % (k[0], k[1], k[2], n, tot / n / 1e3))
    open(os.path.join(outdir, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:4])
