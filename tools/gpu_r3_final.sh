#!/bin/bash
# round 3: whole GPU suite, smoke, bench records (default + driver-style), dry runs of the N > 1 bench flow
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/r3final; export TMPDIR=/tmp
OUT=$R/gpurun_out/r3final
if [ -z "$SKIP_TESTS" ]; then
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/gpu_all.log 2>&1
tail -5 $OUT/gpu_all.log
fi
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/r03_bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_steps20_warmup5.json 2> $OUT/bench_s20.err; echo "bench s20 rc=$?"
for n in ${DRY_NS:-2 4 8}; do
  GM_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $n --steps 20 --warmup 5 > $OUT/r03_bench_dry_n$n.json 2> $OUT/bench_dry_n$n.err; echo "dry run N=$n rc=$?"
done
python - <<'PY'
import json
for f in ("r03_bench_default", "r03_bench_steps20_warmup5", "r03_bench_dry_n2", "r03_bench_dry_n4", "r03_bench_dry_n8"):
    try:
        d = json.loads(open("gpurun_out/r3final/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "%.2f us/step" % (d["ms_per_step"] * 1e3), round(d["value"]), d["config"].get("ranks_seen"), d.get("steady_us_per_step"), d.get("run_fixed_cost_us"))
        if "trainer" in d: print("  trainer", d["trainer"]["ms_per_step"] * 1e3)
        for c in d.get("configs", []): print("  ", c["workload"][:44], round(c["img_s"]), round(c["ms_per_step"] * 1e3, 2), c.get("roofline", {}).get("kernel"), c.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
