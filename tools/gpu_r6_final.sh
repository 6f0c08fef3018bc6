#!/bin/bash
# Round 6: the two bench lines that are committed (driver's flags; defaults) + smoke
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/final_r06
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_r06/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_r06/r06_bench_steps20_warmup5.json 2> gpurun_out/final_r06/bench20.err; echo "bench20 rc=$?"
timeout 900 python bench.py > gpurun_out/final_r06/r06_bench_default.json 2> gpurun_out/final_r06/bench_default.err; echo "bench default rc=$?"
python - <<'PY'
import json
for f in ("r06_bench_steps20_warmup5.json", "r06_bench_default.json"):
    try:
        d = json.loads(open("gpurun_out/final_r06/" + f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("steady_us_per_step"), d.get("run_fixed_cost_us"))
        for c in d["config"].get("other_configs", d.get("configs", [])) or []:
            print("   ", c.get("workload", "")[:60], c.get("ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
PY
