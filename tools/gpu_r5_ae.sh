#!/bin/bash
# Round 5, call AE: first fills submitted at run() entry (GM_EARLY_SUBMIT), the driver's 20-step line either way
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/ae
timeout 200 python -m pytest tests/test_gpu_trainers.py -q -p no:cacheprovider -x -k "golden or determinism or eager" > gpurun_out/ae/tests.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/ae/tests.log
for rep in 1 2 3 4; do for e in 1 0; do
  echo "GM_EARLY_SUBMIT=$e 20 steps: $(GM_EARLY_SUBMIT=$e timeout 200 python bench.py --steps 20 --warmup 5 --reps 15 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("run_fixed_cost_us"), sorted(d["config"].get("reps_ms_per_step"))[:8])')"
done; done 2>&1 | tee gpurun_out/ae/early_submit_ab.txt
