#!/bin/bash
# Round 5, call L: the extended lockstep cases; does a faster host replay (GM_HOST_THREADS) let a cold run use bigger first pieces?
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
run20() { echo "$*: $(env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"; }
for rep in 1 2; do
  run20 GM_HOST_THREADS=1
  run20 GM_HOST_THREADS=4
  run20 GM_HOST_THREADS=4 GM_FIRST_PIECE=4
  run20 GM_HOST_THREADS=4 GM_FIRST_PIECE=4 GM_RAMP=4,16
  run20 GM_HOST_THREADS=8 GM_FIRST_PIECE=4 GM_RAMP=4,16
done
rm -f gpurun_out/parity_full_size.jsonl
timeout 600 python -m pytest tests/test_gpu_trainers.py -q -k "lockstep" > gpurun_out/l_tests.log 2>&1; echo "lockstep tests rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/l_tests.log | cut -c1-250 | tail
