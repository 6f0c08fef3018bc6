#!/bin/bash
# Round 4: pre-staging of the host draws (gm_stage_in_prestaged) and the cold-start plan of a 20-step run
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "stage_in" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
fi
run() {
  echo "$*: $(env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"
}
for rep in 1 2; do
run GM_PRESTAGE=1 GM_FIRST_PIECE=2
run GM_PRESTAGE=1 GM_FIRST_PIECE=4
run GM_PRESTAGE=1 GM_FIRST_PIECE=4 GM_RAMP=2,2,4,8,16
run GM_PRESTAGE=1 GM_FIRST_PIECE=4 GM_RAMP=4,4,8,16
run GM_PRESTAGE=0 GM_FIRST_PIECE=4
run GM_PRESTAGE=0 GM_FIRST_PIECE=2
done
