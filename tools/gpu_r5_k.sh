#!/bin/bash
# Round 5, call K: pre-stage completion through the pinned word gate[3]: trainer + DP tests, then the 20-step figure
# alternating with no pre-staging at all (the only same-call baseline left), then the committed lines again.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out/final_r05; export TMPDIR=/tmp
run20() { echo "$*: $(env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"; }
for rep in 1 2 3; do run20 GM_PRESTAGE=1; run20 GM_PRESTAGE=0; done
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -x -k "engine_vs_oracle or small_ring or golden or failed_host or checkpoint or determinism" > gpurun_out/k_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/k_tests.log | cut -c1-200
bash tools/gpu_r5_final.sh
