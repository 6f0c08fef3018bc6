#!/bin/bash
# Round 5, call C: stage-ahead v2 (riders poll the pre-staged range; standalone rider when the pair cannot be one launch),
# the whole trainer + DP test files, the 20-step A/B.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
timeout 900 python -m pytest tests/test_gpu_trainers.py -q > gpurun_out/c_tests1.log 2>&1; echo "tests trainers rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/c_tests1.log | cut -c1-220 | tail -25
timeout 900 python -m pytest tests/test_gpu_dp.py -q > gpurun_out/c_tests2.log 2>&1; echo "tests dp rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/c_tests2.log | cut -c1-220 | tail -12
for rep in 1 2; do for v in 1 0; do
  echo "GM_STAGE_AHEAD=$v: $(GM_STAGE_AHEAD=$v timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"
done; done
GM_TRACE_RUN=1 timeout 200 python bench.py --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2> gpurun_out/c_trace.err > /dev/null; grep 'trace rep' gpurun_out/c_trace.err | cut -c1-1500 | tail -2
