#!/bin/bash
# Round 5, call H (after the critic-step split and the host-side pre-stage events): the whole GPU suite on the cleaned-up build (x-coalesced loads, no stage-ahead rider, RCCL in graph),
# then the driver's bench command and a long run.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/h_tests.log 2>&1; echo "gpu suite rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/h_tests.log | cut -c1-220 | tail -15
for rep in 1 2; do
  echo "20 steps: $(timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"
done
