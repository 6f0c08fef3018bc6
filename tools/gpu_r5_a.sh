#!/bin/bash
# Round 5, call A: the new parity tests, the fill-law probe with counters, a baseline bench line of this box.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -x -k "default_batch or lockstep or hidden_wider" > gpurun_out/a_tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 gpurun_out/a_tests1.log
timeout 600 python -m pytest tests/test_gpu_dp.py -q -k "one_kernel or peer_allreduce" > gpurun_out/a_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/a_tests2.log
bash tools/fill_law.sh > gpurun_out/a_fill.log 2>&1; tail -45 gpurun_out/a_fill.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/a_bench20.json 2> gpurun_out/a_bench20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/a_bench20.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','steady_us_per_step','run_fixed_cost_us')}, d['roofline'].get('avg_launch_us'))
PY
