#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/ad
GM_TRACE_RUN=1 timeout 200 python bench.py --steps 20 --warmup 5 --reps 6 --no-cpu-baseline --no-configs --sustained 0 > gpurun_out/ad/out.json 2> gpurun_out/ad/trace.txt
grep 'trace rep' gpurun_out/ad/trace.txt | cut -c1-1600
python -c "import json; d=json.loads(open('gpurun_out/ad/out.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['config'].get('reps_ms_per_step'), d['config'].get('run_fixed_cost_us'))"
