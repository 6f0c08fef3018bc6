#!/bin/bash
# gated stage-in: parity tests, then short-run trace and long run
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -x > gpurun_out/k_tests.log 2>&1; echo "trainers rc=$?" >> gpurun_out/k_tests.log
tail -4 gpurun_out/k_tests.log
GM_TRACE_RUN=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/k_short.json 2> gpurun_out/k_short.err
grep trace gpurun_out/k_short.err | cut -c1-900
GM_GATED=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/k_short_ungated.json 2> /dev/null
timeout 300 python bench.py --no-configs > gpurun_out/k_long.json 2> gpurun_out/k_long.err
for f in k_short k_short_ungated k_long; do python -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config']['reps_ms_per_step'], list(d['roofline']['per_kernel_us_per_step'].values()))"; done
