#!/bin/bash
mkdir -p gpurun_out
GM_TRACE_RUN=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/m_short.json 2> gpurun_out/m_short.err
grep trace gpurun_out/m_short.err | cut -c1-700 | head -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/m_short2.json 2> /dev/null
timeout 300 python bench.py --no-configs > gpurun_out/m_long.json 2> gpurun_out/m_long.err
for f in m_short m_short2 m_long; do python -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config']['reps_ms_per_step'], list(d['roofline']['per_kernel_us_per_step'].values()))"; done
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -x -k "golden or oracle or resume" > gpurun_out/m_tests.log 2>&1; echo "trainers rc=$?" >> gpurun_out/m_tests.log
tail -3 gpurun_out/m_tests.log
