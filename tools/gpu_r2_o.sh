#!/bin/bash
mkdir -p gpurun_out
GM_TRACE_RUN=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/o_short.json 2> gpurun_out/o_short.err
grep trace gpurun_out/o_short.err | cut -c1-500 | head -3
GM_FIRST_PIECE=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/o_short_fp4.json 2> /dev/null
GM_FIRST_PIECE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/o_short_fp1.json 2> /dev/null
GM_NATIVE_FILL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/o_short_py.json 2> /dev/null
timeout 300 python bench.py --no-configs > gpurun_out/o_long.json 2> gpurun_out/o_long.err
GM_GRAPH_ITERS=64 timeout 300 python bench.py --no-configs > gpurun_out/o_long_g64.json 2> /dev/null
for f in o_short o_short_fp4 o_short_fp1 o_short_py o_long o_long_g64; do python -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config']['reps_ms_per_step'])"; done
timeout 1200 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -x > gpurun_out/o_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/o_tests.log
tail -3 gpurun_out/o_tests.log
