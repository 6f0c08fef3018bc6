#!/bin/bash
mkdir -p gpurun_out
GM_TRACE_RUN=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/p_short.json 2> gpurun_out/p_short.err
grep trace gpurun_out/p_short.err | cut -c1-400 | head -2
GM_FIRST_PIECE=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/p_short_fp4.json 2> /dev/null
GM_FIRST_PIECE=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/p_short_fp8.json 2> /dev/null
timeout 300 python bench.py --no-configs > gpurun_out/p_long.json 2> gpurun_out/p_long.err
GM_GRAPH_ITERS=64 GM_RING=256 timeout 300 python bench.py --no-configs > gpurun_out/p_long_g64r256.json 2> /dev/null
GM_GRAPH_ITERS=128 GM_RING=256 timeout 300 python bench.py --no-configs > gpurun_out/p_long_g128r256.json 2> /dev/null
GM_GRAPH_ITERS=128 GM_RING=512 timeout 300 python bench.py --no-configs > gpurun_out/p_long_g128r512.json 2> /dev/null
GM_GRAPH_ITERS=32 GM_RING=512 timeout 300 python bench.py --no-configs > gpurun_out/p_long_g32r512.json 2> /dev/null
for f in p_short p_short_fp4 p_short_fp8 p_long p_long_g64r256 p_long_g128r256 p_long_g128r512 p_long_g32r512; do python -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config']['reps_ms_per_step'])"; done
