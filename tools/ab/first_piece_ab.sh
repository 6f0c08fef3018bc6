#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do for fp in 2 4 1; do
GM_FIRST_PIECE=$fp timeout 200 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 --reps 9 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp$fp', round(d['ms_per_step']*1e3,2), [round(x*1e3,1) for x in d['config']['reps_ms_per_step']])"
done; done
