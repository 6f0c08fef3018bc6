#!/bin/bash
mkdir -p gpurun_out
GM_XCD16=1 timeout 200 python tools/gemm_shapes_bench.py fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 fwd:300:784:400 dx:200:100:70 2>&1 | tail -7
for i in 1 2; do
for x in 0 1; do
GM_XCD16=$x timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 1000 --reps 3 > gpurun_out/q_x${x}_$i.json 2> /dev/null
python -c "
import json; d=json.loads(open('gpurun_out/q_x${x}_$i.json').read().strip().splitlines()[-1]); print('x$x', d['ms_per_step'], list(d['roofline']['per_kernel_us_per_step'].items()))"
done; done
