#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do for x in 1 0; do
GM_SPLIT_STAGE=$x timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 2000 --reps 3 > gpurun_out/r_long_s${x}_$i.json 2> gpurun_out/r_long_s${x}_$i.err
GM_SPLIT_STAGE=$x timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r_short_s${x}_$i.json 2> /dev/null
for f in r_long_s${x}_$i r_short_s${x}_$i; do python -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config']['reps_ms_per_step'])"; done
done; done
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -x -k "golden or resume or 50" > gpurun_out/r_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r_tests.log
tail -3 gpurun_out/r_tests.log
