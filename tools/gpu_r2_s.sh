#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_fused_ops.py tests/test_gpu_dp.py -q -x -k "vae or ae or bir or VAE" > gpurun_out/s_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s_tests.log
tail -3 gpurun_out/s_tests.log
timeout 300 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "
import json,sys
for e in json.loads(sys.stdin.read().strip().splitlines()[-1]): print(e['workload'][:60], round(e['img_s']), e['ms_per_step'])"
timeout 300 python bench.py --only dra_b256 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "
import json,sys
for e in json.loads(sys.stdin.read().strip().splitlines()[-1]): print(e['workload'][:60], round(e['img_s']), e['ms_per_step'], e['reps_ms_per_step'])"
