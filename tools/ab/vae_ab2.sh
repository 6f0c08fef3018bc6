#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -x -k "vae or ae or bir or VAE or reparam" > gpurun_out/vae2.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/vae2.log | tail -2
for i in 1 2; do
timeout 300 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 2>/dev/null | grep workload | python -c "
import json,sys
es=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(round(e['img_s']), round(e['ms_per_step']*1e3,1)) for e in es])"
done
