#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -x -k "vae or ae or bir or VAE" > gpurun_out/vae_pf.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/vae_pf.log | tail -2
for i in 1 2 3; do for x in 1 0; do
GM_VAE_PREFETCH_GATHER=$x timeout 300 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 2>/dev/null | grep workload | python -c "
import json,sys
es=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch=$x', [(round(e['img_s']), round(e['ms_per_step']*1e3,1)) for e in es])"
done; done
