#!/bin/bash
# WGAN-GP bs=256: tests of the touched kernels, then throughput with / without the interpolate epilogue
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -q -x > gpurun_out/wgp_ops.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/wgp_ops.log
timeout 900 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -x -k "wgp or WGP or w_gp" > gpurun_out/wgp_tr.log 2>&1; echo "trainers rc=$?"; tail -2 gpurun_out/wgp_tr.log
for i in 1 2; do for x in 1 0; do
GM_WGP_INTERP_EPI=$x timeout 300 python bench.py --only wgp_b256 --steps 400 --warmup 50 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys
for e in json.loads(sys.stdin.read()): print('interp_epi=$x', round(e['img_s']), round(e['ms_per_step']*1e3,2), e['reps_ms_per_step'])"
done; done
