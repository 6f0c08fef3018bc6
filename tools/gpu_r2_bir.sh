#!/bin/bash
# BIR-VAE product tests + quick headline A/B (same box) for the long-run step time
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -x -k "bir" > gpurun_out/bir_tests.log 2>&1
echo "bir trainers rc=$?" >> gpurun_out/bir_tests.log
timeout 300 python -m pytest tests/test_gpu_fused_ops.py -q -x -k "bir" >> gpurun_out/bir_tests.log 2>&1
echo "bir ops rc=$?" >> gpurun_out/bir_tests.log
for i in 1 2; do
  timeout 300 python bench.py --no-configs > gpurun_out/ab_default_$i.json 2> gpurun_out/ab_default_$i.err
  GM_PACKED=0 timeout 300 python bench.py --no-configs > gpurun_out/ab_nopack_$i.json 2> gpurun_out/ab_nopack_$i.err
done
tail -5 gpurun_out/bir_tests.log
for f in gpurun_out/ab_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done
