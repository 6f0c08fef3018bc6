#!/bin/bash
# round 4: the one-kernel exchange -- DP tests (N ranks on one device == 1 rank), then the one-rank price of the
# data-parallel launch structure against the single-GPU step, one- vs two-kernel exchange, same box
mkdir -p gpurun_out/r4m
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_trainers.py -q -k "dp or peer" > gpurun_out/r4m/dp_tests.log 2>&1; echo "dp tests rc=$?"; tail -3 gpurun_out/r4m/dp_tests.log
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --sustained 0 --steps 2000 --warmup 200 --reps 3 2>/dev/null | tail -1 > gpurun_out/r4m/single.$rep.json
  GM_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --no-configs --sustained 0 --steps 2000 --warmup 200 --reps 3 2>/dev/null | tail -1 > gpurun_out/r4m/dp_one.$rep.json
  GM_FORCE_DP=1 GM_DP_TWO_KERNELS=1 timeout 300 python bench.py --no-cpu-baseline --no-configs --sustained 0 --steps 2000 --warmup 200 --reps 3 2>/dev/null | tail -1 > gpurun_out/r4m/dp_two.$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4m/*.json')):
    try: j=json.loads(open(f).read())
    except Exception as e: print(f, 'ERR', e); continue
    print(f.split('/')[-1], '%.2f us'%(j['ms_per_step']*1e3), j['config'].get('launch','')[:70])
PY
