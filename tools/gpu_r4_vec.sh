#!/bin/bash
# round 4: float4 epilogues -- ops tests, then the bench's steady numbers old library vs new, alternating on one box
mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -x -q > gpurun_out/r4i/ops.log 2>&1; echo "ops rc=$?" >> gpurun_out/r4i/ops.log
tail -2 gpurun_out/r4i/ops.log
for rep in 1 2; do
for lib in new old; do
  if [ $lib = old ]; then export GM_LIB_PATH=$PWD/generative_models_amd/ab_libs/r4_dma.so; else unset GM_LIB_PATH; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 512 --warmup 64 --reps 3 > gpurun_out/r4i/bench_$lib.$rep.json 2>> gpurun_out/r4i/bench_err.log
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4i/bench_*.json')):
    try: j=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'parse error',e); continue
    print(f.split('/')[-1], 'headline %.2f us'%(j['ms_per_step']*1e3), ' | '.join('%s %.1f'%(c['workload'][:14],c['ms_per_step']*1e3) for c in j.get('configs',[])))
PY
