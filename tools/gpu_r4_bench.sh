#!/bin/bash
# round 4: ops tests + the bench's config section, DMA weight gradients on (default) and off, same box
mkdir -p gpurun_out/r4h
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r4h/ops.log 2>&1; echo "ops rc=$?" >> gpurun_out/r4h/ops.log
tail -2 gpurun_out/r4h/ops.log
for v in 768 0 768 0; do
  GM_DW_DMA_MIN_K=$v timeout 600 python bench.py --no-cpu-baseline --steps 512 --warmup 64 --reps 3 > gpurun_out/r4h/bench_dma$v.$RANDOM.json 2> gpurun_out/r4h/bench_err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/bench_dma*.json')):
    try:
        j=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e:
        print(f,'parse error',e); continue
    out={'file':f.split('/')[-1],'ms_per_step':j.get('ms_per_step')}
    for c in j.get('configs',[]):
        out[c.get('config',{}).get('workload',c.get('name','?'))[:40]]=c.get('ms_per_step') or c.get('us_per_step')
    print(out)
PY
