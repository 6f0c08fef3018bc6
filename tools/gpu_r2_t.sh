#!/bin/bash
mkdir -p gpurun_out
GM_BENCH_VERBOSE=1 timeout 900 python bench.py > gpurun_out/t_bench_default.json 2> gpurun_out/t_bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs > gpurun_out/t_bench_short.json 2> /dev/null; echo "short rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/t_bench_default.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['value'], d['cpu_baseline']['value'], d['cpu_baseline_compute_only']['value'])
for c in d.get('configs',[]): print(c['workload'][:70], round(c['img_s']), round(c['ms_per_step'],4), (c.get('cpu_baseline') or {}).get('value'))
s=json.loads(open('gpurun_out/t_bench_short.json').read().strip().splitlines()[-1])
print('short', s['ms_per_step'], s['config']['reps_ms_per_step'])
PY
