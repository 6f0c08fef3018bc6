#!/bin/bash
# Round-end measurement on the GPU box: full GPU test suite, smoke, the driver's bench command, the
# default bench line (with cpu_baseline and the configs section).  Profiles: tools/gpu_r2_profiles.sh.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/final_pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/final_smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench_short.json 2> gpurun_out/final_bench_short.err; echo "short bench rc=$?"
GM_BENCH_VERBOSE=1 timeout 900 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
s=json.loads(open('gpurun_out/final_bench_short.json').read().strip().splitlines()[-1])
print('driver-style', s['ms_per_step'], s['config']['reps_ms_per_step'], 'has configs' if 'configs' in s else 'no configs', 'cpu', s.get('cpu_baseline',{}).get('value'))
d=json.loads(open('gpurun_out/final_bench_default.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
for c in d.get('configs',[]): print(c['workload'][:70], round(c['img_s']), round(c['ms_per_step'],4))
PY
