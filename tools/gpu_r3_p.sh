#!/bin/bash
# round 3, call P: warm pieces' ring uploads through the copy engine (GM_WARM_COPY) -- tests, then same-box A/B
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/r3p; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_ops.py -x -q -k "golden or oracle or ring or checkpoint or fail or gate or viz or two_call" ) > gpurun_out/r3p/tests.log 2>&1
tail -4 gpurun_out/r3p/tests.log
for rep in 1 2 3; do for w in 0 1; do
  GM_WARM_COPY=$w timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warm_copy=$w rep $rep long: %.2f us/step' % (d['ms_per_step']*1e3), d['config']['reps_ms_per_step'])"
done; done
for w in 0 1; do
  GM_WARM_COPY=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warm_copy=$w driver-style: %.2f us/step, fixed %.0f' % (d['ms_per_step']*1e3, d['run_fixed_cost_us']), d['config']['reps_ms_per_step'])"
  for c in ns_b1024 wgp_b256 dra_b256; do
  GM_WARM_COPY=$w timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warm_copy=$w $c:', [(round(e['img_s']), round(e['ms_per_step']*1e3, 2)) for e in d])"
  done
done
