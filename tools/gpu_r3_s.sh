#!/bin/bash
# round 3, call S: 48 tiles also for short reductions (GM_DW_TILE48=2) vs 1
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="dw:256:400:784 dw:256:784:400 dw:336:400:784"
for n in 1 2; do
  echo "== GM_DW_TILE48=$n"; GM_DW_TILE48=$n timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
GM_DW_TILE48=2 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "dw or pair" 2>&1 | tail -2
for rep in 1 2 3; do for n in 1 2; do
  GM_DW_TILE48=$n timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile48=$n rep $rep long: %.2f us/step' % (d['ms_per_step']*1e3), d['config']['reps_ms_per_step'], [v for k, v in d['roofline']['per_kernel_us_per_step'].items() if 'pair' in k])"
done; done
for n in 1 2; do for c in wgp_b256 dra_b256; do
  GM_DW_TILE48=$n timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile48=$n $c:', [(round(e['img_s']), round(e['ms_per_step']*1e3, 2)) for e in d])"
done; done
