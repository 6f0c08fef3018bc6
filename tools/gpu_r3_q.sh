#!/bin/bash
# round 3, call Q: 32x48 tiles for the wide weight gradients (GM_DW_NI3=1): correctness (the shapes tool asserts
# against torch), isolated timing, then inside the real steps
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="dw:2048:784:400 dw:1024:784:400 dw:512:784:400 dw:768:784:400 dw:2048:400:784 dw:512:400:784"
for n in 0 1 0 1; do
  echo "== GM_DW_NI3=$n"; GM_DW_NI3=$n timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
timeout 600 env GM_DW_NI3=1 python -m pytest tests/test_gpu_ops.py -x -q -k "dw or pair or head" 2>&1 | tail -2
for rep in 1 2; do for n in 0 1; do
  GM_DW_NI3=$n timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ni3=$n rep $rep long: %.2f us/step' % (d['ms_per_step']*1e3), d['config']['reps_ms_per_step'])"
  for c in ns_b1024 vae_b512; do
  GM_DW_NI3=$n timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ni3=$n $c:', [(round(e['img_s']), round(e['ms_per_step']*1e3, 2)) for e in d])"
  done
done; done
