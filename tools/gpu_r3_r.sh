#!/bin/bash
# round 3, call R: 48-wide / 48-tall tiles for the weight gradients, default on (GM_DW_TILE48): correctness + A/B
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="dw:2048:784:400 dw:512:784:400 dw:768:784:400 dw:2048:400:784 dw:1024:400:784 dw:512:400:784 dw:336:784:400 dw:100:64:48 dw:513:130:500"
for n in 0 1; do
  echo "== GM_DW_TILE48=$n"; GM_DW_TILE48=$n timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -x -q 2>&1 | tail -2
for rep in 1 2; do for n in 0 1; do
  GM_DW_TILE48=$n timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile48=$n rep $rep long: %.2f us/step' % (d['ms_per_step']*1e3), d['config']['reps_ms_per_step'], d['roofline']['per_kernel_us_per_step'])"
  for c in ns_b1024 vae_b512 wgp_b256; do
  GM_DW_TILE48=$n timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile48=$n $c:', [(round(e['img_s']), round(e['ms_per_step']*1e3, 2)) for e in d])"
  done
done; done
