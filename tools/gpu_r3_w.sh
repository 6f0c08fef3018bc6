#!/bin/bash
# round 3, call W: interior chunks of interior tiles skip the bounds selects (-DGM_FAST_INTERIOR=1)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
L=$R/generative_models_amd/ab_libs
SH="dw:2048:784:400 dw:512:784:400 dw:256:400:784 fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784 fwd:100:65:31 dw:100:64:48 dx:33:65:31"
for v in default fast default fast; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  echo "== $v"; GM_LIB_PATH=$lib timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
GM_LIB_PATH=$L/fast.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do for v in default fast; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  GM_LIB_PATH=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep long: %.2f us/step' % (d['ms_per_step']*1e3), d['config']['reps_ms_per_step'], list(d['roofline']['per_kernel_us_per_step'].values()))"
done; done
for v in default fast; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  for c in ns_b1024 vae_b512; do
  GM_LIB_PATH=$lib timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $c:', [(round(e['img_s']), round(e['ms_per_step']*1e3, 2)) for e in d])"
  done
done
