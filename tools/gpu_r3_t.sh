#!/bin/bash
# round 3, call T: leftover reduction chunks spread by k-step over the waves' last round (K = 784: 49 chunks / 16 waves)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
L=$R/generative_models_amd/ab_libs
SH="fwd:512:784:400 fwd:256:784:400 dx:256:400:784 fwd:512:400:784 dw:784:400:784 fwd:100:65:31 dx:64:48:49"
for v in default spread default spread; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  echo "== $v"; GM_LIB_PATH=$lib timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
GM_LIB_PATH=$L/spread.so timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do for v in default spread; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  GM_LIB_PATH=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep long: %.2f us/step' % (d['ms_per_step']*1e3), d['config']['reps_ms_per_step'], d['roofline']['per_kernel_us_per_step'])"
done; done
