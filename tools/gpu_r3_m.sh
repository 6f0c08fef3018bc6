#!/bin/bash
# round 3, call M: library variants inside the real step (alternating, same box)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
L=$R/generative_models_amd/ab_libs
for rep in 1 2 3; do for v in default xdx xdirect; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  GM_LIB_PATH=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep: %.2f us/step' % (d['ms_per_step']*1e3), {k.split('<')[0][7:]+'<'+k.split('<')[1][:22]: v for k, v in d['roofline']['per_kernel_us_per_step'].items()})"
done; done
for v in default xdx xdirect; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  for c in vae_b512 ns_b1024 wgp_b256; do
  GM_LIB_PATH=$lib timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $c:', [(round(e['img_s']), round(e['ms_per_step']*1e3, 2)) for e in d])"
  done
done
