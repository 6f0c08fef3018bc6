#!/bin/bash
# A/B of two builds of libgm_hip.so on one box: generative_models_amd/ab_libs/{base,<name>}.so
# (built beforehand; *.so is git-ignored but travels with gpurun).  usage: lib_ab.sh <name> [bench args]
NAME=$1; shift
L=generative_models_amd
cp $L/libgm_hip.so /tmp/orig.so
for i in 1 2 3; do for v in base $NAME; do
  cp $L/ab_libs/$v.so $L/libgm_hip.so
  python bench.py --no-configs --no-cpu-baseline --steps 1000 --reps 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,2), list(d['roofline']['per_kernel_us_per_step'].values()))"
done; done
cp /tmp/orig.so $L/libgm_hip.so
