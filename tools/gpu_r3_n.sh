#!/bin/bash
# round 3, call N: weight gradient with the next chunk's loads issued before the current chunk is consumed
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
L=$R/generative_models_amd/ab_libs
SH="dw:2048:784:400 dw:1024:784:400 dw:512:784:400 dw:256:784:400 dw:1024:400:784 dw:512:400:784"
for v in default pf2 pf4 pf2xd default; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  echo "== $v"; GM_LIB_PATH=$lib timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
