#!/bin/bash
# round 3, call O: MFMA order -- k-steps outermost (independent consecutive MFMAs) vs accumulator by accumulator
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
L=$R/generative_models_amd/ab_libs
SH="dw:2048:784:400 dw:512:784:400 dw:256:784:400 dw:512:400:784 fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784"
for v in default il ilxd default il; do
  lib=""; [ $v != default ] && lib=$L/$v.so
  echo "== $v"; GM_LIB_PATH=$lib timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
echo "== il, no operand loads"; GM_ABLATED_LIB=1 GM_LIB_PATH=$L/il_noloads.so timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 fwd:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
echo "== default, no operand loads"; GM_ABLATED_LIB=1 GM_LIB_PATH=$L/base_noloads.so timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 fwd:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
