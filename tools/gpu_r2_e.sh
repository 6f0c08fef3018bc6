#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="fwd:2048:784:400 fwd:2048:400:784 fwd:1024:784:400 dx:1024:784:400 dx:1024:400:784 fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784"
for cfg in ${CFGS:-1 2 3 4 5 6}; do
echo "== cfg $cfg full"; GM_LDS_MIN_M=256 GM_LDS_CFG=$cfg python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu
done
