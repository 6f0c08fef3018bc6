#!/bin/bash
# LDS macro-tile GEMM: correctness + timing per tile config vs the split-reduction kernels.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
BIG="fwd:2048:784:400 fwd:2048:400:784 fwd:1024:784:400 dx:1024:784:400 dx:1024:400:784"
SMALL="fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784"
echo "== split-reduction kernels (round 1)"; GM_LDS_MIN_M=1000000 timeout 120 python tools/gemm_shapes_bench.py $BIG $SMALL 2>&1 | grep -v amdgpu.ids | tail -12
for cfg in ${CFGS:-1 2 3 4 5 6 7 8}; do
  echo "== LDS cfg $cfg"; GM_LDS_MIN_M=256 GM_LDS_CFG=$cfg timeout 120 python tools/gemm_shapes_bench.py $BIG $SMALL 2>&1 | grep -v amdgpu.ids | tail -12
done
