#!/bin/bash
# round 3, call H: LDS-staged weight gradient with fully unrolled stage counts vs the split-reduction kernel vs the vendor GEMM
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/r3h; export TMPDIR=/tmp
SH="dw:2048:784:400 dw:1024:784:400 dw:512:784:400 dw:256:784:400 dw:1024:400:784 dw:512:400:784"
echo "== split-reduction (default)"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | tail -8
for cfg in 1 2; do
  echo "== LDS dW, GM_LDS_CFG=$cfg"; GM_LDS_DW_MIN_K=256 GM_LDS_CFG=$cfg timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | tail -8
done
echo "== LDS dW, runtime stage loop (GM_LDS_UNROLL=0), cfg 2"; GM_LDS_DW_MIN_K=256 GM_LDS_CFG=2 GM_LDS_UNROLL=0 timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | tail -3
