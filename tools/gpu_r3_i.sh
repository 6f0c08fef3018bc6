#!/bin/bash
# round 3, call I: weight gradient on the 32x32x2 MFMA form (8 k-rows x 128 B per load instruction for x-contiguous
# operands) vs the 16x16x4 form (16 k-rows x 64 B)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="dw:2048:784:400 dw:1024:784:400 dw:512:784:400 dw:256:784:400 dw:512:400:784 dx:256:784:400 dx:256:400:784"
echo "== 16x16x4 (default)"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | tail -7
echo "== 32x32x2 (GM_MFMA16=0)"; GM_MFMA16=0 timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | tail -7
echo "== 32x32x2, 8 waves for > 256 tiles (GM_MFMA16=0 GM_WAVES8=1)"; GM_MFMA16=0 GM_WAVES8=1 timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | tail -7
