#!/bin/bash
# interleaved dW: double-buffered prefetch on (default lib) / off (variant), isolated shapes, IL forced for all sizes too
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
S="dw:2048:784:400 dw:1024:400:784 dw:1024:784:400 dw:512:784:400 dw:256:400:784"
for rep in 1 2; do
echo "== prefetch on"; timeout 120 python tools/gemm_shapes_bench.py $S 2>&1 | grep -v amdgpu
echo "== prefetch off"; GM_LIB_PATH=$R/generative_models_amd/ab_libs/nopf.so timeout 120 python tools/gemm_shapes_bench.py $S 2>&1 | grep -v amdgpu
done
echo "== prefetch on, IL for every size"; GM_DW_IL_MIN_K=1 timeout 120 python tools/gemm_shapes_bench.py $S dw:336:784:400 dw:37:100:70 2>&1 | grep -v amdgpu
