#!/bin/bash
# LDS-staged weight gradient: correctness (asserted inside the tool) + timing next to the split-reduction
# kernel and the vendor GEMM, per reduction length
SH="dw:2048:784:400 dw:2048:400:784 dw:1024:784:400 dw:1024:400:784 dw:1100:100:72 dw:768:784:400 dw:512:784:400 dw:512:400:784 dw:256:400:784 dw:336:784:400"
echo "== split-reduction (GM_LDS_DW_MIN_K=1000000)"; GM_LDS_DW_MIN_K=1000000 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu
echo "== LDS dW for every reduction length (GM_LDS_DW_MIN_K=64)"; GM_LDS_DW_MIN_K=64 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu
echo "== LDS dW, 64x64 tiles forced"; GM_LDS_CFG=1 GM_LDS_DW_MIN_K=64 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | grep -v amdgpu
