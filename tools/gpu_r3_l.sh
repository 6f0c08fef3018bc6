#!/bin/bash
# round 3, call L: direct dword fragments as a compile-time variant of the library (GM_LIB_PATH), isolated launches
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
L=$R/generative_models_amd/ab_libs
SH="dw:2048:784:400 dw:1024:784:400 dw:512:784:400 dw:256:784:400 dw:512:400:784 dw:256:400:784 dx:256:784:400 dx:512:784:400 dx:256:400:784 dx:1024:400:784"
for rep in 1 2; do
echo "== default build (rep $rep)"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-118
echo "== xdirect build (rep $rep)"; GM_LIB_PATH=$L/xdirect.so timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-118
done
echo "== default, no operand loads"; GM_ABLATED_LIB=1 GM_LIB_PATH=$L/base_noloads.so timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
echo "== xdirect, no operand loads"; GM_ABLATED_LIB=1 GM_LIB_PATH=$L/xd_noloads.so timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
echo "== xdirect, no MFMA"; GM_ABLATED_LIB=1 GM_LIB_PATH=$L/xd_nomfma.so timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
