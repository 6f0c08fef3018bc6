#!/bin/bash
# round 3, call K: x-contiguous operands as direct dword fragments (GM_XDIRECT=1) vs 16-byte loads + quad transposes
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="dw:2048:784:400 dw:1024:784:400 dw:512:784:400 dw:256:784:400 dw:512:400:784 dw:256:400:784 dw:256:20:400 dx:256:784:400 dx:512:784:400 dx:256:400:784 dx:512:400:784"
for x in 0 1; do
  echo "== GM_XDIRECT=$x"; GM_XDIRECT=$x timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-118
done
echo "== GM_XDIRECT=1, no operand loads (GM_ABLATE16=2)"; GM_XDIRECT=1 GM_ABLATE16=2 timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
echo "== GM_XDIRECT=1, no MFMA (GM_ABLATE16=1)"; GM_XDIRECT=1 GM_ABLATE16=1 timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 2>&1 | grep -v amdgpu | cut -c1-60
