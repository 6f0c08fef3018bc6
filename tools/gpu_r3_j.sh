#!/bin/bash
# round 3, call J: where the time of a split-reduction launch goes (ablations of the 16x16x4 body, isolated launches)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
SH="dw:2048:784:400 dw:512:784:400 fwd:512:784:400 dx:256:784:400"
for a in 0 1 2 3; do
  echo "== GM_ABLATE16=$a  (0 full, 1 no MFMA, 2 no operand loads, 3 no reduction/epilogue)"
  GM_ABLATE16=$a timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done
