#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_k
SH="fwd:2048:800:400 fwd:2048:1600:400 fwd:2048:3200:400 fwd:2048:6400:400 fwd:2048:800:784 fwd:2048:1600:784 fwd:2048:3200:784 dx:2048:800:784 dx:2048:3200:784"
for i in 1; do
echo "== lds16 by rule"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== lds16 off"; GM_TMP_LDS16_OFF=1 timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-120
done | tee gpurun_out/r06_k/kscale.txt
