#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_k
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "linear_fwd or linear_bwd_dx or lds or many" > gpurun_out/r06_k/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r06_k/pytest_ops.txt | cut -c1-300
SH="fwd:2048:784:400 fwd:2048:400:784 fwdsig:2048:400:784 dx:1024:784:400 dx:1024:400:784 fwd:1024:784:400 dx:2048:784:400 dx:2048:400:784 fwd:4096:784:400 fwd:1536:784:400"
for i in 1 2; do
echo "== lds16 by rule"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== lds16 off"; GM_TMP_LDS16_OFF=1 timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-120
done | tee gpurun_out/r06_k/shapes.txt
