#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_i
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -q -m gpu -x > gpurun_out/r06_i/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/r06_i/pytest_ops.txt | cut -c1-200
SH="fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784 dw:512:784:400 dwadam:512:784:400 dw:256:400:784"
for i in 1 2; do
echo "== built tree"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-50
done | tee gpurun_out/r06_i/sl.txt
for i in 1 2; do timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('step us', d['ms_per_step']*1e3, d['roofline']['per_kernel_us_per_step'])"; done | tee gpurun_out/r06_i/step.txt
