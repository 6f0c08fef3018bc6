#!/bin/bash
# Round 6 call F: reduction-buffer swizzle: op tests (bit-identical results expected), isolated shapes, step time; fold-prologue stamps
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_f
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -q -m gpu -x > gpurun_out/r06_f/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r06_f/pytest_ops.txt | cut -c1-200
timeout 300 python tools/gemm_shapes_bench.py fwd:512:784:400 fwd:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784 dw:512:784:400 dwadam:512:784:400 dw:256:400:784 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_f/shapes.txt
for i in 1 2; do timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('step us', d['ms_per_step']*1e3, d['roofline']['per_kernel_us_per_step'])"; done | tee gpurun_out/r06_f/step.txt
timeout 300 python tools/wave_timeline.py --variant ns --batch 256 --iters 2 --out gpurun_out/r06_f/ns_b256_timeline.md > /dev/null 2>&1; echo "tl rc=$?"
