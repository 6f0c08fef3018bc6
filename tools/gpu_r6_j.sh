#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_j
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -q -m gpu -x > gpurun_out/r06_j/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/r06_j/pytest_ops.txt | cut -c1-200
for i in 1 2; do timeout 300 python tools/gemm_shapes_bench.py dw:2048:784:400 dwadam:2048:784:400 dw:1024:784:400 dw:1024:400:784 dw:768:784:400 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee gpurun_out/r06_j/dw.txt
for c in ns_b1024 ls_b1024; do timeout 300 python bench.py --only $c --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); d=d[0] if isinstance(d,list) else d; print(d.get('workload','')[:40], d.get('ms_per_step'), d.get('roofline',{}).get('per_kernel_us_per_step'))"; done | tee gpurun_out/r06_j/configs.txt
