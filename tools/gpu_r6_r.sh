#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_r; export PYTHONPATH=$R
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x > gpurun_out/r06_r/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/r06_r/pytest_ops.txt | cut -c1-300
SH="fwd:512:784:400 fwd:256:784:400 dx:256:400:784"
for i in 1 2; do
echo "== quartered tail"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-70
echo "== plain"; GM_TMP_QT_OFF=1 timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-70
done | tee gpurun_out/r06_r/shapes.txt
for i in 1 2 3; do for off in "" 1; do
if [ -n "$off" ]; then export GM_TMP_QT_OFF=1; else unset GM_TMP_QT_OFF; fi
timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('off=$off', 'step us %.2f'%(d['ms_per_step']*1e3))"
done; done | tee gpurun_out/r06_r/step_ab.txt
