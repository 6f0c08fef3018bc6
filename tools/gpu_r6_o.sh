#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_o
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x > gpurun_out/r06_o/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r06_o/pytest_ops.txt | cut -c1-300
SH="fwd:512:20:400 fwd:256:20:400 fwd:1024:20:400 fwd:2048:20:400 fwd:512:32:400 fwd:512:16:400"
for i in 1 2; do
echo "== k32 kernel"; timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-70
echo "== 16-wave kernel"; GM_TMP_K32_OFF=1 timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-70
done | tee gpurun_out/r06_o/shapes.txt
for i in 1 2 3; do for off in "" 1; do
if [ -n "$off" ]; then export GM_TMP_K32_OFF=1; else unset GM_TMP_K32_OFF; fi
timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']['per_kernel_us_per_step']; print('off=$off', 'step us %.2f'%(d['ms_per_step']*1e3), {k[:24]:v for k,v in r.items() if 'gather' in k or 'k32' in k})"
done; done | tee gpurun_out/r06_o/step_ab.txt
unset GM_TMP_K32_OFF
for off in "" 1 "" 1; do
if [ -n "$off" ]; then export GM_TMP_K32_OFF=1; else unset GM_TMP_K32_OFF; fi
timeout 300 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); d=d[0] if isinstance(d,list) else d; print('off=$off ns_b1024', d.get('ms_per_step'))"
done | tee gpurun_out/r06_o/b1024_ab.txt
