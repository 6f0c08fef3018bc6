#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_n
for t in -1 1 2 4 5 -1 1; do
echo "== GM_TMP_TILE=$t (0=32x32 1=32x64 2=64x32 3=16x32 4=32x48 5=48x32)"
GM_TMP_TILE=$t timeout 120 python tools/gemm_shapes_bench.py fwdsig:512:400:784 2>&1 | grep -v amdgpu.ids | cut -c1-70
GM_TMP_TILE=$t timeout 300 python bench.py --steps 512 --warmup 64 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('step us %.2f'%(d['ms_per_step']*1e3))"
done | tee gpurun_out/r06_n/tile_g2.txt
