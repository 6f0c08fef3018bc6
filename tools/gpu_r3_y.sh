#!/bin/bash
# interleaved-fragment weight gradient in the real bs=1024 step: off / on alternating; then isolated shapes
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
for rep in 1 2 3; do
for k in 0 1024; do
echo "== GM_DW_IL_MIN_K=$k"; GM_DW_IL_MIN_K=$k timeout 200 python bench.py --only ns_b1024 --steps 400 --warmup 50 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([ (e['workload'][:24], round(e['ms_per_step']*1e3,2), e.get('reps_ms_per_step')) for e in d])"
done; done
timeout 120 python tools/gemm_shapes_bench.py dw:2048:784:400 dw:1024:400:784 dw:512:784:400 dw:256:400:784 2>&1 | grep -v amdgpu
