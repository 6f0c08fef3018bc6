#!/bin/bash
# WGAN-GP: the penalty's w2 share summed by the head workgroups (13 launches): tests, then off / on alternating
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_ops.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -m gpu -x -k "wgp" 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do
for k in 0 1; do
echo "== GM_WGP_PEN_IN_HEAD=$k"; GM_WGP_PEN_IN_HEAD=$k timeout 300 python bench.py --only wgp_b256 --steps 400 --warmup 50 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([ (e['workload'][:28], round(e['ms_per_step']*1e3,2)) for e in d])"
done; done
