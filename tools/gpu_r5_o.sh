#!/bin/bash
# Round 5, call O: the push exchange -- direct exchange tests (worlds 1 / 2 / 4, three forms), N-rank training with it, one-rank structure cost
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dp.py -q -k "peer_allreduce or one_kernel_exchange" > gpurun_out/o_tests.log 2>&1; echo "dp tests rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/o_tests.log | cut -c1-250 | tail
GM_DP_PUSH=1 GM_FORCE_DP=1 timeout 200 python bench.py --steps 512 --warmup 64 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("one-rank push structure:", round(d["ms_per_step"]*1e3,2), d["config"]["gradient_exchange"])'
GM_FORCE_DP=1 timeout 200 python bench.py --steps 512 --warmup 64 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("one-rank pull structure:", round(d["ms_per_step"]*1e3,2), d["config"]["gradient_exchange"])'
for n in 2; do
  for push in 0 1; do
  GM_DP_PUSH=$push GM_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('dry n=$n push=$push', d['config'].get('gradient_exchange'), round(d['ms_per_step']*1e3,1),'us')"
  done
done
