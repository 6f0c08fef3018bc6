#!/bin/bash
# Round 5, call J: 20-step figure with the pooled pre-stage events (two alternations against tracking off is gone: the
# switch no longer exists -- compare with call I's 75.4 - 77.0 untracked), the two committed bench lines, the one-device
# dry runs of the N = 2 / 4 bench flow, smoke.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out/final_r05; export TMPDIR=/tmp
run20() { echo "$*: $(env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"; }
for rep in 1 2 3; do run20 X=1; done
bash tools/gpu_r5_final.sh
for n in 2 4; do
  GM_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/final_r05/r05_bench_dry_n$n.json 2> gpurun_out/final_r05/dry_n$n.err; echo "dry n=$n rc=$?"
  python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/final_r05/r05_bench_dry_n$n.json').read().strip().split('\n')[-1]); print('  ranks_seen', d['config'].get('ranks_seen'), d['config'].get('gradient_exchange'), round(d['ms_per_step']*1e3,1),'us', d['config'].get('ranks'))
except Exception as e: print('  unreadable', e)"
done
GM_DP_COMM=rccl GM_FORCE_DP=1 timeout 200 python bench.py --steps 512 --warmup 64 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("one-rank rccl-in-graph structure:", round(d["ms_per_step"]*1e3,2), d["config"]["gradient_exchange"])'
GM_FORCE_DP=1 timeout 200 python bench.py --steps 512 --warmup 64 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("one-rank peer structure:", round(d["ms_per_step"]*1e3,2), d["config"]["gradient_exchange"])'
GM_DP_COMM=rccl GM_RCCL_IN_GRAPH=0 GM_FORCE_DP=1 timeout 200 python bench.py --steps 512 --warmup 64 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("one-rank host-launched rccl structure:", round(d["ms_per_step"]*1e3,2), d["config"]["gradient_exchange"])'
