#!/bin/bash
# Round 5, call X: iterations per graph / ring size with the 22 us launches (long runs, same-call alternation)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/x
for rep in 1 2 3; do for cfg in "32 128" "64 256" "128 512" "32 256"; do
  set -- $cfg
  echo "GM_GRAPH_ITERS=$1 GM_RING=$2: $(GM_GRAPH_ITERS=$1 GM_RING=$2 timeout 200 python bench.py --steps 4096 --warmup 512 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("reps_ms_per_step"))')"
done; done 2>&1 | tee gpurun_out/x/graph_iters_ab.txt
GM_GRAPH_ITERS=64 GM_RING=256 timeout 100 python tools/trainer_epoch_ab.py 2>&1 | grep -v amdgpu | head -4
timeout 100 python tools/trainer_epoch_ab.py 2>&1 | grep -v amdgpu | head -4
