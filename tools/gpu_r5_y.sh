#!/bin/bash
# Round 5, call Y: automatic ring / graph size (512 slots, 128 iterations per graph where the draws afford it) against 128 / 32:
# the whole GPU suite, then the configs either way
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/y
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/y/tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/y/tests.log | cut -c1-250
for rep in 1 2; do for cfg in "auto" "32 128"; do
  if [ "$cfg" = "auto" ]; then E=""; else set -- $cfg; E="GM_GRAPH_ITERS=$1 GM_RING=$2"; fi
  echo "[$cfg] nsgan long: $(env $E timeout 200 python bench.py --steps 4096 --warmup 512 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("reps_ms_per_step"))')"
  echo "[$cfg] nsgan 20 steps: $(env $E timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("steady_us_per_step"), d["config"].get("run_fixed_cost_us"))')"
  for c in ns_b1024 wgp_b256 dra_b256; do
    echo "[$cfg] $c: $(env $E timeout 200 python bench.py --only $c --steps 600 --warmup 60 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1])[0]; print(round(d["ms_per_step"]*1e3,2), d["reps_ms_per_step"])')"
  done
  env $E timeout 100 python tools/trainer_epoch_ab.py 2>&1 | grep -v amdgpu | head -2
done; done 2>&1 | tee gpurun_out/y/auto_ring_ab.txt
