#!/bin/bash
# Round 5, call I: on a fresh box first the 20-step A/B of the pre-stage event tracking and of 64-iteration graphs, then
# the round's profiles (rocprofv3 stats + PMC passes), then the two committed bench lines with the fresh profiles in place.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
run20() { echo "$*: $(env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"; }
for rep in 1 2 3; do
  run20 GM_TRACK_PRESTAGE=1
  run20 GM_TRACK_PRESTAGE=0
done
for rep in 1 2; do
  echo "long default: $(timeout 200 python bench.py --steps 2000 --warmup 200 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"]["reps_ms_per_step"])')"
  echo "long graph_iters=64 ring=256: $(GM_GRAPH_ITERS=64 GM_RING=256 timeout 200 python bench.py --steps 2000 --warmup 200 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"]["reps_ms_per_step"])')"
done
bash tools/gpu_r5_profiles.sh all > gpurun_out/i_profiles.log 2>&1; tail -25 gpurun_out/i_profiles.log
cp gpurun_out/profiles_r05/r05_* profiles/ 2>/dev/null
bash tools/gpu_r5_final.sh
