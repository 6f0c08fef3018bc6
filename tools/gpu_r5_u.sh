#!/bin/bash
# Round 5, call U: the gate wait as its own one-wave kernel in front of an ungated in-graph copy (GM_PRESTAGE=0) against the
# side-stream pre-stage (default) and the old per-workgroup poll (GM_STAGE_SPLIT_GATE=0)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/u
timeout 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "stage" > gpurun_out/u/tests.log 2>&1; echo "stage tests rc=$?"; tail -2 gpurun_out/u/tests.log | cut -c1-200
{
for rep in 1 2; do
timeout 120 python tools/piece_cost_probe.py 4,8,16,32,4
GM_PRESTAGE=0 timeout 120 python tools/piece_cost_probe.py 4,8,16,32,4
GM_PRESTAGE=0 GM_STAGE_SPLIT_GATE=0 timeout 120 python tools/piece_cost_probe.py 4,8,16,32,4
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/u/piece_cost.txt | cut -c1-400
for rep in 1 2 3; do for pre in 1 0; do
  echo "GM_PRESTAGE=$pre long: $(GM_PRESTAGE=$pre timeout 200 python bench.py --steps 2000 --warmup 200 --reps 3 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("reps_ms_per_step"))')"
  echo "GM_PRESTAGE=$pre 20 steps: $(GM_PRESTAGE=$pre timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("steady_us_per_step"), d["config"].get("run_fixed_cost_us"))')"
done; done 2>&1 | tee gpurun_out/u/bench_ab.txt
