#!/bin/bash
# Round 4 (VERDICT r3 "missing" 4): is the bench's sustained window visible to an outside sampler of GPU activity?
# rocm-smi --showuse once a second beside `python bench.py --no-cpu-baseline --no-configs`
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
( for i in $(seq 1 40); do echo "t=$i $(rocm-smi --showuse 2>/dev/null | grep -i "GPU use" | head -1)"; sleep 1; done ) > gpurun_out/r04_gpu_busy_samples.txt &
S=$!
timeout 300 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("sustained", d["config"]["sustained_window"])'
kill $S 2>/dev/null; wait $S 2>/dev/null
cat gpurun_out/r04_gpu_busy_samples.txt | tr '\n' ';' | cut -c1-1500
