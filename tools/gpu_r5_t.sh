#!/bin/bash
# Round 5, call T: the cost of one graph launch beyond its iterations, by launch structure
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/t
{
for rep in 1 2; do
timeout 120 python tools/piece_cost_probe.py 4,8,16,32,4
GM_PRESTAGE=0 timeout 120 python tools/piece_cost_probe.py 4,8,16,32,4
GM_GATED=0 timeout 120 python tools/piece_cost_probe.py 4,8,16,32,4
GM_GRAPH_ITERS=64 GM_RING=256 timeout 120 python tools/piece_cost_probe.py 4,8,16,32,64,4
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t/piece_cost.txt | cut -c1-600
