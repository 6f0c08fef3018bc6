#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/ab
{ timeout 100 python tools/trainer_pipeline_probe.py; GM_GRAPH_ITERS=32 GM_RING=128 timeout 100 python tools/trainer_pipeline_probe.py; } 2>&1 | grep -v amdgpu | tee gpurun_out/ab/pipeline_probe.txt | cut -c1-900
