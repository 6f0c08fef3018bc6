#!/bin/bash
# Round 5, call S: timeline of Trainer.train() epochs (why 71 us per step against 68.9 steady)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/s
timeout 300 python tools/trainer_trace_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s/trainer_trace.txt | cut -c1-1500
