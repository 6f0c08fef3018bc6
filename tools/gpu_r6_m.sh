#!/bin/bash
# Round 6 closing call: timelines on the final kernels, all profiles, the committed bench lines
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_m
timeout 300 python tools/wave_timeline.py --variant ns --batch 256 --iters 2 --out gpurun_out/r06_m/r06_ns_b256_wave_timeline.md > gpurun_out/r06_m/tl256.log 2>&1; echo "tl256 rc=$?"
timeout 300 python tools/wave_timeline.py --variant ns --batch 1024 --iters 2 --out gpurun_out/r06_m/r06_ns_b1024_wave_timeline.md > gpurun_out/r06_m/tl1024.log 2>&1; echo "tl1024 rc=$?"
bash tools/gpu_r6_profiles.sh all > gpurun_out/r06_m/profiles.log 2>&1; echo "profiles rc=$?"
bash tools/gpu_r6_profiles.sh sq >> gpurun_out/r06_m/profiles.log 2>&1
bash tools/gpu_r6_profiles.sh variants >> gpurun_out/r06_m/profiles.log 2>&1
bash tools/gpu_r6_final.sh
