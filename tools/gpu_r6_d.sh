#!/bin/bash
# Round 6 call D: per-wave timelines (stamps build) + the new 50-step / epoch fixtures on the HIP path
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_d
timeout 300 python tools/wave_timeline.py --variant ns --batch 256 --iters 2 --out gpurun_out/r06_d/ns_b256_timeline.md > gpurun_out/r06_d/tl256.log 2>&1; echo "tl256 rc=$?"; head -30 gpurun_out/r06_d/ns_b256_timeline.md | cut -c1-200
timeout 300 python tools/wave_timeline.py --variant ns --batch 1024 --iters 2 --out gpurun_out/r06_d/ns_b1024_timeline.md > gpurun_out/r06_d/tl1024.log 2>&1; echo "tl1024 rc=$?"; head -30 gpurun_out/r06_d/ns_b1024_timeline.md | cut -c1-200
rm -f gpurun_out/parity_50step.json
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -m gpu -k "50steps or epoch98" > gpurun_out/r06_d/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06_d/pytest.txt | cut -c1-300
