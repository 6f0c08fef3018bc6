#!/bin/bash
# BIR-VAE with numpy's legacy Gaussian draws replayed in C: parity tests, then the epoch loop with numpy / 1 / 2 / 4 threads
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "bir" 2>&1 | grep -E "passed|failed|error" | tail -2
echo -n "numpy itself: "; GM_NUMPY_REPLAY=0 timeout 200 python tools/variant_times.py bir 2 2>&1 | grep -v amdgpu
for t in 1 2 4; do echo -n "C replay, $t threads: "; GM_NUMPY_THREADS=$t timeout 200 python tools/variant_times.py bir 2 2>&1 | grep -v amdgpu; done
