#!/bin/bash
# BIR-VAE with the candidate stage of the numpy replay on AVX-512: parity tests, epoch-loop timing
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "bir" 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 200 python -m pytest tests/test_host_replay.py -q 2>&1 | tail -1
echo -n "scalar candidates: "; GM_NUMPY_SCALAR=1 timeout 200 python tools/variant_times.py bir 2 2>&1 | grep -v amdgpu
echo -n "avx-512 candidates: "; timeout 200 python tools/variant_times.py bir 2 2>&1 | grep -v amdgpu
echo -n "avx-512 candidates: "; timeout 200 python tools/variant_times.py bir,vae 2 2>&1 | grep -v amdgpu
