#!/bin/bash
# fixed-width MMD kernel: unit tests, BIR-VAE parity tests, timing
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_ops.py -q -m gpu -x -k "bir or std" 2>&1 | tail -3
timeout 900 python -m pytest tests -q -m gpu -x -k "bir" 2>&1 | tail -3
timeout 300 python tools/variant_times.py bir,vae,dra 2 2>&1 | grep -v amdgpu
