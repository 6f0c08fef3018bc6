#!/bin/bash
# multi-workgroup std kernel: tests, DRAGAN parity tests, DRAGAN bench
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_ops.py -q -m gpu -x -k "std or dragan" 2>&1 | tail -5
timeout 900 python -m pytest tests -q -m gpu -x -k "dra" 2>&1 | tail -5
timeout 300 python bench.py --only dra_b256 --steps 400 --warmup 50 --reps 3 2> gpurun_out/dra_plain.log
