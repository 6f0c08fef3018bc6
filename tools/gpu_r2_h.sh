#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_trainers.py -m gpu -q -x -k "dp or peer or rank" 2>&1 | tail -8
GM_FORCE_DP=1 timeout 300 python bench.py --no-configs --no-cpu-baseline > gpurun_out/h_bench_dp1.json 2> gpurun_out/h_bench_dp1.err; echo "force-dp peer rc=$?"; cut -c1-300 gpurun_out/h_bench_dp1.json; tail -3 gpurun_out/h_bench_dp1.err
