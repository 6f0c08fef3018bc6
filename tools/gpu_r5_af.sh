#!/bin/bash
# Round 5, call AF: the N = 2 bench flow on one device (dry run of the code path) on the final tree
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/final_r05
GM_BENCH_ONE_DEVICE=1 timeout 90 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/final_r05/r05_bench_dry_n2.json 2> gpurun_out/final_r05/dry_n2.err; echo "dry n=2 rc=$?"
tail -c 1200 gpurun_out/final_r05/r05_bench_dry_n2.json | cut -c1-1200; tail -3 gpurun_out/final_r05/dry_n2.err | cut -c1-300
