#!/bin/bash
# round 3, call A: DP tests at real shapes, whole GPU suite, headline bench, N=2 dry run on one device
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_dp.py -x -q -k "full_size or strided" ) > gpurun_out/r3a/dp_full.log 2>&1
tail -5 gpurun_out/r3a/dp_full.log
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dp.py::test_n_rank_engine_equals_one_rank_full_size ) > gpurun_out/r3a/gpu_all.log 2>&1
tail -5 gpurun_out/r3a/gpu_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_20.json 2> gpurun_out/r3a/bench_20.err
tail -c 600 gpurun_out/r3a/bench_20.json
GM_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r3a/bench_dry_n2.json 2> gpurun_out/r3a/bench_dry_n2.err
tail -c 1500 gpurun_out/r3a/bench_dry_n2.json
