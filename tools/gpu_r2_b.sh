#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/b_pytest.log 2>&1; tail -3 gpurun_out/b_pytest.log
for i in 1 2; do
GM_TRACE_RUN=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/b_bench_short$i.json 2> gpurun_out/b_bench_short$i.err; echo "short bench rc=$?"; cut -c1-800 gpurun_out/b_bench_short$i.json; grep trace gpurun_out/b_bench_short$i.err | cut -c1-900
done
timeout 300 python bench.py --no-configs --no-cpu-baseline > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; echo "bench rc=$?"; cut -c1-800 gpurun_out/b_bench.json
