#!/bin/bash
# Round-2 first GPU pass: GPU tests, driver-style short bench, default bench.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/a_pytest.log 2>&1; tail -5 gpurun_out/a_pytest.log
for i in 1 2; do
GM_BENCH_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/a_bench_short$i.json 2> gpurun_out/a_bench_short$i.err; echo "short bench rc=$?"; cut -c1-900 gpurun_out/a_bench_short$i.json
done
GM_HOST_REPLAY=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/a_bench_short_noreplay.json 2>/dev/null; cut -c1-400 gpurun_out/a_bench_short_noreplay.json
GM_BENCH_VERBOSE=1 timeout 900 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"; cat gpurun_out/a_bench.json; tail -30 gpurun_out/a_bench.err
