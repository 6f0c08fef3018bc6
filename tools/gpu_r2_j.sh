#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dp.py > gpurun_out/j_pytest.log 2>&1; tail -4 gpurun_out/j_pytest.log
for p in 1 0; do GM_PACKED=$p timeout 300 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | cut -c1-250; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | cut -c1-250
