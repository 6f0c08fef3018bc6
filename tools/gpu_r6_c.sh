#!/bin/bash
# Round 6 call C: the captured general path (README override): tests + timing + a kernel trace of it
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_c
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -m gpu -k "override or captured" > gpurun_out/r06_c/pytest.txt 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r06_c/pytest.txt | cut -c1-300
timeout 600 python bench.py --general-path > gpurun_out/r06_c/general_path.json 2> gpurun_out/r06_c/general_path.err; echo "bench rc=$?"; cat gpurun_out/r06_c/general_path.json; tail -5 gpurun_out/r06_c/general_path.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_c/prof -o gp -- python $R/bench.py --general-path > /dev/null 2>&1; echo "prof rc=$?"
S=$(find $R/gpurun_out/r06_c/prof -name "*kernel_stats.csv" | head -1); head -40 "$S"
find $R/gpurun_out/r06_c/prof -name "*kernel_trace.csv" -delete
