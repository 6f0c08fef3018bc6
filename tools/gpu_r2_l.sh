#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_l -o ns -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-configs --reps 1 > $R/gpurun_out/l_under_rocprof.json 2> $R/gpurun_out/prof_l.log; echo "prof rc=$?"
T=$(find $R/gpurun_out/prof_l -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $T | tee $R/gpurun_out/l_gaps.txt
find $R/gpurun_out -name "*kernel_trace.csv" -size +20M -delete
