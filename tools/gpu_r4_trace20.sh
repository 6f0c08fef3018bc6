#!/bin/bash
# kernel-trace timeline of 20-step timed runs (stage-in launches, graph boundaries), GM_PRESTAGE on / off
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $R/gpurun_out/tr20
  GM_PRESTAGE=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr20 -o ns -- python $R/bench.py --steps 20 --warmup 5 --reps 5 --no-cpu-baseline --no-configs --sustained 0 > /dev/null 2>&1
  T=$(find $R/gpurun_out/tr20 -name "*kernel_trace.csv" | head -1)
  echo "== GM_PRESTAGE=$v"; python $R/tools/trace_runs.py $T --idle 60 | grep -v "largest\|first tenth" | cut -c1-160 | head -60
done
rm -rf $R/gpurun_out/tr20
