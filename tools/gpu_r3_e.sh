#!/bin/bash
# round 3, call E: in-situ kernel times of the headline step, folded vs unfolded head (rocprofv3 --kernel-trace --stats)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/r3e; export TMPDIR=/tmp
OUT=$R/gpurun_out/r3e
cd /tmp
for f in 1 0; do
  tag=nsgan_b256_fold$f
  GM_FOLD_HEAD=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_$tag -o ns -- python $R/bench.py --steps 400 --warmup 50 --reps 1 --no-cpu-baseline --no-configs > $OUT/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/pr_$tag.log; echo "$tag stats rc=$?"
  python $R/profiles/make_summary.py $R/gpurun_out/pr_$tag $tag $OUT > /dev/null
  T=$(find $R/gpurun_out/pr_$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_gaps.py $T > $OUT/${tag}_gaps.txt 2>&1
  head -24 $OUT/${tag}_summary.md
  head -14 $OUT/${tag}_gaps.txt
  find $R/gpurun_out -name "*kernel_trace.csv" -delete
done
