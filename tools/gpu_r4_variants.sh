#!/bin/bash
# round 4: kernel stats of the variants that are not BASELINE configurations (ra, fisher, be, info), through their drop-in trainers
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/profiles_r04; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles_r04
cd /tmp
for v in ${1:-ra fisher be info}; do E=${2:-1}
  tag=r04_${v}_b256
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_$tag -o ns -- python $R/tools/variant_times.py $v $E > $OUT/${tag}_variant_times.txt 2> $R/gpurun_out/pr_$tag.log; echo "$v rc=$?"
  python $R/profiles/make_summary.py $R/gpurun_out/pr_$tag $tag $OUT > /dev/null
  T=$(find $R/gpurun_out/pr_$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_gaps.py $T > $OUT/${tag}_gaps.txt 2>&1
  find $R/gpurun_out -name "*kernel_trace.csv" -delete
  grep -v amdgpu $OUT/${tag}_variant_times.txt
  sed -n 1,5p $OUT/${tag}_summary.md; head -3 $OUT/${tag}_gaps.txt
done
