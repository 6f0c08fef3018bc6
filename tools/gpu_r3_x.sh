#!/bin/bash
# DRAGAN bs=256: rocprofv3 kernel stats + idle gaps (where do its 185 us go?)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/profiles_r03; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles_r03
cd /tmp
tag=r03_dra_b256
timeout 300 python $R/bench.py --only dra_b256 --steps 400 --warmup 50 --reps 3 > $OUT/${tag}_bench_plain.json 2> $R/gpurun_out/dra_plain.log; echo "plain rc=$?"
cat $OUT/${tag}_bench_plain.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_$tag -o ns -- python $R/bench.py --only dra_b256 --steps 200 --warmup 20 --reps 1 > $OUT/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/pr_$tag.log; echo "stats rc=$?"
python $R/profiles/make_summary.py $R/gpurun_out/pr_$tag $tag $OUT > /dev/null
T=$(find $R/gpurun_out/pr_$tag -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $T > $OUT/${tag}_gaps.txt 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -delete
cat $OUT/${tag}_summary.md | head -60
