#!/bin/bash
# Round-end measurement on the GPU box: full GPU test suite, smoke, bench line (with cpu_baseline),
# rocprofv3 kernel-trace stats of the bench command, and the two PMC passes (each in its own run).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/final_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/final_pytest.log | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/final_smoke.log
GM_BENCH_VERBOSE=1 timeout 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; cat gpurun_out/bench_final.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o ns -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_final.log; echo "prof rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcF_$c -o ns -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline > $R/gpurun_out/pmcF_$c.log 2>&1; echo "pmc $c rc=$?"
done
find $R/gpurun_out/prof_final $R/gpurun_out/pmcF_FETCH_SIZE -type f | head -20
# keep the merged output small: the per-dispatch traces are large
find $R/gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh $R/gpurun_out
