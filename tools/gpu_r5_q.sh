#!/bin/bash
# Round 5, call Q: bit-packed rows as a GEMM operand (GM_PACKED_OPERAND=1): parity tests, the
# NSGAN bs=256 step either way (same-call alternation), per-kernel durations and HBM write bytes either way.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/q; export TMPDIR=/tmp
O=$R/gpurun_out/q
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_trainers.py -q -p no:cacheprovider -k "packed or folded" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-220
for rep in 1 2 3; do for pk in 0 1; do
  echo "GM_PACKED_OPERAND=$pk: $(GM_PACKED_OPERAND=$pk timeout 200 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>$O/bench_err_$pk.log | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"].get("reps_ms_per_step"), d["config"].get("steady_us_per_step"))')"
done; done
cd /tmp
for pk in 0 1; do
  GM_PACKED_OPERAND=$pk timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$pk -o ns -- python $R/bench.py --steps 400 --warmup 50 --reps 1 --no-cpu-baseline --no-configs --sustained 0 > $O/kt_$pk.json 2> $O/kt_$pk.log; echo "kernel trace pk=$pk rc=$?"
  python $R/profiles/make_summary.py $O/kt_$pk r05_nsgan_b256_packed_operand_$pk $O > /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    GM_PACKED_OPERAND=$pk timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${pk}_$c -o ns -- python $R/bench.py --steps 400 --warmup 50 --reps 1 --no-cpu-baseline --no-configs --sustained 0 > $O/pmc_${pk}_$c.log 2>&1; echo "pmc pk=$pk $c rc=$?"
  done
  python $R/profiles/make_pmc_summary.py $O/pmc_${pk}_ r05_nsgan_b256_packed_operand_$pk $O "NSGAN bs=256, GM_PACKED_OPERAND=$pk, bench.py --steps 400 --warmup 50" > /dev/null
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls $O; du -sh $O
