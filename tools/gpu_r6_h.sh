#!/bin/bash
# Round 6 call H: instruction-cache counters of the NSGAN bs=256 step (is a launch's entry phase instruction fetch?)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_h; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -iE "ICACHE|IFETCH|SQ_INST_LEVEL|SQC_" | head -40 > $R/gpurun_out/r06_h/avail.txt; cat $R/gpurun_out/r06_h/avail.txt | cut -c1-160
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d $R/gpurun_out/r06_h/ic -o ns -- python $R/bench.py --steps 200 --warmup 50 --reps 1 --no-cpu-baseline --no-configs --sustained 0 > $R/gpurun_out/r06_h/ic.log 2>&1; echo "pmc rc=$?"
python $R/tools/pmc_sq_summary.py $R/gpurun_out/r06_h/ic $R/gpurun_out/r06_h/icache.json > $R/gpurun_out/r06_h/icache.txt 2>&1; head -120 $R/gpurun_out/r06_h/icache.txt
find $R/gpurun_out/r06_h -name "*counter_collection.csv" -size +20M -delete; find $R/gpurun_out/r06_h -name "*kernel_trace.csv" -delete
