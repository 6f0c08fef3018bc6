#!/bin/bash
# Fill-law measurement: tools/fill_probe.bin alone (timing) and under rocprofv3 --pmc passes, one
# hardware block per pass (--pmc with --kernel-trace only).  Output: gpurun_out/fill_law/{timing.txt,pmc_*.csv} and
# gpurun_out/fill_law/r05_fill_law.md (tools/fill_law_report.py).
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$R/gpurun_out/fill_law
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 120 $R/tools/fill_probe.bin 20 $FILL_SET > $OUT/timing.txt 2>&1; echo "fill timing rc=$?"
pass() {   # tag, counters...
  tag=$1; shift
  timeout ${PMC_TIMEOUT:-90} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw_$tag -o p -- $R/tools/fill_probe.bin 3 $FILL_SET > $OUT/pmc_$tag.log 2>&1
  rc=$?
  f=$(find $OUT/raw_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/fill_law_report.py --reduce "$f" > $OUT/pmc_$tag.csv
  echo "fill pmc $tag rc=$rc $(wc -l < $OUT/pmc_$tag.csv 2>/dev/null) rows"
  rm -rf $OUT/raw_$tag
}
# (TA_*, TD_* and TCC_* counter passes abort inside rocprofv3 on this image -- signal 6, then a hang until the time-out:
# call A of round 5 -- so the TCP / SQ blocks carry the evidence)
pass sq    SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
pass tcp1  TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
pass tcp2  TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcp3  TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
python $R/tools/fill_law_report.py $OUT > $OUT/r05_fill_law.md 2> $OUT/report.err; echo "report rc=$?"
head -40 $OUT/timing.txt
