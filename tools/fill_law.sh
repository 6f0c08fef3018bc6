#!/bin/bash
# Fill-law measurement: tools/fill_probe.bin alone (timing) and under rocprofv3 --pmc passes, one
# hardware block per pass (--pmc with --kernel-trace only).  Output: gpurun_out/fill_law/{timing.txt,pmc_*.csv} and
# gpurun_out/fill_law/r05_fill_law.md (tools/fill_law_report.py).
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$R/gpurun_out/fill_law
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 120 $R/tools/fill_probe.bin 20 > $OUT/timing.txt 2>&1; echo "fill timing rc=$?"
pass() {   # tag, counters...
  tag=$1; shift
  timeout 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw_$tag -o p -- $R/tools/fill_probe.bin 3 > $OUT/pmc_$tag.log 2>&1
  rc=$?
  f=$(find $OUT/raw_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/fill_law_report.py --reduce "$f" > $OUT/pmc_$tag.csv
  echo "fill pmc $tag rc=$rc $(wc -l < $OUT/pmc_$tag.csv 2>/dev/null) rows"
  rm -rf $OUT/raw_$tag
}
pass sq    SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
pass ta    TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass td    TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum
pass tcp1  TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
pass tcp2  TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcp3  TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass tcc   TCC_BUSY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum
python $R/tools/fill_law_report.py $OUT > $OUT/r05_fill_law.md 2> $OUT/report.err; echo "report rc=$?"
head -40 $OUT/timing.txt
