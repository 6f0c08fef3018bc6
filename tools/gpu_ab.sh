#!/bin/bash
# A/B on the GPU box: parity tests, then bench.py with each launch-fusion toggle switched off.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export GM_BENCH_VERBOSE=1
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -5 | tee gpurun_out/ab_ops.log
timeout 400 python -m pytest tests/test_gpu_trainers.py -q 2>&1 | tail -8 | tee gpurun_out/ab_trainers.log
run() {
  echo "== $1" | tee -a gpurun_out/ab_bench.log
  env $1 timeout 200 python bench.py --no-cpu-baseline 2> gpurun_out/ab_err_$2.log | tee -a gpurun_out/ab_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', d['ms_per_step'], d['roofline']['per_kernel_us_per_step'])"
  grep -E "us  " gpurun_out/ab_err_$2.log | tail -12
}
run "GM_NOOP=1" all_on
run "GM_HEAD_FINAL=0" head_final_off
run "GM_GROUP_HEAD=0" group_head_off
run "GM_RIDE_GATHER=0" ride_gather_off
run "GM_PAIR_DW=0" pair_dw_off
run "GM_NARROW_TILES=0" narrow_off
run "GM_NOOP=2" all_on_again
