#!/bin/bash
# A/B on the GPU box: parity tests, then bench.py with launch-fusion toggles switched off one at a
# time.  Usage: bash tools/gpu_ab.sh "GM_X=0" "GM_Y=0" ...   (each argument = one extra bench run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export GM_BENCH_VERBOSE=1
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -5 | tee gpurun_out/ab_ops.log
timeout 400 python -m pytest tests/test_gpu_trainers.py -q 2>&1 | tail -8 | tee gpurun_out/ab_trainers.log
run() {
  echo "== $1" | tee -a gpurun_out/ab_bench.log
  env $1 timeout 200 python bench.py --no-cpu-baseline 2> gpurun_out/ab_err.log | tee -a gpurun_out/ab_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', d['ms_per_step'])"
  grep -E "us  " gpurun_out/ab_err.log | tail -12
}
run "GM_NOOP=1"
for t in "$@"; do run "$t"; done
run "GM_NOOP=2"
