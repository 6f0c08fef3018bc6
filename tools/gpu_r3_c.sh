#!/bin/bash
# round 3, call C: folded head (contiguous partials) + Adam prefetch -- tests, same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "folded or head or dw_adam or pair" ) > gpurun_out/r3c/ops.log 2>&1
tail -6 gpurun_out/r3c/ops.log
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dp.py::test_n_rank_engine_equals_one_rank_full_size ) > gpurun_out/r3c/gpu_all.log 2>&1
tail -6 gpurun_out/r3c/gpu_all.log
show() {
python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: %.2f us/step (steady %.2f, fixed %.0f)" % (sys.argv[2], d["ms_per_step"]*1e3, d["steady_us_per_step"], d["run_fixed_cost_us"]),
      {k.split("<")[0][7:]+"<"+k.split("<")[1][:24]: v for k, v in d["roofline"]["per_kernel_us_per_step"].items()})
PY
}
for rep in 1 2; do
  for cfg in "0 0" "0 1" "1 0" "1 1"; do
    set -- $cfg
    GM_FOLD_HEAD=$1 GM_ADAM_PREFETCH=$2 timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline > gpurun_out/r3c/long_f$1_p$2_$rep.json 2> gpurun_out/r3c/long_f$1_p$2_$rep.err
    show gpurun_out/r3c/long_f$1_p$2_$rep.json "fold=$1 prefetch=$2 rep=$rep long"
  done
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/r3c/s20.json 2> gpurun_out/r3c/s20.err
show gpurun_out/r3c/s20.json "defaults, driver-style"
for n in 4 8; do
  GM_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $n --steps 20 --warmup 5 > gpurun_out/r3c/bench_dry_n$n.json 2> gpurun_out/r3c/bench_dry_n$n.err
  echo "dry run N=$n rc=$?"; tail -c 400 gpurun_out/r3c/bench_dry_n$n.json
done
