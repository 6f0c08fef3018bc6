#!/bin/bash
# round 3, call B: folded critic head -- op tests, whole GPU suite, same-box A/B of the headline step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "folded" ) > gpurun_out/r3b/folded_ops.log 2>&1
tail -15 gpurun_out/r3b/folded_ops.log
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dp.py::test_n_rank_engine_equals_one_rank_full_size ) > gpurun_out/r3b/gpu_all.log 2>&1
tail -12 gpurun_out/r3b/gpu_all.log
for rep in 1 2; do
  for f in 0 1; do
    GM_FOLD_HEAD=$f timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline > gpurun_out/r3b/long_fold${f}_$rep.json 2> gpurun_out/r3b/long_fold${f}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r3b/long_fold${f}_$rep.json").read().strip().splitlines()[-1])
print("fold=$f rep=$rep long: %.2f us/step" % (d["ms_per_step"]*1e3), d["roofline"]["per_kernel_us_per_step"])
PY
  done
done
for f in 0 1; do
  GM_FOLD_HEAD=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/r3b/s20_fold$f.json 2> gpurun_out/r3b/s20_fold$f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3b/s20_fold$f.json").read().strip().splitlines()[-1])
print("fold=$f driver-style: %.2f us/step, steady %.2f, fixed %.1f" % (d["ms_per_step"]*1e3, d["steady_us_per_step"], d["run_fixed_cost_us"]))
PY
done
