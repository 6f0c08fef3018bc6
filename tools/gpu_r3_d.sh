#!/bin/bash
# round 3, call D: folded head with the prologue behind the first operand loads; VAE epilogue fusions
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -x -q -k "folded or head or dw_adam or pair or epilogue or finalize or sqerr or reparam" ) > gpurun_out/r3d/ops.log 2>&1
tail -6 gpurun_out/r3d/ops.log
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dp.py::test_n_rank_engine_equals_one_rank_full_size ) > gpurun_out/r3d/gpu_all.log 2>&1
tail -6 gpurun_out/r3d/gpu_all.log
show() {
python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: %.2f us/step (steady %.2f, fixed %.0f)" % (sys.argv[2], d["ms_per_step"]*1e3, d["steady_us_per_step"], d["run_fixed_cost_us"]),
      {k.split("<")[0][7:]+"<"+k.split("<")[1][:24]: v for k, v in d["roofline"]["per_kernel_us_per_step"].items()})
PY
}
for rep in 1 2; do
  for f in 0 1; do
    GM_FOLD_HEAD=$f timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline > gpurun_out/r3d/long_f${f}_$rep.json 2> gpurun_out/r3d/long_f${f}_$rep.err
    show gpurun_out/r3d/long_f${f}_$rep.json "fold=$f rep=$rep long"
  done
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/r3d/s20.json 2> gpurun_out/r3d/s20.err
show gpurun_out/r3d/s20.json "defaults, driver-style"
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $cfg
    GM_VAE_FUSE_SQERR=$1 GM_VAE_FUSE_REPARAM_BWD=$2 timeout 300 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 > gpurun_out/r3d/vae_s$1_r$2_$rep.json 2> gpurun_out/r3d/vae_s$1_r$2_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r3d/vae_s$1_r$2_$rep.json").read().strip().splitlines()[-1])
print("VAE sqerr-fused=$1 reparam-bwd-fused=$2 rep=$rep:", [(round(e["img_s"]), round(e["ms_per_step"]*1e3, 2)) for e in d])
PY
  done
done
for f in 0 1; do
  GM_FOLD_HEAD=$f timeout 300 python bench.py --only wgp_b256 --steps 200 --warmup 20 --reps 3 > gpurun_out/r3d/wgp_f$f.json 2> gpurun_out/r3d/wgp_f$f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3d/wgp_f$f.json").read().strip().splitlines()[-1])
print("WGAN-GP fold(G step)=$f:", [(round(e["img_s"]), round(e["ms_per_step"]*1e3, 2)) for e in d])
PY
done
