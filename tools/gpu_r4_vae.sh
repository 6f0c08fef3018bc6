#!/bin/bash
# Round 4: the VAE batch's two launch fusions (reparameterisation + decoder layer 1; loss sums + tick in the last dW pair)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
timeout 300 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_ops.py -q -m gpu -x -k "reparam or closing or pair or finalize" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_trainers.py -q -m gpu -x -k "vae or VAE" 2>&1 | tail -5
fi
for f in "1 1" "0 1" "1 0" "0 0" "1 1" "0 0"; do set -- $f
  echo "reparam_fwd=$1 fin_in_dw=$2: $(GM_VAE_FUSE_REPARAM_FWD=$1 GM_VAE_FINALIZE_IN_DW=$2 timeout 200 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print([(round(e["ms_per_step"]*1e3,2), round(e["img_s"])) for e in d])')"
done
