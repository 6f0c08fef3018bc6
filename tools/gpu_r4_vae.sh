#!/bin/bash
# Round 4: the VAE batch's launch fusions (reparameterisation + decoder layer 1; loss sums + tick in the last dW pair;
# the two narrow backward GEMMs + reparameterisation backward as one launch)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
timeout 300 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_ops.py -q -m gpu -x -k "reparam or closing or pair or finalize or mid_chain" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
timeout 600 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -m gpu -x -k "vae or VAE" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
fi
for f in "1" "0" "1" "0"; do
  echo "GM_VAE_BWD_MID=$f: $(GM_VAE_BWD_MID=$f timeout 200 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print([(round(e["ms_per_step"]*1e3,2), round(e["img_s"])) for e in d])')"
done
