#!/bin/bash
# Round 4: RaGAN / Fisher critic steps on the folded head (two-phase prologue) -- tests, then us per iteration through
# the drop-in trainers with and without it (same box, alternating)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_ops.py -q -m gpu -x -k "ra or fisher or summation_order or folded" 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
for v in 1 0 1 0; do
  echo "GM_FOLD_HEAD_TP=$v: $(GM_FOLD_HEAD_TP=$v timeout 300 python tools/variant_times.py ra,fisher 3 2>/dev/null | tail -3 | tr '\n' ' ')"
done
