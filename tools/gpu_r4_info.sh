#!/bin/bash
# Round 4: InfoGAN -- Q loss with the row in registers, the Q step's four weight gradients as two pairs
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_ops.py tests/test_gpu_dp.py -q -m gpu -x -k "info" 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
for v in 1 0 1 0; do
  echo "GM_PAIR_DW=$v: $(GM_PAIR_DW=$v timeout 300 python tools/variant_times.py info,be 3 2>/dev/null | tail -2 | tr '\n' ' ')"
done
