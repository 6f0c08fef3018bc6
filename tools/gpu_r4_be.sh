#!/bin/bash
# Round 4: BEGAN -- the autoencoder critic's two weight gradients as one launch
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_dp.py -q -m gpu -x -k "be or began or BEGAN" 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
for v in 1 0 1 0; do
  echo "GM_PAIR_DW=$v: $(GM_PAIR_DW=$v timeout 300 python tools/variant_times.py be 3 2>/dev/null | tail -1)"
done
