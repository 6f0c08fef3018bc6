#!/bin/bash
# Round 5, call W: the whole GPU suite on the tree without the pre-stage, then the committed bench lines
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/w
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/w/tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/w/tests.log | cut -c1-250
bash tools/gpu_r5_final.sh
