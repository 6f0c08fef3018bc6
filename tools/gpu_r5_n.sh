#!/bin/bash
# Round 5, last call: the whole GPU suite on the final build (parity records), smoke, the two committed bench lines.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out/final_r05; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/n_tests.log 2>&1; echo "gpu suite rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/n_tests.log | cut -c1-220 | tail -8
bash tools/gpu_r5_final.sh
