#!/bin/bash
# Round 5, call P: the whole GPU suite on the tree with the push exchange / new region layout
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/p
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/p/tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/p/tests.log
