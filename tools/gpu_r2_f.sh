#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/f_pytest.log 2>&1; tail -25 gpurun_out/f_pytest.log
cat gpurun_out/parity_50step.json
