#!/bin/bash
# Round 6: the whole GPU suite + smoke (what the driver runs at round end)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_suite; rm -f gpurun_out/parity_full_size.jsonl gpurun_out/parity_50step.json
timeout 2400 python -m pytest tests/ -q -m gpu -x > gpurun_out/r06_suite/pytest.txt 2>&1; echo "suite rc=$?"; tail -8 gpurun_out/r06_suite/pytest.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
