#!/bin/bash
# Round 6 call B: the full-size parity tests with the element-masked parameter bounds
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_b; rm -f gpurun_out/parity_full_size.jsonl
timeout 1500 python -m pytest tests/test_gpu_trainers.py -q -m gpu -k "full_size or baseline_configs or vae_b512_ragged or reference_default_batch or vae_reference_default" > gpurun_out/r06_b/pytest.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r06_b/pytest.txt
