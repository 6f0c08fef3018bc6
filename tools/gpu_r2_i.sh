#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/i_pytest.log 2>&1; tail -4 gpurun_out/i_pytest.log
python - <<'PY'
import sys, json, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'generative_models_amd/src')
import bench, os
dev = torch.device("cuda", 0)
for stack in ("1", "0"):
    os.environ["GM_WGP_STACK"] = stack
    eng, secs = bench.bench_gan("wgp", 256, 50, 400, 3, dev, lrs=(1e-4, 1e-4))
    import numpy as np
    print("WGAN-GP bs=256 D_steps=1, stacked=%s: %.1f us/iteration" % (stack, float(np.median(secs)) / 400 * 1e6))
    del eng
PY
