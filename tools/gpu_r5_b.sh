#!/bin/bash
# Round 5, call B: stage-ahead rider (tests + 20-step A/B), fill-probe access patterns with TCP counters, VAE divergence probe.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
timeout 900 python -m pytest tests/test_gpu_trainers.py -q > gpurun_out/b_tests1.log 2>&1; echo "tests trainers rc=$?"; tail -15 gpurun_out/b_tests1.log | cut -c1-300
for rep in 1 2; do for v in 1 0; do
  echo "GM_STAGE_AHEAD=$v: $(GM_STAGE_AHEAD=$v timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"
done; done
FILL_SET=patterns PMC_TIMEOUT=60 bash tools/fill_law.sh > gpurun_out/b_fill.log 2>&1; grep -v '^find\|No such' gpurun_out/b_fill.log | tail -16
timeout 400 python tools/vae_divergence_probe.py 100 50000 > gpurun_out/b_vae100.log 2>&1; cat gpurun_out/b_vae100.log
timeout 200 python tools/vae_divergence_probe.py 512 50000 > gpurun_out/b_vae512.log 2>&1; head -4 gpurun_out/b_vae512.log
