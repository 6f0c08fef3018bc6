#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/ac
{ timeout 100 python tools/trainer_pipeline_probe.py | head -3
for cfg in "auto" "32 128" "auto" "32 128"; do
  if [ "$cfg" = "auto" ]; then E=""; else set -- $cfg; E="GM_GRAPH_ITERS=$1 GM_RING=$2"; fi
  echo "== [$cfg]"; env $E timeout 100 python tools/trainer_epoch_ab.py 2>&1 | grep -v amdgpu | head -4
done; } 2>&1 | grep -v amdgpu | tee gpurun_out/ac/trainer_after_fix.txt | cut -c1-600
