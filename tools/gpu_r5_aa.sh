#!/bin/bash
# Round 5, call AA: Trainer.train() per step by ring / graph size, epochs enqueued ahead or not
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/aa
for cfg in "auto" "64 512" "128 256" "32 512" "64 256" "32 128"; do
  if [ "$cfg" = "auto" ]; then E=""; else set -- $cfg; E="GM_GRAPH_ITERS=$1 GM_RING=$2"; fi
  echo "== [$cfg]"; env $E timeout 100 python tools/trainer_epoch_ab.py 2>&1 | grep -v amdgpu | head -4
done 2>&1 | tee gpurun_out/aa/trainer_ring_graph.txt
