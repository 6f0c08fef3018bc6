#!/bin/bash
# Round 5, call V: GM_PRESTAGE=0 (gate wait kernel + ungated in-graph copy) against the side-stream pre-stage on the other configs
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/v
for rep in 1 2; do for pre in 1 0; do for c in ns_b1024 wgp_b256 dra_b256; do
  echo "GM_PRESTAGE=$pre $c: $(GM_PRESTAGE=$pre timeout 200 python bench.py --only $c --steps 400 --warmup 40 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1])[0]; print(round(d["ms_per_step"]*1e3,2), d["reps_ms_per_step"])')"
done; done; done 2>&1 | tee gpurun_out/v/bench_ab.txt
