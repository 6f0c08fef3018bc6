#!/bin/bash
# Round 4, last call: re-take the passes whose kernels changed after the first profile run (headline incl. SQ, DRAGAN),
# variant summaries, the two committed bench lines, the one-device dry runs of the N = 2 / 4 bench flow
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
tools/gpu_r4_profiles.sh nsgan_b256,sq,dra_b256 2>&1 | grep "rc=" 
tools/gpu_r4_variants.sh "ra fisher be info" 2 2>&1 | grep "us / iteration\|rc="
tools/gpu_r4_final.sh 2>&1 | tail -20
cd "$R"; mkdir -p gpurun_out/final_r04
for n in 2 4; do
  GM_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/final_r04/r04_bench_dry_n$n.json 2> gpurun_out/final_r04/dry_n$n.err; echo "dry n=$n rc=$?"
  python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/final_r04/r04_bench_dry_n$n.json').read().strip().split('\n')[-1]); print('  ranks_seen', d['config'].get('ranks_seen'), d['config'].get('gradient_exchange'), round(d['ms_per_step']*1e3,1),'us')
except Exception as e: print('  unreadable', e)"
done
du -sh gpurun_out
