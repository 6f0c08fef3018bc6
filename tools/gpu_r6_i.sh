#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_i
for i in 1 2 3; do
for lib in libgm_hip.so libgm_hip_lean.so; do
GM_LIB_PATH=$R/generative_models_amd/$lib timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']['per_kernel_us_per_step']; print('$lib', 'step us %.2f'%(d['ms_per_step']*1e3), {k[:22]:v for k,v in r.items() if 'head' in k})"
done; done | tee gpurun_out/r06_i/ab_fold_side.txt
for f in 1 2 1 2; do GM_FOLD_HEAD=$f timeout 300 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); d=d[0] if isinstance(d,list) else d; print('GM_FOLD_HEAD=$f', d.get('ms_per_step'), d.get('roofline',{}).get('per_kernel_us_per_step'))"; done | tee gpurun_out/r06_i/fold1024.txt
