#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_v
for lib in libgm_hip.so libgm_hip_exp.so; do for v in ns ls; do GM_LIB_PATH=$R/generative_models_amd/$lib timeout 200 python tools/run_digest.py $v 256 64 2>&1 | tail -1 | sed "s/^/$lib /"; done; done | tee gpurun_out/r06_v/digest.txt
for i in 1 2 3; do
for lib in libgm_hip.so libgm_hip_exp.so; do
GM_LIB_PATH=$R/generative_models_amd/$lib timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']['per_kernel_us_per_step']; print('$lib', 'step us %.2f'%(d['ms_per_step']*1e3), {k[:24]:v for k,v in r.items() if 'dx_head' in k})"
done; done | tee gpurun_out/r06_v/ab_fold2_pre.txt
