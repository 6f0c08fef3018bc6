#!/bin/bash
# A/B: Adam moments + stored gradient as nontemporal accesses (libgm_hip_exp.so = gm_gemm.hip under -DGM_ADAM_NT)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_l
for i in 1 2 3; do
for lib in libgm_hip.so libgm_hip_exp.so; do
GM_LIB_PATH=$R/generative_models_amd/$lib timeout 300 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']['per_kernel_us_per_step']; print('$lib', 'step us %.2f'%(d['ms_per_step']*1e3), {k[:24]:v for k,v in r.items()})"
done; done | tee gpurun_out/r06_l/ab_adam_nt.txt
for lib in libgm_hip.so libgm_hip_exp.so libgm_hip.so libgm_hip_exp.so; do GM_LIB_PATH=$R/generative_models_amd/$lib timeout 300 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); d=d[0] if isinstance(d,list) else d; print('$lib', d.get('ms_per_step'))"; done | tee gpurun_out/r06_l/ab_adam_nt_1024.txt
