#!/bin/bash
# after removing the tile-map table and giving the interleaved form its own kernel instantiations: the bs=256 step
# against the library built from commit 4a4ab25's gm_gemm.hip, same box, alternating; op tests; bs=1024 step
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
for rep in 1 2; do
for lib in generative_models_amd/ab_libs/old_gemm.so generative_models_amd/libgm_hip.so; do
echo -n "$lib: "; GM_LIB_PATH=$R/$lib timeout 200 python bench.py --steps 1000 --warmup 100 --reps 3 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), d['config'].get('reps_ms_per_step'))"
done; done
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 200 python bench.py --only ns_b1024 --steps 400 --warmup 50 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([ (e['workload'][:24], round(e['ms_per_step']*1e3,2)) for e in d])"
