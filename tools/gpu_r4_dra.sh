#!/bin/bash
# Round 4: DRAGAN's stacked critic step -- tests, then the step with and without it; NSGAN step against a library
# built WITHOUT the head backward's gb2_add path (same box, alternating)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_fused_ops.py tests/test_gpu_dp.py -q -m gpu -x -k "dra or DRAGAN or summation_order or dragan" 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 1 0; do
  echo "GM_DRA_STACK=$v: $(GM_DRA_STACK=$v timeout 200 python bench.py --only dra_b256 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print([(round(e["ms_per_step"]*1e3,2), e["reps_ms_per_step"]) for e in d])')"
done
for lib in "" generative_models_amd/ab_libs/r4_nogb2.so "" generative_models_amd/ab_libs/r4_nogb2.so "" generative_models_amd/ab_libs/r4_nogb2.so; do
  echo "lib=${lib:-default}: $(GM_LIB_PATH=$lib timeout 200 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"]["reps_ms_per_step"])')"
done
