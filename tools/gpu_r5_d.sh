#!/bin/bash
# Round 5, call D: coalesced fragment loads (GM_COALESCED_LOADS) -- op-level correctness, isolated GEMM shapes and the
# NSGAN step against the same sources built with the round-4 loads (ab_libs/uncoalesced.so), alternating in one call.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -q -x > gpurun_out/d_tests_ops.log 2>&1; echo "ops tests rc=$?"; tail -3 gpurun_out/d_tests_ops.log | cut -c1-200
SHAPES="fwd:512:784:400 fwdsig:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784 dw:512:784:400 dwadam:512:784:400 dwadam:256:400:784 dw:256:20:400 fwd:768:784:400 dw:768:784:400"
for rep in 1 2; do
  for lib in "" generative_models_amd/ab_libs/uncoalesced.so; do
    echo "== lib=${lib:-default(coalesced)} rep $rep"; GM_LIB_PATH=$lib timeout 300 python tools/gemm_shapes_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | cut -c1-70
  done
done
for rep in 1 2; do for lib in "" generative_models_amd/ab_libs/uncoalesced.so; do
  echo "step lib=${lib:-default(coalesced)}: $(GM_STAGE_AHEAD=0 GM_LIB_PATH=$lib timeout 200 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"]["reps_ms_per_step"], {k:v for k,v in d["roofline"]["per_kernel_us_per_step"].items()})')"
done; done
timeout 900 python -m pytest tests/test_gpu_trainers.py -q > gpurun_out/d_tests1.log 2>&1; echo "tests trainers rc=$?"; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/d_tests1.log | cut -c1-220 | tail -12
for rep in 1 2; do for v in 1 0; do
  echo "GM_STAGE_AHEAD=$v: $(GM_STAGE_AHEAD=$v timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"
done; done
