#!/bin/bash
# Round 5, call E: prefetch-all window x coalesced loads: the NSGAN step and isolated shapes for five builds of gm_gemm.hip.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused_ops.py -q -x > gpurun_out/e_tests_ops.log 2>&1; echo "ops tests (default build) rc=$?"; tail -2 gpurun_out/e_tests_ops.log | cut -c1-200
L=generative_models_amd/ab_libs
for rep in 1 2; do for lib in "" $L/pf_x_k.so $L/pf_only.so $L/xonly.so $L/r4loads.so; do
  echo "step lib=${lib:-default(pf+x)}: $(GM_STAGE_AHEAD=0 GM_LIB_PATH=$lib timeout 200 python bench.py --steps 512 --warmup 64 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json,re; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), d["config"]["reps_ms_per_step"], {re.sub(r"gemm16_|_kernel","",k)[:30]:v for k,v in d["roofline"]["per_kernel_us_per_step"].items()})')"
done; done
SHAPES="fwd:512:784:400 fwdsig:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784 dw:512:784:400 dwadam:512:784:400 dwadam:256:400:784 fwd:768:784:400 dw:768:784:400 fwd:1024:784:400 dw:1024:784:400"
for lib in "" $L/r4loads.so; do
  echo "== lib=${lib:-default(pf+x)}"; GM_LIB_PATH=$lib timeout 300 python tools/gemm_shapes_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | cut -c1-70
done
for lib in "" $L/r4loads.so; do
  echo "bs1024 lib=${lib:-default}: $(GM_LIB_PATH=$lib timeout 200 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1])[0]; print(round(d["ms_per_step"]*1e3,2), d["reps_ms_per_step"])')"
  echo "wgp lib=${lib:-default}: $(GM_LIB_PATH=$lib timeout 200 python bench.py --only wgp_b256 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1])[0]; print(round(d["ms_per_step"]*1e3,2), d["reps_ms_per_step"])')"
  echo "vae lib=${lib:-default}: $(GM_LIB_PATH=$lib timeout 200 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1])[0]; print(round(d["ms_per_step"]*1e3,2))')"
done
