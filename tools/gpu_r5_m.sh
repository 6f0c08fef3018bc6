#!/bin/bash
# Round 5, call M: the LDS macro-tile kernel at 64x64 (one workgroup per CU) against 32x64 (two per CU) on the bs=1024 shapes
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; export TMPDIR=/tmp
SHAPES="fwd:2048:784:400 fwdsig:2048:400:784 fwd:1024:784:400 dx:1024:784:400 dx:1024:400:784 dx:2048:784:400"
for rep in 1 2; do for c in 0 1 2; do
  echo "== GM_LDS_CFG=$c"; GM_LDS_CFG=$c timeout 300 python tools/gemm_shapes_bench.py $SHAPES 2>&1 | grep -v amdgpu.ids | cut -c1-120
done; done
for c in 0 2; do
  echo "bs1024 GM_LDS_CFG=$c: $(GM_LDS_CFG=$c timeout 200 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1])[0]; print(round(d["ms_per_step"]*1e3,2), d["reps_ms_per_step"])')"
done
