#!/bin/bash
# Round 6 call E: many-row LDS kernel configurations, isolated (GM_LDS_FORCE: 0 = shipped rule)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_e
SH="fwd:2048:784:400 fwd:2048:400:784 fwd:1024:784:400 dx:1024:400:784 dx:1024:784:400"
for f in 0 3 4 5 6 0 3; do
  echo "== GM_LDS_FORCE=$f"; GM_LDS_FORCE=$f timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06_e/lds_cfgs.txt 2>&1
cat gpurun_out/r06_e/lds_cfgs.txt
