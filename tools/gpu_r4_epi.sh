#!/bin/bash
mkdir -p gpurun_out/r4j
SH="dwadam:512:784:400 dwadam:256:400:784 dwadam:2048:784:400 dw:512:784:400 fwd:512:784:400 fwdsig:512:400:784 fwd:256:784:400 dx:256:784:400 dx:256:400:784 fwd:2048:784:400 dx:1024:784:400"
for v in 1 0 1 0; do
  echo "== GM_VEC_EPI=$v" >> gpurun_out/r4j/epi.log
  GM_VEC_EPI=$v timeout 300 python tools/gemm_shapes_bench.py $SH 2>&1 | grep -v amdgpu.ids | cut -c1-62 >> gpurun_out/r4j/epi.log
done
cat gpurun_out/r4j/epi.log
