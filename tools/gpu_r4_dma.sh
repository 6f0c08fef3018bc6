#!/bin/bash
# round 4: the LDS-DMA weight gradient against round 3's forms, same box (isolated launches), + the ops tests
mkdir -p gpurun_out/r4g
SH="dw:2048:784:400 dw:1024:784:400 dw:768:784:400 dw:512:784:400 dw:256:784:400 dw:2048:400:784 dw:1024:400:784 dw:512:400:784 dw:256:400:784"
for v in 1 0; do
  echo "== GM_DW_DMA=$v" >> gpurun_out/r4g/shapes.log
  GM_DW_DMA=$v timeout 300 python tools/gemm_shapes_bench.py $SH >> gpurun_out/r4g/shapes.log 2>&1
done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r4g/ops.log 2>&1; echo "ops rc=$?" >> gpurun_out/r4g/ops.log
tail -3 gpurun_out/r4g/ops.log
