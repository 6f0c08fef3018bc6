#!/bin/bash
# Build tools/slab_probe.bin (links the in-tree libgm_hip.so for the shipped-kernel baselines and rocBLAS for the vendor one).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value "$@" $R/tools/slab_probe.hip \
  -L$R/generative_models_amd -lgm_hip -L/opt/rocm/lib -lrocblas -Wl,-rpath,'$ORIGIN/../generative_models_amd' -Wl,-rpath,/opt/rocm/lib \
  -o $R/tools/slab_probe.bin
echo built $R/tools/slab_probe.bin
