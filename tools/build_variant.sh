#!/bin/bash
# Build a VARIANT of libgm_hip.so with extra compile-time defines for gm_gemm.hip (the other objects are
# reused from the in-tree build): generative_models_amd/ab_libs/<name>.so, selected with GM_LIB_PATH.
#   tools/build_variant.sh xdirect -DGM_XDIRECT=1
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/generative_models_amd/csrc
NAME=$1; shift
mkdir -p $R/generative_models_amd/ab_libs /tmp/gm_variant_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c $C/gm_gemm.hip -o /tmp/gm_variant_$NAME/gm_gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gm_variant_$NAME/gm_gemm.o $C/gm_ops.o $C/gm_fused.o $C/gm_comm.o $C/gm_hostrng.o -lpthread \
  -o $R/generative_models_amd/ab_libs/$NAME.so
echo "built generative_models_amd/ab_libs/$NAME.so ($*)"
