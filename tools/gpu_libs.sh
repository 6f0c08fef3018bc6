#!/bin/bash
# bench.py against several builds of the library (GM_LIB_PATH): bash tools/gpu_libs.sh lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export GM_BENCH_VERBOSE=1
for lib in "" "$@" ""; do
  echo "== lib=${lib:-default}"
  GM_LIB_PATH=$lib timeout 200 python bench.py --no-cpu-baseline 2> gpurun_out/e.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', d['ms_per_step'])"
  grep "us  " gpurun_out/e.log | tr -s ' ' | cut -d' ' -f3- | paste -sd'|' | cut -c1-400
done
