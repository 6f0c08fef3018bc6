#!/bin/bash
# Round 6 call A: parameter-deviation attribution data + the many-row GEMM shapes against the vendor
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_a
timeout 900 python tools/param_attribution_probe.py > gpurun_out/r06_a/attribution.txt 2>&1; echo "attr rc=$?"
timeout 300 python tools/gemm_shapes_bench.py fwd:2048:784:400 fwd:2048:400:784 dx:2048:784:400 dx:2048:400:784 dx:1024:784:400 dx:1024:400:784 fwd:1024:784:400 dw:2048:784:400 dw:1024:784:400 dw:1024:400:784 dwadam:2048:784:400 > gpurun_out/r06_a/shapes.txt 2>&1; echo "shapes rc=$?"
tail -30 gpurun_out/r06_a/attribution.txt; cat gpurun_out/r06_a/shapes.txt
