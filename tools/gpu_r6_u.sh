#!/bin/bash
# Round 6: the N = 2 / 4 bench flows on one device (dry run of the multi-process code path: torchrun, RCCL control plane,
# peer exchange with both ranks on one GPU, first-contact self-check) on the final tree, and the two-rank GPU tests
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/final_r06
for n in 2 4; do
GM_BENCH_ONE_DEVICE=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/final_r06/r06_bench_dry_n$n.json 2> gpurun_out/final_r06/dry_n$n.err; echo "dry n=$n rc=$?"
tail -c 900 gpurun_out/final_r06/r06_bench_dry_n$n.json | cut -c1-900; tail -2 gpurun_out/final_r06/dry_n$n.err | cut -c1-300
done
