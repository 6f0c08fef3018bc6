#!/bin/bash
# Round 5, call R: epochs enqueued ahead of the loss read-back (trainers._train): the trainer tests + train(3) / train(10) either way
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r
timeout 900 python -m pytest tests/test_gpu_trainers.py -q -p no:cacheprovider > gpurun_out/r/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r/tests.log | cut -c1-220
timeout 400 python tools/trainer_epoch_ab.py gpurun_out/r/r05_trainer_epoch_pipeline_ab.json 2>&1 | grep -v amdgpu.ids
