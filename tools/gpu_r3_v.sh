#!/bin/bash
# round 3, call V: which half of the folded head pays -- critic step (dW consumer) or generator step (dX consumer)?
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
run() { label=$1; shift
  env "$@" timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %.2f us/step' % ('$label', d['ms_per_step']*1e3), d['config']['reps_ms_per_step'])"
}
for rep in 1 2 3; do
  run fold_both GM_NOP=1
  run fold_G_only GM_FOLD_HEAD_D=0
  run fold_D_only GM_FOLD_HEAD_G=0
  run unfolded GM_FOLD_HEAD=0
done
