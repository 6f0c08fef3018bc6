#!/bin/bash
# round 3, call U: the driver-style 20-step run against the first-piece / ramp settings (same box, alternating)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; export TMPDIR=/tmp
run() { label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --reps 9 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-26s %.2f us/step, fixed %.0f us' % ('$label', d['ms_per_step']*1e3, d['run_fixed_cost_us']), sorted(d['config']['reps_ms_per_step'])[:5])"
}
for rep in 1 2; do
  run default GM_NOP=1
  run first1 GM_FIRST_PIECE=1
  run first4 GM_FIRST_PIECE=4
  run ramp_2_2_4 GM_RAMP=2,2,4,8,16
  run ramp_1_3 GM_RAMP=1,3,4,8,16
done
