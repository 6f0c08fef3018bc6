#!/bin/bash
# Round 5, call G: small A/Bs on the final build: piece-event join on / off, 16x32 tiles for the 2B-row forwards, first piece.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
run20() { echo "$*: $(env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2), "fixed", round(d["run_fixed_cost_us"],1), d["config"]["reps_ms_per_step"])')"; }
for rep in 1 2; do
  run20 GM_JOIN_PIECES=1
  run20 GM_JOIN_PIECES=0
  run20 GM_JOIN_PIECES=0 GM_T12_MAX_TILES=208
  run20 GM_JOIN_PIECES=0 GM_FIRST_PIECE=4
  run20 GM_JOIN_PIECES=0 GM_PRESTAGE=0
done
