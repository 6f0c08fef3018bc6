#!/bin/bash
# Round 4: does pre-staging cost the bs=1024 step anything?  (same box, alternating)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
run() {
  echo "$*: $(env "$@" timeout 200 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print([(round(e["ms_per_step"]*1e3,2), e["reps_ms_per_step"]) for e in d])')"
}
timeout 200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "stage_in" 2>&1 | tail -1
run GM_PRESTAGE=1; run GM_PRESTAGE=0; run GM_PRESTAGE=1; run GM_PRESTAGE=0
echo "bs=256 steady:"
for v in 1 0; do echo "GM_PRESTAGE=$v $(GM_PRESTAGE=$v timeout 200 python bench.py --steps 20 --warmup 5 --reps 5 --no-cpu-baseline --no-configs --sustained 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["ms_per_step"]*1e3,2), "steady", round(d["steady_us_per_step"],2))')"; done
