#!/bin/bash
# round 3, call G: tile-shape switches measured INSIDE the real step (same box, alternating), one-rank DP structure cost
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/r3g; export TMPDIR=/tmp
OUT=$R/gpurun_out/r3g
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline > $OUT/$label.json 2> $OUT/$label.err
  python - "$OUT/$label.json" "$label" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %.2f us/step" % (sys.argv[2], d["ms_per_step"]*1e3), d["config"]["reps_ms_per_step"],
      {k.split("<")[0][7:]+"<"+k.split("<")[1][:26]: v for k, v in d["roofline"]["per_kernel_us_per_step"].items()})
PY
}
for rep in 1 2; do
  run base_$rep GM_NOP=1
  run narrow256_$rep GM_NARROW_MAX_TILES=256
  run nowide_$rep GM_WIDE_TILES=0
  run waves8_$rep GM_WAVES8=1
done
run forcedp GM_FORCE_DP=1
run forcedp_nofold GM_FORCE_DP=1 GM_FOLD_HEAD=0
