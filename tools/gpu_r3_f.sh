#!/bin/bash
# round 3, call F: GM_BATCH_LOADS in the real step (A/B, same box) + in-situ kernel times under rocprofv3
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/r3f; export TMPDIR=/tmp
OUT=$R/gpurun_out/r3f
for rep in 1 2 3; do
  for b in 0 1; do
    GM_BATCH_LOADS=$b timeout 300 python bench.py --steps 2000 --warmup 200 --reps 3 --no-configs --no-cpu-baseline > $OUT/long_b${b}_$rep.json 2> $OUT/long_b${b}_$rep.err
    python - <<PY
import json
d=json.loads(open("$OUT/long_b${b}_$rep.json").read().strip().splitlines()[-1])
print("batch_loads=$b rep=$rep: %.2f us/step" % (d["ms_per_step"]*1e3), d["config"]["reps_ms_per_step"])
PY
  done
done
for b in 0 1; do
  GM_BATCH_LOADS=$b timeout 300 python bench.py --only vae_b512 --steps 200 --warmup 20 --reps 3 > $OUT/vae_b$b.json 2> $OUT/vae_b$b.err
  GM_BATCH_LOADS=$b timeout 300 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 > $OUT/ns1024_b$b.json 2> $OUT/ns1024_b$b.err
  python - <<PY
import json
for t in ("vae", "ns1024"):
    d=json.loads(open("$OUT/%s_b$b.json" % t).read().strip().splitlines()[-1])
    print(t, "batch_loads=$b:", [(round(e["img_s"]), round(e["ms_per_step"]*1e3, 2)) for e in d])
PY
done
cd /tmp
for b in 1; do
  tag=nsgan_b256_batch$b
  GM_BATCH_LOADS=$b timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_$tag -o ns -- python $R/bench.py --steps 400 --warmup 50 --reps 1 --no-cpu-baseline --no-configs > $OUT/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/pr_$tag.log; echo "$tag stats rc=$?"
  python $R/profiles/make_summary.py $R/gpurun_out/pr_$tag $tag $OUT > /dev/null
  head -12 $OUT/${tag}_summary.md
  find $R/gpurun_out -name "*kernel_trace.csv" -delete
done
