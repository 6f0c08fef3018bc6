#!/bin/bash
# Round-6 profiles: rocprofv3 kernel-trace stats + the PMC passes (each in its OWN run: --pmc with --kernel-trace
# only) for the headline workload and the configs 3/4/5 legs; summaries land in gpurun_out/profiles_r06/.
# usage: gpu_r4_profiles.sh [all | comma list of nsgan_b256,ns_b1024,wgp_b256,dra_b256,vae_b512,sq]
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out/profiles_r06; export TMPDIR=/tmp
OUT=$R/gpurun_out/profiles_r06
cd /tmp
prof() {   # tag, description, bench args...
  tag=$1; what=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_$tag -o ns -- python $R/bench.py "$@" > $OUT/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/pr_$tag.log; echo "$tag stats rc=$?"
  python $R/profiles/make_summary.py $R/gpurun_out/pr_$tag $tag $OUT > /dev/null
  T=$(find $R/gpurun_out/pr_$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_gaps.py $T > $OUT/${tag}_gaps.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -o ns -- python $R/bench.py "$@" > $R/gpurun_out/pmc_${tag}_$c.log 2>&1; echo "$tag pmc $c rc=$?"
  done
  python $R/profiles/make_pmc_summary.py $R/gpurun_out/pmc_${tag}_ $tag $OUT "$what" > /dev/null
  find $R/gpurun_out -name "*kernel_trace.csv" -delete; find $R/gpurun_out -name "*counter_collection.csv" -delete
}
WHICH="${1:-all}"
want() { [ "$WHICH" = "all" ] || [[ ",$WHICH," == *",$1,"* ]]; }
want nsgan_b256 && prof r06_nsgan_b256 "NSGAN bs=256 (headline), bench.py --steps 400 --warmup 50" --steps 400 --warmup 50 --reps 1 --no-cpu-baseline --no-configs --sustained 0
if want sq; then
  tag=r06_nsgan_b256
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/sq_$tag -o ns -- python $R/bench.py --steps 400 --warmup 50 --reps 1 --no-cpu-baseline --no-configs --sustained 0 > $R/gpurun_out/sq_$tag.log 2>&1; echo "$tag sq pmc rc=$?"
  python $R/tools/pmc_sq_summary.py $R/gpurun_out/sq_$tag $OUT/${tag}_sq_pmc.json > $OUT/${tag}_sq_pmc.txt 2>&1
  find $R/gpurun_out -name "*counter_collection.csv" -delete
fi
want ns_b1024 && prof r06_ns_b1024 "NSGAN bs=1024 (configs[4] single-GPU leg), bench.py --only ns_b1024" --only ns_b1024 --steps 200 --warmup 20 --reps 1
want wgp_b256 && prof r06_wgp_b256 "WGAN-GP bs=256 D_steps=1 (configs[2]), bench.py --only wgp_b256" --only wgp_b256 --steps 200 --warmup 20 --reps 1
want dra_b256 && prof r06_dra_b256 "DRAGAN bs=256 D_steps=1 (VERDICT r2 item 10), bench.py --only dra_b256" --only dra_b256 --steps 200 --warmup 20 --reps 1
want vae_b512 && prof r06_vae_b512 "VAE bs=512 full epochs (configs[3]), bench.py --only vae_b512" --only vae_b512 --steps 200 --warmup 20 --reps 1
# variants outside BASELINE.json: kernel-trace summaries of InfoGAN / BEGAN through their trainers
variant() {
  v=$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pr_var_$v -o ns -- python $R/bench.py --only ${v}_b256 --steps 200 --warmup 20 --reps 1 > $OUT/r06_variant_${v}_bench_under_rocprof.json 2> $R/gpurun_out/pr_var_$v.log; echo "variant $v rc=$?"
  python $R/profiles/make_summary.py $R/gpurun_out/pr_var_$v r06_variant_$v $OUT > /dev/null
  rm -f $OUT/r06_variant_${v}_kernel_stats.csv
}
want variants && { variant info; variant be; }
find $R/gpurun_out -name "*kernel_trace.csv" -delete
ls $OUT | head -80; du -sh $R/gpurun_out
