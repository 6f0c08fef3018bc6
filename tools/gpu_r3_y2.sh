#!/bin/bash
# FETCH_SIZE of the isolated weight-gradient launch with the balanced XCD map off / on
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for k in 0 1; do
  rm -rf /tmp/pm_$k
  GM_XMAP_MIN_K=$k timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm_$k -o x -- python $R/tools/gemm_shapes_bench.py dw:2048:784:400 dw:512:784:400 > /tmp/pm_$k.log 2>&1
  echo "== GM_XMAP_MIN_K=$k"; grep "^dw" /tmp/pm_$k.log
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pm_$k/**/*counter_collection.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='FETCH_SIZE' and 'gemm16' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:60], r['Grid_Size'])].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, len(v), 'avg FETCH_SIZE KiB %.1f -> x2 MB %.2f' % (sum(v)/len(v), 2*sum(v)/len(v)/1024))
PY
done
