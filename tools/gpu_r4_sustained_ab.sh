#!/bin/bash
# Round 4: does the 6 s sustained window in front of the configs section change the configs' figures?
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for s in 0 6 0; do
  echo "--sustained $s: $(timeout 600 python bench.py --no-cpu-baseline --sustained $s 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
cs=d.get("other_configs") or d.get("configs") or []
print(round(d["ms_per_step"]*1e3,2), [round(c["ms_per_step"]*1e3,1) for c in cs])')"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
