#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r06_t
timeout 1500 python -m pytest tests/ -q -m gpu -x -k "1024 or b1024 or lockstep or graph_size" > gpurun_out/r06_t/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_t/pytest.txt | cut -c1-300
for i in 1 2 3; do
timeout 300 python bench.py --only ns_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); d=d[0] if isinstance(d,list) else d; print('ns_b1024', d.get('ms_per_step'))"
timeout 300 python bench.py --only ls_b1024 --steps 200 --warmup 20 --reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); d=d[0] if isinstance(d,list) else d; print('ls_b1024', d.get('ms_per_step'))"
done | tee gpurun_out/r06_t/b1024.txt
