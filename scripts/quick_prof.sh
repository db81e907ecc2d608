#!/bin/bash
# quick GPU loop: bench (3 steps) under rocprofv3 --kernel-trace --stats; prints the value and the per-kernel table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra > gpurun_out/bench_$TAG.log 2>&1
grep '^{' gpurun_out/bench_$TAG.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('VALUE',d['value'],'ms',d['ms_per_step'],d['stage_ms_per_step'],'rounds',d['bwt_rounds'],'A/n',d['A_over_n'],d['checks'])"
python3 - $TAG <<'PY'
import csv,glob,sys
f=glob.glob(f'gpurun_out/prof_{sys.argv[1]}/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} ms/step={float(r['TotalDurationNs'])/1e6/7:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
