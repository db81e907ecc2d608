#!/bin/bash
# quick GPU loop: parity on the stream checks + bench + kernel stats
cd $GRAFT_REPO_ROOT
python scripts/gpu_encode_check.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra > gpurun_out/bench_q.log 2>&1
grep '^{' gpurun_out/bench_q.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('VALUE',d['value'],'ms',d['ms_per_step'],d['roofline']['achieved'],d['stage_ms_per_step'],d['checks'])"
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_q/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6/7:8.2f}/step avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
