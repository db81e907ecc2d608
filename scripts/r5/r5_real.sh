#!/bin/bash
# kernel table of real text (100 MB) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_real
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_real -- python3 scripts/gpu_one.py real-text-100MB 3 > gpurun_out/real.log 2>&1
tail -1 gpurun_out/real.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_real/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} ms/encode={float(r['TotalDurationNs'])/1e6/3:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
rm -rf gpurun_out/prof_real
