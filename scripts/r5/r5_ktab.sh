#!/bin/bash
# kernel table of one workload (default enwik) under rocprofv3: scripts/r5/r5_ktab.sh [workload] [rows]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-enwik}
rm -rf gpurun_out/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 scripts/gpu_one.py $W 3 > gpurun_out/ktab.log 2>&1
tail -1 gpurun_out/ktab.log
python3 - ${2:-24} <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_k/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[1])]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} ms/encode={float(r['TotalDurationNs'])/1e6/3:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
python3 scripts/timeline_step.py gpurun_out/prof_k 2 > gpurun_out/r5_timeline_k.txt 2>&1
rm -rf gpurun_out/prof_k
