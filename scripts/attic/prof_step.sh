#!/bin/bash
# rocprofv3 kernel trace of scripts/gpu_one.py NAME (3 encodes); writes the last encode's timeline to gpurun_out/timeline_NAME.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_one_$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one_$1 -- python3 scripts/gpu_one.py $1 3 2>&1 | grep MB/s
python3 scripts/timeline_step.py gpurun_out/prof_one_$1 2 > gpurun_out/timeline_$1.txt
python3 - $1 <<'PY'
import csv,glob,sys
f=glob.glob(f'gpurun_out/prof_one_{sys.argv[1]}/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} ms/enc={float(r['TotalDurationNs'])/1e6/3:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
