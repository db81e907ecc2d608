#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_real
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_real -- python3 scripts/gpu_one.py real-text-100MB 2 > gpurun_out/real.log 2>&1
python3 - <<'PY' > gpurun_out/r5_timeline_real.txt
import csv, glob, os
f = max(glob.glob('gpurun_out/prof_real/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('plan_starts')]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp']); pe = t0
for r in rows[start:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:30]
    print(f"{(s-t0)/1e3:9.1f} gap {(s-pe)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {name} q={r.get('Queue_Id','')} wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
    pe = max(pe, e)
PY
rm -rf gpurun_out/prof_real
tail -3 gpurun_out/real.log
