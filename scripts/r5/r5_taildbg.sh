#!/bin/bash
# timing experiments on tail_round (wrong results on purpose): duration of the FIRST non-empty tail_round<false> launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in enwik real-text-100MB; do
for D in 0 1 2 4 3 7; do
  export BZH_TAIL_DBG=$D
  rm -rf gpurun_out/prof_dbg
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_dbg -- python3 scripts/gpu_one.py $W 1 > /dev/null 2>&1
  python3 - $W $D <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_dbg/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('void tail_round<false>')]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
big=[x for x in d if x>30]
print(sys.argv[1],'DBG',sys.argv[2],'first real tail_round<false> us', big[0] if big else None, 'all', [round(x) for x in d[:6]])
PY
done
done
rm -rf gpurun_out/prof_dbg
