#!/bin/bash
# timing experiments on chunk_finish (wrong results on purpose): BZH_MSD_DBG 32 = keys without the text gather, 64 = no all-pairs loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 0 32 64 96; do
  export BZH_MSD_DBG=$D
  rm -rf gpurun_out/prof_dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -- python3 scripts/gpu_one.py enwik 3 > /dev/null 2>&1
  python3 - $D <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_dbg/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('chunk_finish'): print('DBG', sys.argv[1], 'chunk_finish avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done
rm -rf gpurun_out/prof_dbg
