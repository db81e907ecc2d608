#!/bin/bash
# timing experiments on chunk_finish (results of the encode are WRONG with BZH_MSD_DBG != 0): per-variant kernel time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 5; do
  rm -rf gpurun_out/prof_abl$v
  BZH_MSD_DBG=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abl$v -- python3 scripts/gpu_one.py enwik 3 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob(f'gpurun_out/prof_abl{sys.argv[1]}/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('chunk_finish','bigram_scatter','seg_scatter')):
        print('dbg',sys.argv[1], r['Name'][:30], 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
done
