#!/bin/bash
# huff_build's time against the number of blocks in the batch (the same text, cut shorter)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nb in 1 2 4 8 16 32 64 112; do
  w=enwik:$((nb * 890000))
  rm -rf gpurun_out/prof_h
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 scripts/gpu_one.py $w 3 > /dev/null 2>&1
  python3 - $nb <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_h/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('huff_build'): print('blocks', sys.argv[1], 'huff_build avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
rm -rf gpurun_out/prof_h
