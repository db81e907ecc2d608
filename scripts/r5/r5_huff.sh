#!/bin/bash
# register heap against the oracle (code lengths per table, bits), then timing of both heaps
cd $GRAFT_REPO_ROOT
python scripts/gpu_huff_check.py 2>&1 | tail -7
python -m pytest tests/test_gpu_parity.py -q -x -k "huffman or fixed or golden_streams or model" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for H in reg lds; do
  export BZH_HEAP=$H
  rm -rf gpurun_out/prof_h
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 scripts/gpu_one.py enwik 3 > /dev/null 2>&1
  python3 - $H <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_h/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('huff_build'): print('heap', sys.argv[1], 'huff_build avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
unset BZH_HEAP
rm -rf gpurun_out/prof_h
python3 scripts/gpu_one.py c2 1 2>/dev/null | tail -1
