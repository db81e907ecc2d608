#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/gpu_mtf_check.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mtf or stream_bit_exact or golden or corpus" 2>&1 | tail -2
bash scripts/r5/r5_ab.sh 2>&1 | grep -v "ticks\|initial"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 scripts/gpu_one.py enwik 3 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_h/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('mtf_'): print(r['Name'][:40], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
rm -rf gpurun_out/prof_h
