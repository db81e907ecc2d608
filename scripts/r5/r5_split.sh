#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/gpu_encode_check.py 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "rle1 or split or stream or golden or fuzz or corpus or sharded or shard or facade or beyond" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 scripts/gpu_one.py enwik 3 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_h/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('plan_'): print(r['Name'][:40], 'avg us', round(float(r['AverageNs'])/1e3,1), 'calls', r['Calls'])
PY
rm -rf gpurun_out/prof_h
python3 scripts/gpu_one.py enwik 4 2>&1 | tail -2
for w in c5-zeros c5-cycling-runs shared-libs; do python3 scripts/gpu_one.py $w 2 2>&1 | tail -1; done
