#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the LARGEST launch of refine_one<false>, mid_sort, tail_round (round 0/1); separate --pmc passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf gpurun_out/pmc_cf
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_cf -- python3 scripts/gpu_one.py enwik 1 > /dev/null 2>&1
  python3 - $c <<'PY'
import csv, glob, os, sys
f = max(glob.glob('gpurun_out/pmc_cf/*/*counter_collection.csv'), key=os.path.getsize)
mx = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if r['Counter_Name'] == sys.argv[1]:
        d = mx.setdefault(k, {})
        key = r.get('Dispatch_Id') or r.get('Correlation_Id')
        d[key] = d.get(key, 0) + float(r['Counter_Value'])
for k in ('refine_one<false>', 'mid_sort', 'tail_round<false>', 'active_gen', 'bwt_emit', 'rank_apply'):
    if k in mx: print(sys.argv[1], k, 'MB of the largest launch', round(max(mx[k].values()) / 1024, 1), 'launches', len(mx[k]))
PY
done
rm -rf gpurun_out/pmc_cf
