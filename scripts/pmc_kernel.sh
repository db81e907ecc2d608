#!/bin/bash
# usage: pmc_kernel.sh "<COUNTER ...>" <kernel-name-substring>   (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_k
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/pmc_k -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_k.log 2>&1
python3 - "$2" <<'PY'
import csv, glob, collections, sys, os
f = max(glob.glob('gpurun_out/pmc_k/*/*counter_collection.csv'), key=os.path.getsize)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if sys.argv[1] in k:
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k, {c: (round(v), n[(k, c)]) for c, v in d.items()})
PY
