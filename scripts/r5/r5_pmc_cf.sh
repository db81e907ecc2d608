#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of chunk_finish for the working copy's build (B) and libbzhip_a.so (A); separate --pmc passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in A B; do
  if [ $L = A ]; then export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/libbzhip_a.so; else unset BZH_LIB; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf gpurun_out/pmc_cf
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_cf -- python3 scripts/gpu_one.py enwik 2 > /dev/null 2>&1
    python3 - $L $c <<'PY'
import csv, glob, os, sys
f = max(glob.glob('gpurun_out/pmc_cf/*/*counter_collection.csv'), key=os.path.getsize)
tot = {}; n = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if r['Counter_Name'] == sys.argv[2]:
        tot[k] = tot.get(k, 0) + float(r['Counter_Value']); n[k] = n.get(k, 0) + 1
for k in ('chunk_finish', 'rank_apply', 'bigram_scatter', 'tail_round<false>', 'active_gen', 'refine_one<false>'):
    if k in tot: print(sys.argv[1], sys.argv[2], k, 'MB per launch', round(tot[k] / n[k] / 1024, 1), 'launches', n[k])
PY
  done
done
rm -rf gpurun_out/pmc_cf
