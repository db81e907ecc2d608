#!/bin/bash
# A/B of library variants on one box: scripts/ab_libs2.sh KERNEL_SUBSTRING lib1.so lib2.so ... (default library first)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=$1; shift
for L in default "$@"; do
  if [ $L = default ]; then unset BZH_LIB; else export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$L; fi
  rm -rf gpurun_out/prof_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra > gpurun_out/bench_ab.log 2>&1
  python3 - "$L" "$K" <<'PY'
import csv,glob,sys,json
f=glob.glob('gpurun_out/prof_ab/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Name']: print(sys.argv[1], r['Name'][:40], 'avg_us', round(float(r['AverageNs'])/1e3,1))
d=[json.loads(l) for l in open('gpurun_out/bench_ab.log') if l.startswith('{')][-1]
print(sys.argv[1], 'value', d['value'], 'ms', d['ms_per_step'], d['checks'])
PY
done
