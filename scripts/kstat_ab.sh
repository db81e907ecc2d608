#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the headline encode for each library variant named
# (NAME -> banzai_amd/libbzhip_NAME.so, "cur" = the working copy's build), kernels filtered by a regular expression:
#   scripts/kstat_ab.sh 'seg_|bigram' base cur
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PAT=$1; shift
for v in "$@"; do
  if [ "$v" = cur ]; then unset BZH_LIB; else export BZH_LIB=$PWD/banzai_amd/libbzhip_$v.so; fi
  rm -rf gpurun_out/prof_k
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -- python3 scripts/gpu_one.py ${KSTAT_WORKLOAD:-enwik} 3 > /dev/null 2>&1
  python3 - "$v" "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob('gpurun_out/prof_k/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Name']):
        print(f"{sys.argv[1]:8s} {r['Name'][:44]:44s} calls {r['Calls']:>4s} avg us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
rm -rf gpurun_out/prof_k
