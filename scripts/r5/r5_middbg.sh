#!/bin/bash
# timing experiments on mid_sort: kernel time with parts left out (results are wrong then)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 7; do
  export BZH_MID_DBG=$d
  rm -rf gpurun_out/prof_md
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_md -- python3 scripts/gpu_one.py enwik 3 > /dev/null 2>&1
  f=$(ls gpurun_out/prof_md/*/*kernel_stats.csv | head -1)
  echo "dbg=$d $(grep -E 'mid_sort' $f | cut -d, -f1-4)"
done
rm -rf gpurun_out/prof_md
