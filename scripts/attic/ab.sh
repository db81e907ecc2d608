#!/bin/bash
# A/B timing on ONE box: A = library built from git HEAD's csrc (gpurun_out is scratch, so the
# caller builds it first: scripts/ab_build.sh), B = the working copy's banzai_amd/libbzhip.so.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/libbzhip_A.so; else unset BZH_LIB; fi
    python3 bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], d['checks'])"
  done
done
