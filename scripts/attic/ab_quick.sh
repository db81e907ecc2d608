#!/bin/bash
# A/B of library variants on the bench workload only: scripts/ab_quick.sh libA.so libB.so ... ("-" = default build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset BZH_LIB; else export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$v; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['stage_ms_per_step'], d['checks'])"
  done
done
