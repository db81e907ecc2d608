#!/bin/bash
# A/B timing of library variants on ONE box: scripts/ab_libs.sh libA.so libB.so ...  (paths under banzai_amd/; "-" = the default build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset BZH_LIB; else export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$v; fi
    python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], d['bwt_rounds'], d['checks'])"
    for w in python-sources shared-libs c5-tile1024; do python3 scripts/gpu_one.py $w 3 2>/dev/null | tail -1; done
  done
done
