#!/bin/bash
# A/B of library variants on ONE box: headline (bench, 8 steps) + real-text-100MB:  scripts/ab_quick2.sh libA.so - libB.so
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset BZH_LIB; else export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$v; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'headline ms', d['ms_per_step'], 'bwt', d['stage_ms_per_step']['ms_bwt'], 'rounds', d['bwt_rounds'], d['checks'])"
    python3 scripts/gpu_one.py real-text-100MB 4 2>/dev/null | tail -1 | sed "s/^/$v /"
  done
done
