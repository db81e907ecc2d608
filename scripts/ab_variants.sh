#!/bin/bash
# A/B on one box: for every variant library named on the command line (NAME -> banzai_amd/libbzhip_NAME.so; "cur" = the
# working copy's build): bit-exactness against the oracle on a 14-block batch, then timed encodes of the headline, real text, 28 MB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = cur ]; then unset BZH_LIB; else export BZH_LIB=$PWD/banzai_amd/libbzhip_$v.so; fi
  if [ $rep = 1 ]; then python3 scripts/gpu_check.py 2>&1 | tail -1 | sed "s/^/$v /"; fi
  python3 scripts/gpu_one.py enwik 5 2>&1 | tail -2 | sed "s/^/$v /"
  python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1 | sed "s/^/$v /"
  if [ $rep = 1 ]; then python3 scripts/gpu_one.py enwik:28000000 3 2>&1 | tail -1 | sed "s/^/$v /"; fi
done
done
