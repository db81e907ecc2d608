#!/bin/bash
# A/B on one box: banzai_amd/libbzhip_a.so (the build saved before the change) against the working copy's build
cd $GRAFT_REPO_ROOT
python scripts/gpu_msd_check.py msd 2>&1 | tail -2
for rep in 1 2; do
  BZH_LIB=$PWD/banzai_amd/libbzhip_a.so python3 scripts/gpu_one.py enwik 4 2>&1 | tail -1 | sed 's/^/A /'
  python3 scripts/gpu_one.py enwik 4 2>&1 | tail -1 | sed 's/^/B /'
done
BZH_LIB=$PWD/banzai_amd/libbzhip_a.so python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1 | sed 's/^/A /'
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1 | sed 's/^/B /'
BZH_MSD_DBG=16 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2>&1 | grep -E "ticks" | tail -1
