#!/bin/bash
# round 5: mid_sort (round 0's large groups in LDS) against the global passes (BZH_MID=0): correctness, then A/B on one box
cd $GRAFT_REPO_ROOT
python scripts/gpu_msd_check.py msd 2>&1 | tail -3
python scripts/gpu_encode_check.py 2>&1 | tail -2
for rep in 1 2; do
  for v in 1 0; do
    echo "BZH_MID=$v"
    BZH_MID=$v python3 scripts/gpu_one.py enwik 6 2>&1 | tail -1
    BZH_MID=$v python3 scripts/gpu_one.py real-text-100MB 4 2>&1 | tail -1
  done
done
BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 1 2>&1 | grep -E "round 0|round 1|initial" | head -4
