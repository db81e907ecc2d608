#!/bin/bash
# mid_sort: tile cap sweep (records per tile), headline and real text
cd $GRAFT_REPO_ROOT
for cap in 8192 6144 5120 4096 3072; do
  export BZH_MID_CAP=$cap
  echo "cap=$cap $(python3 scripts/gpu_one.py enwik 6 2>&1 | tail -1) | $(python3 scripts/gpu_one.py real-text-100MB 4 2>&1 | tail -1)"
done
