#!/bin/bash
cd $GRAFT_REPO_ROOT
for nb in 12 14 20 28 56; do
  for v in 1 0; do
    echo "blocks=$nb BZH_MID=$v $(BZH_MID=$v python3 scripts/gpu_one.py enwik:$((nb * 890000)) 8 2>&1 | tail -1)"
  done
done
