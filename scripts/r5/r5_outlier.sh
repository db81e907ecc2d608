#!/bin/bash
# how often does an encode take much longer than its neighbours?  30 encodes of real text and of the headline, both ways
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  for w in real-text-100MB enwik; do
    echo "BZH_MID=$v $w: $(BZH_MID=$v BZH_TRACE_ROUNDS= python3 scripts/gpu_one.py $w 30 2>&1 | grep -o '[0-9.]* ms' | tr '\n' ' ')"
  done
done
