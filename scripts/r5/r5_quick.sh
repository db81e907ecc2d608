#!/bin/bash
# quickest look: both initial sorts against the oracle, phase ticks, 4 timed encodes of the headline and of real text
cd $GRAFT_REPO_ROOT
python scripts/gpu_msd_check.py msd 2>&1 | tail -2
BZH_MSD_DBG=16 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2>&1 | grep -E "ticks" | tail -1
python3 scripts/gpu_one.py enwik 5 2>&1 | tail -3
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -2
