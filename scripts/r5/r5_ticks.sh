#!/bin/bash
# chunk_finish phase ticks with and without the fused first doubling step; then plain timings of both
cd $GRAFT_REPO_ROOT
BZH_MSD_DBG=16 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2>&1 | grep -E "ticks" | tail -1
BZH_R0=0 BZH_MSD_DBG=16 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2>&1 | grep -E "ticks" | tail -1
python3 scripts/gpu_one.py enwik 4 2>&1 | tail -2
BZH_R0=0 python3 scripts/gpu_one.py enwik 4 2>&1 | tail -2
