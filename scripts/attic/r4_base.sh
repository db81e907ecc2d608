#!/bin/bash
# baseline of the working copy on one box: GPU tests, kernel table of the headline, phase cycles of chunk_finish, real text
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_tests.log 2>&1; tail -3 gpurun_out/r4_tests.log
bash scripts/quick_prof.sh base
BZH_MSD_DBG=16 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2> gpurun_out/r4_msd_trace.txt | tail -2
tail -30 gpurun_out/r4_msd_trace.txt
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -2
