#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "lookback" --durations=5 2>&1 | tail -9
python scripts/gpu_three_procs.py 100 gpurun_out/r05_three_processes.json | tail -1
