#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/gpu_msd_check.py msd 2>&1 | tail -2
python scripts/gpu_msd_check.py lsd 2>&1 | tail -2
python scripts/gpu_encode_check.py 2>&1 | tail -1
python scripts/gpu_c5.py 2>&1 | tail -6
python scripts/gpu_v1.py 2>&1 | tail -2
python3 scripts/gpu_one.py enwik 5 2>&1 | tail -1
python3 scripts/gpu_one.py real-text-100MB 4 2>&1 | tail -1
python3 scripts/gpu_one.py enwik:899999 5 2>&1 | tail -1
