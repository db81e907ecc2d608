#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/gpu_msd_check.py lsd 2>&1 | tail -2
python scripts/gpu_msd_check.py msd 2>&1 | tail -1
python scripts/gpu_encode_check.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "periodic or bwt or other_workloads or initial_sorts or lookback" 2>&1 | tail -3
for w in c5-tile1024 c5-abab c5-zeros c5-cycling-runs; do python3 scripts/gpu_one.py $w 3 2>&1 | tail -1; done
python3 /tmp/c2.py c2 2>/dev/null || true
python scripts/gpu_fuzz.py 90 78 gpurun_out/fz.json | tail -1 | cut -c1-200
python scripts/gpu_fuzz_batches.py 120 79 gpurun_out/fzb.json | tail -1 | cut -c1-160
