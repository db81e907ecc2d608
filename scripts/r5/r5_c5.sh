#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "periodic or config5 or c5 or pathological or other_workloads" 2>&1 | tail -3
for w in c5-tile1024 c5-abab c5-zeros c5-cycling-runs; do python3 scripts/gpu_one.py $w 3 2>&1 | tail -1; done
python scripts/gpu_fuzz.py 60 77 gpurun_out/fz.json | tail -1 | cut -c1-200
