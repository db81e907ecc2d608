#!/bin/bash
# round-6 baseline on one box: headline + real text timings, kernel table, timelines of one step each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-base}
python3 scripts/gpu_one.py enwik 6 2>&1 | tail -4
python3 scripts/gpu_one.py real-text-100MB 4 2>&1 | tail -2
python3 scripts/gpu_one.py enwik:28000000 4 2>&1 | tail -2
bash scripts/quick_prof.sh $TAG
python3 scripts/timeline_step.py gpurun_out/prof_$TAG 2 > gpurun_out/r6_timeline_$TAG.txt
rm -rf gpurun_out/prof_$TAG
rm -rf gpurun_out/prof_real
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_real -- python3 scripts/gpu_one.py real-text-100MB 2 > gpurun_out/real.log 2>&1
python3 scripts/timeline_step.py gpurun_out/prof_real 1 > gpurun_out/r6_timeline_real_$TAG.txt 2>&1 || true
rm -rf gpurun_out/prof_real
