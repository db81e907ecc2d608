#!/bin/bash
# timeline of one step of the headline (and optionally another workload) with the working copy's build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-enwik}; TAG=${2:-cur}
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 scripts/gpu_one.py $W 3 > gpurun_out/tl.log 2>&1
python3 scripts/timeline_step.py gpurun_out/prof_tl 1 > gpurun_out/r6_timeline_${TAG}.txt 2>&1
rm -rf gpurun_out/prof_tl
tail -2 gpurun_out/tl.log
