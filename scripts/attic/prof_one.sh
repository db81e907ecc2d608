#!/bin/bash
# rocprofv3 kernel trace of scripts/gpu_one.py NAME -> gpurun_out/prof_one_NAME ; prints the last step's timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_one_$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one_$1 -- python3 scripts/gpu_one.py $1 3 2>&1 | grep MB/s
python3 scripts/timeline.py gpurun_out/prof_one_$1 > gpurun_out/timeline_$1.txt
