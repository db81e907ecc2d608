#!/bin/bash
# round 5 baseline / quick look at the working copy on one box: stream check, kernel table + timeline of the headline,
# per-round trace of the headline and of real text.   scripts/r5/r5_base.sh [tag]
cd $GRAFT_REPO_ROOT
TAG=${1:-base}
python scripts/gpu_encode_check.py 2>&1 | tail -2
bash scripts/quick_prof.sh $TAG | head -40
python3 scripts/timeline_step.py gpurun_out/prof_$TAG > gpurun_out/r5_timeline_$TAG.txt 2>&1
BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2> gpurun_out/r5_trace_enwik_$TAG.txt | tail -1
BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py real-text-100MB 2 2> gpurun_out/r5_trace_real_$TAG.txt | tail -1
for rep in 1 2; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['checks'])"
done
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
rm -rf gpurun_out/prof_$TAG/*/*.db
