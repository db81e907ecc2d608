#!/bin/bash
# quick look at the working copy: stream check, kernel table + timeline of the headline, optional phase ticks
cd $GRAFT_REPO_ROOT
python scripts/gpu_encode_check.py 2>&1 | tail -2
bash scripts/quick_prof.sh try | head -${2:-14}
python3 scripts/timeline_step.py gpurun_out/prof_try > gpurun_out/r4_timeline.txt 2>&1
if [ -n "$1" ]; then BZH_MSD_DBG=$1 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 1 2>&1 | grep -E "ticks|MB/s"; fi
for rep in 1 2; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], d['checks'])"
done
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
