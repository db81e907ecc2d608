#!/bin/bash
# working copy on one box: GPU tests, kernel table of the headline, then the A/B switch given as $1 (an env assignment)
cd $GRAFT_REPO_ROOT
AB=${1:-BZH_NO_DESC=1}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_tests.log 2>&1; tail -5 gpurun_out/r4_tests.log
bash scripts/quick_prof.sh try
python3 scripts/timeline_step.py gpurun_out/prof_try > gpurun_out/r4_timeline.txt 2>&1
for rep in 1 2; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], d['checks'])"
  env $AB python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$AB', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], d['checks'])"
done
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -2
env $AB python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
