#!/bin/bash
# A/B of an environment switch on one box: scripts/r4_ab_env.sh "VAR=value" [kernel substring]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/gpu_encode_check.py 2>&1 | tail -1
for rep in 1 2 3; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], [k['us_per_step'] for k in d['roofline']['kernels'] if '$2' in k['kernel']], d['checks'])"
  env $1 python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], [k['us_per_step'] for k in d['roofline']['kernels'] if '$2' in k['kernel']], d['checks'])"
done
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
env $1 python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
