#!/bin/bash
# A/B of a library variant on one box: scripts/r4_ab_lib.sh libbzhip_X.so [kernel substring]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$1 python scripts/gpu_encode_check.py 2>&1 | tail -1
for rep in 1 2 3; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], [(k['kernel'][:14],k['us_per_step']) for k in d['roofline']['kernels'] if any(x in k['kernel'] for x in '$2'.split(','))], d['checks'])"
  BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$1 python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['stage_ms_per_step']['ms_bwt'], [(k['kernel'][:14],k['us_per_step']) for k in d['roofline']['kernels'] if any(x in k['kernel'] for x in '$2'.split(','))], d['checks'])"
done
python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$1 python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
