#!/bin/bash
# several library variants on one box: scripts/r4_ab_libs.sh "kernel,substrings" lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=$1; shift
for L in "$@"; do BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$L python scripts/gpu_encode_check.py 2>&1 | tail -1; done
for rep in 1 2; do
  for L in default "$@"; do
    if [ $L = default ]; then unset BZH_LIB; else export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$L; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['stage_ms_per_step'], [(k['kernel'][:14],k['us_per_step']) for k in d['roofline']['kernels'] if any(x in k['kernel'] for x in '$K'.split(','))], d['checks'])"
  done
done
unset BZH_LIB
for L in default "$@"; do
  if [ $L = default ]; then unset BZH_LIB; else export BZH_LIB=$GRAFT_REPO_ROOT/banzai_amd/$L; fi
  python3 scripts/gpu_one.py real-text-100MB 3 2>&1 | tail -1
done
