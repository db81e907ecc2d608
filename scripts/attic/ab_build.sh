#!/bin/bash
# Builds banzai_amd/libbzhip_A.so from the csrc of a git revision (default HEAD) for scripts/ab.sh.
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src/banzai_amd /tmp/ab_src/include
git archive $rev banzai_amd/csrc include | tar -x -C /tmp/ab_src
make -s -C /tmp/ab_src/banzai_amd/csrc -j4 ../libbzhip.so
cp /tmp/ab_src/banzai_amd/libbzhip.so banzai_amd/libbzhip_A.so
echo "A = $(git rev-parse --short $rev)"
