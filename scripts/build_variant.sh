#!/bin/bash
# builds banzai_amd/libbzhip_<name>.so from the working tree with extra compiler flags (A/B runs on one box: BZH_LIB=...)
#   scripts/build_variant.sh NAME "-DFOO=1 ..."
set -e
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BD=/tmp/bzh_build_$NAME
mkdir -p $BD
cd $ROOT/banzai_amd/csrc
pids=()
for f in api bwt mtf huffman rle1 multi; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -fvisibility=hidden -fno-gpu-rdc -DBZH_BUILD $FLAGS -c $f.hip -o $BD/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/banzai_amd/libbzhip_$NAME.so $BD/api.o $BD/bwt.o $BD/mtf.o $BD/huffman.o $BD/rle1.o $BD/multi.o
ls -la $ROOT/banzai_amd/libbzhip_$NAME.so
