#!/bin/bash
# Coverage-guided fuzzing of the CPU oracle + strict decoder (oracle/covfuzz.c) on every core, then greedy minimisation:
#   scripts/covfuzz.sh SECONDS [WORKDIR]   ->  WORKDIR/min/*.bin ; pack with scripts/covfuzz_pack.py into tests/golden/fuzz_corpus.zip
set -e
cd "$(dirname "$0")/.."
SEC=${1:-300}; W=${2:-/tmp/covfuzz}
make -s -C oracle covfuzz
NP=$(nproc); mkdir -p "$W/all"
for k in $(seq 1 $NP); do mkdir -p "$W/c$k"; ./oracle/_san/covfuzz run "$W/c$k" "$SEC" "$k" > "$W/c$k.log" 2>&1 & done
wait
cat "$W"/c*.log
for k in $(seq 1 $NP); do cp -n "$W/c$k"/*.bin "$W/all/" 2>/dev/null || true; done
rm -rf "$W/min"; ./oracle/_san/covfuzz min "$W/all" "$W/min"
