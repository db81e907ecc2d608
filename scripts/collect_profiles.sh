#!/bin/bash
# Collects the round's measurement artefacts ON THE GPU BOX into gpurun_out/profiles_<tag>/ (copy what is to be judged
# into profiles/):   scripts/collect_profiles.sh r03
#   <tag>_bench_n1.json                 the default bench line
#   <tag>_bench_n1_under_rocprof.json   the line of the same command under rocprofv3 --kernel-trace --stats
#   <tag>_kernel_stats_bench_n1.csv     rocprofv3's kernel summary of that run
#   <tag>_pmc_traffic.json              FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, no tracing domains)
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/bench.err"
# BZH_NO_OVERLAP=1: the product runs the big-list passes on a second stream beside tail_round; a trace of that pass shows
# kernel durations stretched by the sharing.  The roofline figure is defined on the serialized pass (HIP events, profiling
# on), so the trace that has to agree with it is taken serialized as well.  (Exported before the profiler starts: the
# program after `--` must be the interpreter itself.)
export BZH_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats -d "$OUT/prof" --output-format csv -- python3 bench.py --no-extra > "$OUT/${TAG}_bench_n1_under_rocprof.json" 2> "$OUT/prof.err"
unset BZH_NO_OVERLAP
cp "$(ls "$OUT"/prof/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats_bench_n1.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extra > "$OUT/pmc_$c.log" 2>&1
done
python3 scripts/pmc_summary.py "$OUT" "$TAG" > "$OUT/${TAG}_pmc_traffic.json"
rm -rf "$OUT/prof" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
ls -la "$OUT"
