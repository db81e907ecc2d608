#!/bin/bash
# Collects the round's measurement artefacts ON THE GPU BOX into gpurun_out/profiles_<tag>/ (copy what is to be judged
# into profiles/):   scripts/collect_profiles.sh r03
#   <tag>_bench_n1.json                 the default bench line
#   <tag>_bench_n1_under_rocprof.json   the line of the same command under rocprofv3 --kernel-trace --stats
#   <tag>_kernel_stats_bench_n1.csv     rocprofv3's kernel summary of that run
#   <tag>_pmc_traffic.json              FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes, no tracing domains)
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
# the PMC passes first: the bench line names the newest PMC file under profiles/ as the source of `roofline.traffic`, and that
# has to be THIS round's (the file is written into the box's copy of profiles/ as well as into $OUT)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extra > "$OUT/pmc_$c.log" 2>&1
done
python3 scripts/pmc_summary.py "$OUT" "$TAG" > "$OUT/${TAG}_pmc_traffic.json"
cp "$OUT/${TAG}_pmc_traffic.json" profiles/
python3 bench.py > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/bench.err"
# BZH_NO_OVERLAP=1: the product runs the big-list passes on a second stream beside tail_round; a trace of that pass shows
# kernel durations stretched by the sharing.  The roofline figure is defined on the serialized pass (HIP events, profiling
# on), so the trace that has to agree with it is taken serialized as well.  (Exported before the profiler starts: the
# program after `--` must be the interpreter itself.)
export BZH_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats -d "$OUT/prof" --output-format csv -- python3 bench.py --no-extra > "$OUT/${TAG}_bench_n1_under_rocprof.json" 2> "$OUT/prof.err"
unset BZH_NO_OVERLAP
cp "$(ls "$OUT"/prof/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats_bench_n1.csv"
rm -rf "$OUT/prof" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
# ---- round 4: the bucket-first initial sort (default for text since round 4) next to the 8 radix passes it replaces
# (BZH_INIT=lsd; same box, same command), the kernel table of the 8-pass run and the cycles per phase of chunk_finish;
# the CLI's wall clock; the streaming API's trigger sweep; two half-batch lanes against one
python3 bench.py --no-extra --no-cpu --steps 10 > "$OUT/msd_msd_line.json" 2>/dev/null
BZH_INIT=lsd python3 bench.py --no-extra --no-cpu --steps 10 > "$OUT/msd_lsd_line.json" 2>/dev/null
export BZH_INIT=lsd BZH_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats -d "$OUT/prof_lsd" --output-format csv -- python3 bench.py --no-extra --no-cpu --steps 3 > /dev/null 2> "$OUT/prof_lsd.err"
unset BZH_INIT BZH_NO_OVERLAP
cp "$(ls "$OUT"/prof_lsd/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_kernel_stats_lsd.csv"
rm -rf "$OUT/prof_lsd"
BZH_MSD_DBG=16 BZH_TRACE_ROUNDS=1 python3 scripts/gpu_one.py enwik 2 2> "$OUT/msd_trace.txt" > /dev/null
python3 scripts/msd_ablation.py "$OUT" "$TAG" > "$OUT/${TAG}_msd_ablation.json"
python3 scripts/cli_wallclock.py "$OUT/${TAG}_cli_wallclock.json" > /dev/null 2>&1
python3 scripts/gpu_stream_sweep.py > "$OUT/${TAG}_stream_api_sweep.txt" 2>&1
python3 scripts/gpu_lanes.py > "$OUT/${TAG}_lanes.txt" 2>&1
# ---- round 6: what the host<->device copies of the API paths cost on this box (the floor of value_stream_api /
# value_host_inclusive: DESIGN section 6); the Rust facade's calling pattern with its context pool, first call against
# steady state (tests/abi_facade.c, 16 MiB slices = encode_file, 100 MB, output to /dev/null); the devices-behind-one-handle
# flow on this box's one GPU (three contexts, 3 x 30 MB)
if [ ! -x scripts/micro/copy_rates ]; then (cd scripts/micro && hipcc -O2 --offload-arch=gfx950 -o copy_rates copy_rates.hip -lpthread > /dev/null 2>&1); fi
./scripts/micro/copy_rates > "$OUT/${TAG}_copy_rates.txt" 2>&1
gcc -O2 -Wall -o /tmp/abi_facade tests/abi_facade.c -Lbanzai_amd -lbzhip -Wl,-rpath,$PWD/banzai_amd
python3 -c "
import sys; sys.path.insert(0, '.')
from banzai_amd import corpus
corpus.workload(100_000_000)[0].tofile('/tmp/facade_in.bin')"
{ echo "# tests/abi_facade.c = the calling pattern of rust/src/lib.rs (context pool, pooled buffers); 100 MB of the bench workload, output to /dev/null"
  echo "# in-memory reader (fill_buf = everything that is left: what value_stream_api feeds from), 6 calls in one process:"
  /tmp/abi_facade 9 0 /tmp/facade_in.bin /dev/null 6
  echo "# encode_file's reader (16 MiB BufReader slices from the page cache), 6 calls:"
  /tmp/abi_facade 9 16777216 /tmp/facade_in.bin /dev/null 6; } > "$OUT/${TAG}_facade_calls.txt" 2>&1
rm -f /tmp/facade_in.bin
python3 bench.py --gpus 3 --single-process --devices 0,0,0 --bytes 30000000 --steps 5 > "$OUT/${TAG}_single_process_3x30MB.json" 2>/dev/null
rm -f "$OUT"/pmc_*.log "$OUT"/*.err
ls -la "$OUT"
