#!/bin/bash
# Runs on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 2 > gpurun_out/bench_final.log 2>&1
tail -1 gpurun_out/bench_final.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_final gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-extra > gpurun_out/bench_final_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extra > gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f'gpurun_out/pmc_{c}/*/*counter_collection.csv'))[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c: continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    res[c] = agg
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --no-cpu --no-extra` (3 passes of the hot path: warm-up, timed, profiled); "
               "values in KB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads half of wide streaming reads, "
               "MI355X_MICROARCH.md HBM section; 8-byte-per-lane accesses are uncalibrated)", "kernels": {}}
names = set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])
for k in names:
    f = res["FETCH_SIZE"].get(k, [0, 0.0]); w = res["WRITE_SIZE"].get(k, [0, 0.0])
    n = max(f[0], w[0])
    out["kernels"][k] = {"launches": n, "fetch_kb": f[1], "write_kb": w[1],
                         "hbm_bytes_per_launch": (2 * f[1] + w[1]) * 1024 / max(1, n)}
sc = [v for k, v in out["kernels"].items() if k.startswith("radix_scatter")]
tot_launch = sum(v["launches"] for v in sc)
out["radix_scatter_all"] = {"launches": tot_launch,
                            "hbm_bytes_per_launch": sum((2 * v["fetch_kb"] + v["write_kb"]) * 1024 for v in sc) / max(1, tot_launch)}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["radix_scatter_all"]))
PY
