#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/pmc_$c.log 2>&1
  ls gpurun_out/pmc_$c/*/ | head
done
python3 - <<'PY'
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f'gpurun_out/pmc_{c}/*/*counter_collection.csv')
    if not f:
        print("no counter file for", c, glob.glob(f'gpurun_out/pmc_{c}/*/*')); continue
    rows = list(csv.DictReader(open(f[0])))
    if c == "FETCH_SIZE": print(rows[0].keys())
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r['Counter_Name'] != c: continue
        k = r['Kernel_Name'].split('(')[0]
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    res[c] = agg
names = sorted(set(res.get("FETCH_SIZE", {})) | set(res.get("WRITE_SIZE", {})), key=lambda k: -(res["FETCH_SIZE"].get(k,[0,0])[1] + res["WRITE_SIZE"].get(k,[0,0])[1]))
print(f"{'kernel':44s} {'launches':>8s} {'FETCH_SIZE(KB)':>16s} {'WRITE_SIZE(KB)':>16s}")
for k in names[:22]:
    f = res["FETCH_SIZE"].get(k, [0, 0]); w = res["WRITE_SIZE"].get(k, [0, 0])
    print(f"{k[:44]:44s} {f[0]:8d} {f[1]:16.0f} {w[1]:16.0f}")
PY
