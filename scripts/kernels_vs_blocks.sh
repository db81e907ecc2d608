#!/bin/bash
# per-kernel time per encode against the number of blocks in the batch (the same text, cut shorter): where are the floors?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nb in 14 28 56 112; do
  w=enwik:$((nb * 890000))
  rm -rf gpurun_out/prof_h
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 scripts/gpu_one.py $w 4 > /dev/null 2>&1
  cp gpurun_out/prof_h/*/*kernel_stats.csv gpurun_out/scale_$nb.csv
done
rm -rf gpurun_out/prof_h
python3 - <<'PY'
import csv
tabs = {}
for nb in (14, 28, 56, 112):
    for r in csv.DictReader(open(f'gpurun_out/scale_{nb}.csv')):
        k = r['Name'].split('(')[0].replace('void ', '')
        tabs.setdefault(k, {})[nb] = float(r['TotalDurationNs']) / 4 / 1e3
rows = sorted(tabs.items(), key=lambda kv: -kv[1].get(112, 0))
print(f"{'kernel':34s} {'14':>8s} {'28':>8s} {'56':>8s} {'112':>8s}  ratio112/14")
for k, v in rows[:40]:
    a = v.get(14, 0); d = v.get(112, 0)
    print(f"{k[:34]:34s} {v.get(14,0):8.1f} {v.get(28,0):8.1f} {v.get(56,0):8.1f} {v.get(112,0):8.1f}  {d/a if a else 0:5.1f}")
PY
