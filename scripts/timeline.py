import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)  # newest trace
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('plan_starts')]
seq = rows[idx[-1]:]
t0 = int(seq[0]['Start_Timestamp'])
tot = {}
for r in seq:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:28]
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    st = (int(r['Start_Timestamp']) - t0) / 1e3
    tot[name] = tot.get(name, 0) + dur
    if dur > thr:
        print(f"{st:10.1f}us {name:30s} {dur:9.1f}us grid={r['Grid_Size_X']}")
print("---- totals (last step)")
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:14]:
    print(f"{k:30s} {v/1e3:8.2f} ms")
