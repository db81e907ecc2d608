"""First N launches of the named kernels in the LAST BWT step of a rocprofv3 kernel trace: python scripts/timeline_kern.py DIR name1,name2 [N]"""
import csv, glob, os, sys
d = sys.argv[1]
names = sys.argv[2].split(',')
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = max(glob.glob(os.path.join(d, '*', '*kernel_trace.csv')), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('byte_count')]
start = idx[-1]
seen = {}
for r in rows[start:]:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')
    for want in names:
        if nm.startswith(want):
            seen.setdefault(nm, [])
            if len(seen[nm]) < N:
                seen[nm].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in seen.items():
    print(k, ' '.join(f'{x:8.1f}' for x in v))
