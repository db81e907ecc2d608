"""Timeline of the last BWT step in a rocprofv3 kernel trace (newest *kernel_trace.csv under the given directory)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_q'
f = max(glob.glob(os.path.join(d, '*', '*kernel_trace.csv')), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('byte_count')]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r in rows[start:start + 400]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
    print(f"{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {name}  wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
    prev_end = e
    if name.startswith('bwt_emit'):
        break
