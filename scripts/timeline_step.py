"""Timeline of one timed step of bench.py in a rocprofv3 kernel trace: python scripts/timeline_step.py DIR [steps-from-the-end]"""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_q'
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = max(glob.glob(os.path.join(d, '*', '*kernel_trace.csv')), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('plan_granules')]
start = idx[-back]
end = idx[-back + 1] if back > 1 else len(rows)
t0 = int(rows[start]['Start_Timestamp'])
pe = t0
for r in rows[start:end]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:30]
    print(f"{(s-t0)/1e3:9.1f} gap {(s-pe)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {name} q={r.get('Queue_Id','')} wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
    pe = max(pe, e)
