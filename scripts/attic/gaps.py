"""Idle time between kernels of the last bench step in a rocprofv3 kernel trace (newest trace in the directory)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [k for k, r in enumerate(rows) if r['Kernel_Name'].startswith('plan_starts')]
rows = rows[idx[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end, prev_name, gaps, busy, last = None, '', [], 0, t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if r['Kernel_Name'].startswith('__amd_rocclr_copyBuffer') and e - s > 300_000:
        break  # D2H of the finished stream: after the step
    if prev_end is not None and s > prev_end:
        gaps.append((s - prev_end, prev_name, r['Kernel_Name'][:36], (s - t0) / 1e3))
    busy += e - s
    prev_end = max(prev_end or 0, e)
    prev_name = r['Kernel_Name'][:36]
    last = e
print(f"span {(last - t0) / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  gaps {sum(g[0] for g in gaps) / 1e3:.0f} us in {len(gaps)}")
gaps.sort(reverse=True)
for g in gaps[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{g[0] / 1e3:8.1f}us after {g[1]:36s} before {g[2]:36s} at {g[3]:.0f}us")
