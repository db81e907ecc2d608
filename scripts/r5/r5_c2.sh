#!/bin/bash
# config 2 (one random 899,999-byte block) and a 31-block text batch: wall times + timeline of config 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/c2.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from banzai_amd import _native as nv, corpus
which = sys.argv[1]
data = corpus.xorshift_bytes(899_999) if which == "c2" else corpus.workload(100_000_000)[0][:int(which)]
n = int(data.size); dev = torch.device("cuda", 0)
d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.from_numpy(np.array(data)).to(dev)
cap = (n + n // 4 + (1 << 20)) & ~3; d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
ctx = nv.Context(0, 9, 128)
best = 1e9
for it in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
print(which, n, "bytes ->", ln, "best ms", round(best * 1e3, 3))
PY
python3 /tmp/c2.py c2; python3 /tmp/c2.py 27900000; python3 /tmp/c2.py 7000000
rm -rf gpurun_out/prof_c2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c2 -- python3 /tmp/c2.py c2 > /dev/null 2>&1
python3 - <<'PY' > gpurun_out/r5_timeline_c2.txt
import csv, glob, os
f = max(glob.glob('gpurun_out/prof_c2/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('plan_starts')]
start = idx[-1]; t0 = int(rows[start]['Start_Timestamp']); pe = t0
for r in rows[start:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} gap {(s-pe)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {r['Kernel_Name'].split('(')[0].replace('void ','')[:34]} wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
    pe = max(pe, e)
PY
rm -rf gpurun_out/prof_c2
