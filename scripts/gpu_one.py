"""Encode one named workload a few times on cuda:0 (for rocprofv3 timelines): python scripts/gpu_one.py NAME [reps]
NAME: enwik | enwik:NBYTES | python-sources | shared-libs | c5-zeros | c5-tile1024 | c5-abab | c5-cycling-runs"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if name == "enwik":
    data = corpus.workload(100_000_000)[0]
elif name.startswith("enwik:"): # the first N bytes of the headline text (small batches)
    data = corpus.workload(100_000_000)[0][:int(name[6:])]
elif name in corpus.IMAGE_SETS:
    data = corpus.image_corpus(name)
else:
    data = dict(corpus.c5_parts(100_000_000))[name]
n = int(data.size)
dev = torch.device("cuda", 0)
d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
d_in[:n] = torch.from_numpy(np.array(data)).to(dev)
cap = (n + n // 4 + (1 << 20)) & ~3
d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
ctx = nv.Context(0, 9, 128)
for it in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name}: {n} bytes -> {ln}; {dt*1e3:.2f} ms = {n/dt/1e6:.0f} MB/s")
