import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus
data = corpus.enwik_synthetic(100_000_000)
n = int(data.size)
dev = torch.device("cuda", 0)
d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.from_numpy(np.array(data)).to(dev)
cap = (n + n // 4 + (1 << 20)) & ~3
d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
ctx = nv.Context(0, 9, 128)
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"v1 {os.environ.get('BZH_INIT','default')}: {dt*1e3:.2f} ms = {n/dt/1e6:.0f} MB/s")
