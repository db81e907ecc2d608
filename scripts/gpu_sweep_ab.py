"""The threshold of the initial SWEEP mode (BZH_SWEEP_DIV) on the workloads that have blocks on the 8 passes."""
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from banzai_amd import _native as nv, corpus
dev = torch.device("cuda", 0)
ctx = nv.Context(0, 9, 128)
sets = [(name, corpus.image_corpus(name)) for name in corpus.IMAGE_SETS] + corpus.c5_parts(100_000_000)
for name, data in sets:
    n = int(data.size)
    if n < 500000: continue
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.from_numpy(np.array(data, dtype=np.uint8, copy=True)).to(dev)
    cap = (n + n // 4 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    best = None
    for it in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        if it: best = dt if best is None or dt < best else best
    print(f"{name:18s} init={os.environ.get('BZH_INIT','auto'):4s} div={os.environ.get('BZH_SWEEP_DIV','8'):3s} {best*1e3:8.2f} ms {n/best/1e6:8.0f} MB/s", flush=True)
'''
for init in (None, "lsd"):
    for div in ("8", "128", "256", "512", "2048"):
        env = dict(os.environ, BZH_SWEEP_DIV=div)
        if init: env["BZH_INIT"] = init
        subprocess.call([sys.executable, "-c", code], env=env)
