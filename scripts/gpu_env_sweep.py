"""Every workload of bench.py under several values of one environment variable (own process per value):
python scripts/gpu_env_sweep.py VAR v1 v2 ..."""
import os, subprocess, sys
var, vals = sys.argv[1], sys.argv[2:]
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from banzai_amd import _native as nv, corpus
dev = torch.device("cuda", 0)
ctx = nv.Context(0, 9, 128)
sets = [("v2", corpus.workload(100_000_000)[0]), ("v1", corpus.enwik_synthetic(100_000_000))]
sets += [(name, corpus.image_corpus(name)) for name in corpus.IMAGE_SETS] + corpus.c5_parts(100_000_000)
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
    print(f"{name:18s} %s=%-6s {best*1e3:8.2f} ms {n/best/1e6:8.0f} MB/s" % (os.environ["SWEEP_VAR"], os.environ.get(os.environ["SWEEP_VAR"], "-")), flush=True)
'''
for v in vals:
    env = dict(os.environ, SWEEP_VAR=var)
    if v != "-": env[var] = v
    subprocess.call([sys.executable, "-c", code], env=env)
