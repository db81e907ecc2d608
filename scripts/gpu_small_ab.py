"""Small batches (1, 2, 4, 8, 16 blocks of the headline text; one random block) with both initial sorts."""
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from banzai_amd import _native as nv, corpus
dev = torch.device("cuda", 0)
ctx = nv.Context(0, 9, 128)
text = corpus.workload(32_000_000)[0]
sets = [(f"text x{k}", text[:k * 890_000]) for k in [int(x) for x in os.environ.get("BZH_AB_BLOCKS", "1,2,4,8,16").split(",")]] + [("random x1", corpus.xorshift_bytes(899_999))]
for name, data in sets:
    n = int(data.size)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.from_numpy(np.array(data, dtype=np.uint8, copy=True)).to(dev)
    cap = (n + n // 4 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    best = None
    for it in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        if it: best = dt if best is None or dt < best else best
    print(f"{name:10s} init={os.environ.get('BZH_INIT','auto'):4s} {best*1e3:8.3f} ms {n/best/1e6:8.0f} MB/s", flush=True)
'''
for init in (None, "lsd"):
    env = dict(os.environ)
    if init: env["BZH_INIT"] = init
    subprocess.call([sys.executable, "-c", code], env=env)
