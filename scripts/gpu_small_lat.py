"""Latency of small inputs (one random block = config 2, 1 / 2 / 4 / 8 text blocks): best of 5 encodes each, bit-exact vs the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus
from oracle import pyoracle as po
text = corpus.workload(100_000_000)[0]
sets = [("c2-random-block", corpus.xorshift_bytes(899_999))] + [(f"text-{k}-blocks", text[:k * 890_000]) for k in (1, 2, 4, 5, 8)]
ctx = nv.Context(0, 9)
for name, data in sets:
    n = int(data.size)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda"); d_in[:n] = torch.from_numpy(np.array(data)).cuda()
    cap = (n + n // 4 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    ok = d_out[:ln].cpu().numpy().tobytes() == po.encode(np.array(data).tobytes(), 9)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"{name}: {best*1e3:.3f} ms  bit-exact {ok}")
