import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus
from oracle import pyoracle as po
n = 100_000_000
data = corpus.pathological(n)
dev = torch.device("cuda", 0)
d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.from_numpy(data).to(dev)
cap = (n // 2 + (1 << 20)) & ~3
d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
ctx = nv.Context(0, 9, 128)
ctx.set_profiling(True)
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
st = ctx.stats()
print("C5 pathological 100MB: %.1f ms = %.1f MB/s, out %d bytes, blocks %d, rounds %d, A/n %.2f" % (dt*1e3, n/dt/1e6, ln, st['blocks'], st['bwt_rounds'], st['bwt_active_sum']/max(1,st['rle_bytes'])))
print({k: round(v, 2) for k, v in st.items() if k.startswith('ms_')})
g = d_out[:ln].cpu().numpy().tobytes()
parts = [(0, 25_000_000), (25_000_000, 50_000_000), (50_000_000, 75_000_000), (75_000_000, n)]
t = time.perf_counter(); o = po.encode(data.tobytes(), 9); dtc = time.perf_counter() - t
print("oracle: %.1f s = %.1f MB/s; bit-exact:" % (dtc, n/dtc/1e6), g == o)
for a, b in parts:
    seg = data[a:b].tobytes()
    t = time.perf_counter(); po.encode(seg, 9); d1 = time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.encode_device(d_in.data_ptr() + a, b - a, d_out.data_ptr(), cap) if a % 16 == 0 else None
    torch.cuda.synchronize(); d2 = time.perf_counter() - t
    print("  part %d-%d: cpu %.1f MB/s  gpu %.1f MB/s" % (a, b, (b-a)/d1/1e6, (b-a)/d2/1e6))
