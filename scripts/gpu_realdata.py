import sys, os, time, glob, bz2
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv
from oracle import pyoracle as po
def collect(patterns, limit):
    buf = bytearray()
    for pat in patterns:
        for f in sorted(glob.glob(pat, recursive=True)):
            try:
                if os.path.isfile(f): buf += open(f, 'rb').read()
            except Exception: pass
            if len(buf) >= limit: return bytes(buf[:limit])
    return bytes(buf)
sets = {"python-sources": collect(['/usr/lib/python3.10/**/*.py', '/usr/lib/python3/dist-packages/**/*.py'], 60_000_000),
        "shared-libs": collect(['/usr/lib/x86_64-linux-gnu/*.so*'], 60_000_000)}
dev = torch.device("cuda", 0)
ctx = nv.Context(0, 9, 128)
for name, d in sets.items():
    n = len(d)
    if n < 1_000_000: print(name, "too small", n); continue
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.frombuffer(bytearray(d), dtype=torch.uint8).to(dev)
    cap = (n + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    ctx.set_profiling(True)
    for it in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = ctx.stats()
    g = d_out[:ln].cpu().numpy().tobytes()
    t = time.perf_counter(); o = po.encode(d, 9); dtc = time.perf_counter() - t
    print("%s: %d bytes -> %d (%.3f); GPU %.1f ms = %.0f MB/s; oracle %.1f MB/s; x%.0f; bit-exact %s; rounds %d, A/n %.2f, bwt %.1f ms"
          % (name, n, ln, ln / n, dt * 1e3, n / dt / 1e6, n / dtc / 1e6, dtc / dt, g == o, st['bwt_rounds'],
             st['bwt_active_sum'] / max(1, st['rle_bytes']), st['ms_bwt']))
