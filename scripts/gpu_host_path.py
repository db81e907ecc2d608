import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv, corpus
n = 100_000_000
data, _ = corpus.workload(n)
ctx = nv.Context(0, 9, 128)
cap = n // 2 + (1 << 20)
out = np.zeros(cap, dtype=np.uint8)
olen = ctypes.c_size_t(0); used = ctypes.c_size_t(0)
for it in range(3):
    t = time.perf_counter()
    ctx.check(nv.lib().bzh_encode(ctx.handle, nv.ptr(data), n, nv.ptr(out), cap, ctypes.byref(olen), ctypes.byref(used)))
    dt = time.perf_counter() - t
    print("bzh_encode host->host (pageable numpy buffers): %.1f ms = %.0f MB/s, %d bytes out" % (dt * 1e3, n / dt / 1e6, olen.value))
# streaming API, 16 MiB feeds
for it in range(2):
    t = time.perf_counter()
    ctx.stream_begin()
    tot = 0
    for k in range(0, n, 16 << 20):
        tot += len(ctx.stream_feed(data[k:k + (16 << 20)].tobytes()))
    tot += len(ctx.stream_feed(b"", eof=True))
    dt = time.perf_counter() - t
    print("stream_feed 16 MiB chunks: %.1f ms = %.0f MB/s, %d bytes out" % (dt * 1e3, n / dt / 1e6, tot))
