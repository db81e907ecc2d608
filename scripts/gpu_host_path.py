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
# the C entry point alone: numpy views in, preallocated output (no Python-side copies)
sbuf = np.empty(400 << 20, dtype=np.uint8)  # >= bzh_stream_bound() for every call below
got = ctypes.c_size_t(0)
for it in range(2):
    t = time.perf_counter()
    ctx.stream_begin()
    tot = 0
    for k in range(0, n, 16 << 20):
        v = data[k:k + (16 << 20)]
        ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(v), v.size, 0, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
        tot += got.value
    ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(sbuf), 0, 1, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
    tot += got.value
    dt = time.perf_counter() - t
    print("bzh_stream_feed (C ABI only) 16 MiB chunks: %.1f ms = %.0f MB/s, %d bytes out" % (dt * 1e3, n / dt / 1e6, tot))
# a longer stream (4 x the workload): passes overlap with feeding
big = np.concatenate([data] * 4)
for it in range(2):
    t = time.perf_counter()
    ctx.stream_begin()
    tot = 0
    for k in range(0, big.size, 16 << 20):
        v = big[k:k + (16 << 20)]
        ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(v), v.size, 0, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
        tot += got.value
    ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(sbuf), 0, 1, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
    tot += got.value
    dt = time.perf_counter() - t
    print("bzh_stream_feed (C ABI only) 400 MB in 16 MiB chunks: %.1f ms = %.0f MB/s, %d bytes out" % (dt * 1e3, big.size / dt / 1e6, tot))
import bz2
ctx.stream_begin()
outb = b"".join(ctx.stream_feed(big[k:k + (48 << 20)].tobytes()) for k in range(0, big.size, 48 << 20)) + ctx.stream_feed(b"", eof=True)
print("400 MB stream decodes to the input:", bz2.decompress(outb) == big.tobytes())
