"""Streaming API (bzh_stream_feed) on the bench workload: trigger size sweep through the C ABI."""
import sys, os, time, ctypes, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import banzai_amd
from banzai_amd import _native as nv, corpus
n = 100_000_000
data, _ = corpus.workload(n)
ctx = nv.Context(0, 9, 128)
sbuf = np.empty(400 << 20, dtype=np.uint8)
got = ctypes.c_size_t(0)
ref = None
FEED = 16 << 20
for chunk_mb in (0, 16, 24, 32, 48, 64):
    best = None
    for it in range(3):
        t = time.perf_counter()
        ctx.stream_begin((chunk_mb << 20) if chunk_mb else None)
        parts = []
        for k in range(0, n, FEED):
            v = data[k:k + FEED]
            ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(v), v.size, 0, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
            if it == 0 and got.value: parts.append(sbuf[:got.value].tobytes())
        ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(sbuf), 0, 1, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
        if it == 0: parts.append(sbuf[:got.value].tobytes())
        dt = time.perf_counter() - t
        if it == 0:
            s = b"".join(parts)
            if ref is None: ref = s
            assert s == ref, "stream differs"
        else:
            best = dt if best is None or dt < best else best
    print("C ABI, trigger %s MiB, 16 MiB feeds: %.2f ms = %.0f MB/s" % (chunk_mb or "default", best * 1e3, n / best / 1e6), flush=True)
# (banzai_amd.encode over BytesIO is timed by bench.py: value_stream_api.python_encode)
