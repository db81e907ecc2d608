"""bzh_stream_feed from host memory the runtime has never seen (a fresh 100 MB array per encode, all kept alive) against
the same array fed again: what a one-shot caller pays for pageable H2D.  Also banzai_amd.encode over fresh BytesIO."""
import ctypes, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import banzai_amd
from banzai_amd import _native as nv, corpus
n = 100_000_000
base = corpus.workload(n)[0]
ctx = nv.Context(0, 9, 128)
FEED = 16 << 20
sbuf = np.empty(int(nv.lib().bzh_stream_bound(ctx.handle, n)) + (64 << 20), dtype=np.uint8)
got = ctypes.c_size_t(0)
def run(a):
    t = time.perf_counter()
    ctx.stream_begin()
    for k in range(0, n, FEED):
        v = a[k:k + FEED]
        ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(v), v.size, 0, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
    ctx.check(nv.lib().bzh_stream_feed(ctx.handle, nv.ptr(sbuf), 0, 1, nv.ptr(sbuf), sbuf.size, ctypes.byref(got)))
    return (time.perf_counter() - t) * 1e3
warm = np.array(base, copy=True)
run(warm); run(warm)
keep = []
for it in range(4):
    a = np.array(base, copy=True); keep.append(a)
    print("fresh array: %.2f ms   the same array again: %.2f ms   the warm array: %.2f ms" % (run(a), run(a), run(warm)), flush=True)
for it in range(3):
    raw = bytes(base.tobytes()); keep.append(raw)
    out = io.BytesIO()
    t = time.perf_counter()
    banzai_amd.encode(io.BytesIO(raw), out, 9)
    print("banzai_amd.encode over fresh bytes: %.2f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
