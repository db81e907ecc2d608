"""Multi-batch streams on one GPU: 1 GB (config 4, ten 100 MB segments of the generator) and a C4 rank's 125 MB, lanes 1 vs 2."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t = time.perf_counter()
segs = [corpus.workload(100_000_000, s)[0] for s in range(nseg)]
print("generated %d segments in %.1f s" % (nseg, time.perf_counter() - t), flush=True)
for total in ([125_000_000] + ([nseg * 100_000_000] if nseg > 1 else [])):
    n = total
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
    off = 0
    for s in segs:
        k = min(len(s), n - off)
        if k <= 0: break
        d_in[off:off + k] = torch.from_numpy(s[:k]).cuda(); off += k
    cap = (n // 2 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    ref = None
    for lanes, mb in ((1, 128), (1, 160), (1, 256), (1, 384), (1, 576), (2, 256)):
        ctx = nv.Context(0, 9, mb)
        ctx.set_lanes(lanes)
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        out = d_out[:ln].clone()
        if ref is None: ref = out
        print("%d bytes lanes %d max_batch %d: %.2f ms  %.0f MB/s  same bytes: %s" % (n, lanes, mb, dt * 1e3, n / dt / 1e6, bool(torch.equal(out, ref))), flush=True)
        ctx.close()
    del d_in, d_out
