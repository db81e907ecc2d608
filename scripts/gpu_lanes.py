"""Throughput of bzh_encode_device with 1 and 2 lanes (bzh_set_lanes) on the bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from banzai_amd import _native as nv, corpus
n = 100_000_000
seg, _ = corpus.workload(n)
d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda"); d_in[:n] = torch.from_numpy(seg).cuda()
cap = (n // 2 + (1 << 20)) & ~3
d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
ref = None
for lanes in (1, 2, 1, 2):
    ctx = nv.Context(0, 9, 128)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_lanes(lanes)
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    out = d_out[:ln].clone()
    if ref is None: ref = out
    print("lanes", lanes, "%.2f ms  %.0f MB/s  same bytes: %s" % (dt * 1e3, n / dt / 1e6, bool(torch.equal(out, ref))), flush=True)
    ctx.close()
