"""Throughput of bzh_encode_device on the bench workload for several batch sizes (blocks per kernel batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from banzai_amd import _native as nv, corpus
n = 100_000_000
seg, _ = corpus.workload(n)
d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda"); d_in[:n] = torch.from_numpy(seg).cuda()
cap = (n // 2 + (1 << 20)) & ~3
d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
for mb in (128, 64, 57, 38, 29, 16, 128):
    ctx = nv.Context(0, 9, mb)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("max_batch", mb, "%.2f ms  %.0f MB/s" % (dt * 1e3, n / dt / 1e6), flush=True)
    ctx.close()
