"""Stage milliseconds (profiling on) and wall time (profiling off) of small text batches: argv = block counts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus
dev = torch.device("cuda", 0)
ctx = nv.Context(0, 9, 128)
text = corpus.workload(32_000_000)[0]
for k in [int(x) for x in sys.argv[1:]] or [3, 4, 5, 6]:
    data = text[:k * 890_000]
    n = int(data.size)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev); d_in[:n] = torch.from_numpy(np.array(data, dtype=np.uint8, copy=True)).to(dev)
    cap = (n + n // 4 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    best = None
    for it in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        if it: best = dt if best is None or dt < best else best
    ctx.set_profiling(True)
    ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
    st = ctx.stats()
    ks = sorted(ctx.kernel_stats(), key=lambda r: -r["ms"])[:7]
    ctx.set_profiling(False)
    print(f"x{k}: {best*1e3:.3f} ms; stages", {s: round(st[s], 3) for s in ("ms_plan", "ms_rle1", "ms_bwt", "ms_mtf", "ms_huff", "ms_pack")}, "rounds", st["bwt_rounds"],
          [(r["name"][:18], round(r["ms"], 3)) for r in ks], flush=True)
