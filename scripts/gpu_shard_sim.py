"""What every rank of an N-rank sharded encode does, timed on a single GPU (no communication): the chained split
(tables over the own range + look-ahead, split from the start handed over), the encode of the own blocks.
    python scripts/gpu_shard_sim.py [N] [total_bytes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus, sharded

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
seg = 100_000_000
host = np.empty(total, dtype=np.uint8)
for k in range((total + seg - 1) // seg):
    part = corpus.workload(min(seg, total - k * seg), segment=k)[0]
    host[k * seg:k * seg + part.size] = part
ctx = nv.Context(0, 9, 128)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
slab = sharded.worst_case_slab(total, N, 9)
d_out = torch.zeros(16, dtype=torch.uint8, device="cuda")
def sync(): torch.cuda.synchronize()
rows = []
for it in range(2):
    start, rows = 0, []
    for r in range(N):
        lo, hi = sharded.resident_range(total, r, N)
        d_r = torch.zeros(hi - lo + 16, dtype=torch.uint8, device="cuda")
        d_r[:hi - lo] = torch.from_numpy(host[lo:hi]).cuda()
        eng = sharded.DeviceEngine(ctx, d_r, total, d_out, slab, resident=hi - lo, lo=lo)
        sync(); t0 = time.perf_counter()
        eng.tables(); sync(); t1 = time.perf_counter()
        blocks, b0, b1, start = sharded.own_blocks(eng, r, N, start); sync(); t2 = time.perf_counter()
        part, nbits = eng.encode_range(b0, b1); sync(); t3 = time.perf_counter()
        rows.append({"rank": r, "resident_MB": round((hi - lo) / 1e6, 1), "blocks": b1 - b0, "ms_tables": round(1e3 * (t1 - t0), 3),
                     "ms_split": round(1e3 * (t2 - t1), 3), "ms_encode": round(1e3 * (t3 - t2), 3)})
        del eng, d_r, part
tot = [r["ms_tables"] + r["ms_split"] + r["ms_encode"] for r in rows]
import json
print(json.dumps({"world": N, "total_bytes": total, "ranks": rows, "plan_plus_encode_max_over_mean": round(max(tot) / (sum(tot) / len(tot)), 4),
                  "chain_ms_before_last_rank": round(sum(r["ms_split"] for r in rows[:-1]), 3)}))
