"""What ONE rank of an N-rank sharded encode does, timed on a single GPU (no communication):
plan over the whole N x 100 MB input, encode its own 1/N of the blocks, assemble N segments."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from banzai_amd import _native as nv, corpus, sharded

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seg = 100_000_000
parts = [corpus.workload(seg, segment=k % 2)[0] for k in range(2)]
total = seg * N
d_in = torch.empty(total + 16, dtype=torch.uint8, device="cuda")
for k in range(N):
    d_in[k * seg:(k + 1) * seg] = torch.from_numpy(parts[k % 2]).cuda()
out_cap = (total // 3 + total // 8 + (1 << 20)) & ~3
d_out = torch.zeros(out_cap, dtype=torch.uint8, device="cuda")
ctx = nv.Context(0, 9, 128)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
seg_cap = (seg // 3 + seg // 8 + (1 << 20)) & ~3
eng = sharded.DeviceEngine(ctx, d_in, total, d_out, seg_cap)
def sync(): torch.cuda.synchronize()
for it in range(3):
    sync(); t0 = time.perf_counter()
    r = N // 2
    blocks, b0, b1 = sharded.own_blocks(eng, r, N); sync(); t1 = time.perf_counter()
    part, nbits = eng.encode_range(b0, b1); sync(); t2 = time.perf_counter()
    segs = [(part, nbits)] * N
    crcs = eng.crcs(b0, b1) * N  # timing only: the other ranks' CRCs would arrive with the gather
    n_out = eng.assemble(segs, crcs); sync(); t3 = time.perf_counter()
    print(f"N={N} rank {r}: planned {int(blocks[-1][0]) + int(blocks[-1][1])} of {total} bytes, own blocks={b1-b0}  plan {1e3*(t1-t0):.2f} ms  encode_range {1e3*(t2-t1):.2f} ms  assemble({N} segs, {nbits//8/1e6:.1f} MB each) {1e3*(t3-t2):.2f} ms  total {1e3*(t3-t0):.2f}", flush=True)
for r in (0, N - 1):
    sync(); t0 = time.perf_counter()
    blocks, b0, b1 = sharded.own_blocks(eng, r, N); sync(); t1 = time.perf_counter()
    part, nbits = eng.encode_range(b0, b1); sync(); t2 = time.perf_counter()
    print(f"N={N} rank {r}: own blocks={b1-b0}  plan {1e3*(t1-t0):.2f} ms  encode_range {1e3*(t2-t1):.2f} ms  sum {1e3*(t2-t0):.2f}", flush=True)
