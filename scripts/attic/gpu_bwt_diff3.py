import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv, corpus
from oracle import pyoracle as po
ctx = nv.Context(0, 9, 128)
data = corpus.image_corpus("python-sources").tobytes()
infos, chunks = ctx.rle1_split(data)
res = ctx.bwt_batch(chunks)
for k, (r, blk) in enumerate(zip(res, chunks)):
    o = po.bwt(blk)
    if r[0] == o[0] and r[1] == o[1]:
        continue
    gb = np.frombuffer(r[0], np.uint8); ob = np.frombuffer(o[0], np.uint8)
    diff = np.flatnonzero(gb != ob)
    n = len(blk)
    print(f"block {k} n={n}: {diff.size} positions differ, first {diff[:8]}, ptr {r[1]} vs {o[1]}", flush=True)
    s = np.frombuffer(blk, np.uint8)
    idx = np.arange(n); rank = s.astype(np.int64); kk = 1
    while True:
        key = rank * (n + 257) + rank[(idx + kk) % n]
        order = np.argsort(key, kind="stable"); ks = key[order]
        rank = np.empty(n, np.int64); rank[order] = np.concatenate(([0], np.cumsum(ks[1:] != ks[:-1])))
        if rank.max() == n - 1: break
        kk *= 2
        if kk >= n: order = np.lexsort((-idx, rank)); break
    sa = order
    dd = np.concatenate([s, s])
    for p in diff[:3]:
        for q in range(int(p) - 1, int(p) + 3):
            i = int(sa[q])
            # common prefix length with the next suffix in true order
            j = int(sa[q + 1]) if q + 1 < n else i
            l = 0
            while l < 5000 and dd[i + l] == dd[j + l]: l += 1
            print(f"   true pos {q}: suffix {i} lcp-with-next {l}  {bytes(dd[i:i+40])!r}")
    # also single-block run of the same block
    r1 = ctx.bwt(blk)
    print("   same block alone:", "equal" if r1[0] == o[0] and r1[1] == o[1] else "DIFF too")
    break
else:
    print("all blocks equal")
