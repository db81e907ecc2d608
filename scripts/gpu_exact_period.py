"""Text-like blocks that are EXACTLY periodic (a stretch of text repeated k times, shorter than a block): every group holds k
identical rotations, nothing is refined once the depth exceeds the period, the block jumps to depth "h >= n" and its groups
are ordered by index -- through the small-group kernel (k <= 64) and through the large groups' path (k = 113:
mid_sort on the bucket-first sort, the global passes otherwise).  argv[1] = msd | lsd | default (BZH_INIT).  Exit code 1 on
a mismatch with the oracle.  Run by tests/test_gpu_parity.py::test_exactly_periodic_text_blocks."""
import os, sys
if len(sys.argv) > 1 and sys.argv[1] in ("msd", "lsd"):
    os.environ["BZH_INIT"] = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv, corpus
from oracle import pyoracle as po
text = corpus.enwik_synthetic_v2(3_000_000, seed=9).tobytes()
bad = 0
with nv.Context(0, 9, 16) as ctx:
    for p, k in ((100_003, 8), (50_021, 17), (299_993, 3), (7_919, 113), (449_999, 2)):
        d = text[1000:1000 + p] * k
        g, o = ctx.bwt(d), po.bwt(d)
        ok = g[0] == o[0] and g[1] == o[1]
        print('exact period', p, 'x', k, 'n', len(d), 'ok' if ok else 'MISMATCH')
        bad += not ok
    # the same inside a stream of text blocks (a batch on the bucket-first sort)
    d = text[:2_000_000] + text[1000:1000 + 100_003] * 8 + text[:1_500_000]
    ok = ctx.encode(d) == po.encode(d, 9)
    print('stream with an exactly periodic stretch', 'ok' if ok else 'MISMATCH')
    bad += not ok
print('bad:', bad)
sys.exit(1 if bad else 0)
