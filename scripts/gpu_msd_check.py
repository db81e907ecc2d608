"""The two initial sorts of the suffix sorter (argv[1] = msd: bucket-first, banzai_amd/csrc/bwt_msd.h, the default; lsd: the 8
passes for every block) against the oracle: last column + origin
pointer of single blocks and whole streams at levels 2, 5 and 9 (level 1's blocks keep the 8-pass path), over inputs
that reach every part of it -- text (units of packed small buckets, oversized buckets split level by level), runs of
one byte (a bucket that stays oversized through all five levels: the "one group" units), repetitive blocks (kept on
the 8-pass path by the sample test), random bytes (all 65,536 buckets: the window packing), short and tiny blocks,
mixed batches.  Run by tests/test_gpu_parity.py::test_bucket_first_initial_sort in a process of its own: the switch is
read once per process.  Exit code 1 on any mismatch."""
import os
import sys

if len(sys.argv) > 2 and sys.argv[2] == "nomid":  # the large groups of the bucket-first blocks through the global passes as well (BZH_MID=0)
    os.environ["BZH_MID"] = "0"
os.environ["BZH_INIT"] = sys.argv[1] if len(sys.argv) > 1 else "msd"  # "msd": bucket-first where the plan allows it (the default); "lsd": 8 passes everywhere
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv, corpus
from oracle import pyoracle as po
from tests import cases

bad = 0
text = corpus.enwik_synthetic_v2(4_000_000, seed=5).tobytes()
code = (b"    " * 3 + b"if (x[i] == y[i]) {\n" + b"        " + b"return value;\n" + b"    }\n") * 40_000
blocks = {
    "text": text[:899_999], "text-short": text[:200_000], "text-17": text[:17], "one": b"q", "two": b"qq",
    "code-with-indent-runs": code[:899_999],
    "runs": cases.gen(899_999, "longruns", 3), "shortruns": cases.gen(500_000, "shortruns", 4),
    "random": corpus.xorshift_bytes(899_999).tobytes(), "random-small": corpus.xorshift_bytes(20_000).tobytes(),
    "lowalpha": cases.gen(600_000, "lowalpha", 2), "periodic": cases.gen(700_001, "periodic", 9),
    "same": b"\x07" * 300_000, "phrases": cases.repeats(880_000, 12),
}
with nv.Context(0, 9, 8) as ctx:
    for name, d in blocks.items():
        g, o = ctx.bwt(d), po.bwt(d)
        if not (g[0] == o[0] and g[1] == o[1] and np.array_equal(g[2], o[2])):
            bad += 1
            print("BWT MISMATCH", name, len(d))
streams = [(9, text + cases.gen(1_300_000, "longruns", 8) + code[:2_000_000] + corpus.xorshift_bytes(950_000).tobytes() + text[:123_457]),
           (5, text[:2_700_001] + b"\0" * 400_000 + cases.repeats(900_000, 3)), (2, text[:1_000_000] + code[:777_777]),
           (9, b""), (9, b"x"), (9, corpus.pathological(6_000_000).tobytes())]
for level, d in streams:
    with nv.Context(0, level, 8) as ctx:
        if ctx.encode(d) != po.encode(d, level):
            bad += 1
            print("STREAM MISMATCH level", level, len(d))
print("initial sort", os.environ["BZH_INIT"] + ":", len(blocks), "blocks,", len(streams), "streams, mismatches:", bad)
sys.exit(1 if bad else 0)
