"""Soak: random mixtures at levels 1/3/9 through the C ABI against the oracle for ~150 s (argv[1] = seed)."""
import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banzai_amd import _native as nv
from oracle import pyoracle as po
from tests import cases
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctxs = {lv: nv.Context(0, lv, 8) for lv in (1, 3, 9)}
t0 = time.time(); n = 0; tot = 0
while time.time() - t0 < 150:
    lv = rng.choice([1, 1, 3, 9])
    d = cases.mixture(rng, rng.choice([5000, 120_000, 400_000, 1_500_000]))
    if rng.random() < 0.2: d = cases.phrase_groups(max(70_000, len(d)), rng.randrange(1000))
    g = ctxs[lv].encode(d)
    if g != po.encode(d, lv):
        print("MISMATCH", n, lv, len(d)); open("gpurun_out/soak_fail.bin", "wb").write(d); sys.exit(1)
    n += 1; tot += len(d)
print("soak ok:", n, "inputs,", tot, "bytes")
