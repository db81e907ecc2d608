"""Find small inputs on which the GPU BWT differs from the oracle's and describe the first difference."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv
from oracle import pyoracle as po
from tests import cases
ctx = nv.Context(0, 9, 8)
rng = random.Random(1)
def check(d, tag):
    g = ctx.bwt(d); o = po.bwt(d)
    if g[0] == o[0] and g[1] == o[1]:
        return True
    gb, ob = g[0], o[0]
    k = next((i for i in range(len(d)) if gb[i] != ob[i]), None)
    nd = sum(1 for i in range(len(d)) if gb[i] != ob[i])
    print(f"MISMATCH {tag}: n={len(d)} first diff at SA position {k}, {nd} positions differ, ptr {g[1]} vs {o[1]}")
    # true suffix array by numpy for context
    s = np.frombuffer(d, dtype=np.uint8)
    n = len(d)
    if n <= 200000 and k is not None:
        dd = np.concatenate([s, s])
        order = sorted(range(n), key=lambda i: (bytes(dd[i:i + n]), -i))
        for p in range(max(0, k - 2), min(n, k + 3)):
            i = order[p]
            print(f"   pos {p}: suffix {i}: {bytes(dd[i:i+48])!r}")
    return False
ok = True
for n, seed, counts in ((120_000, 5, (600, 900)), (60_000, 6, (300, 500)), (200_000, 7, (700, 1500))):
    ok &= check(cases.phrase_groups(n, seed, counts), f'phrase n={n}')
if not ok: sys.exit(1)
for n in (50, 200, 1000, 5000, 20000, 100000):
    for trial in range(6):
        kind = trial % 3
        if kind == 0:
            d = cases.phrase_groups(n, trial + n, counts=(max(2, n // 80), max(2, n // 40)))
        elif kind == 1:
            d = bytes(rng.choice(b"abc") for _ in range(n))
        else:
            w = bytes(rng.choice(b"abcdefgh") for _ in range(rng.choice([3, 17, 40])))
            d = (w * (n // len(w) + 1))[:n - 1] + b"z"
        ok &= check(d, f"kind{kind} n={n} trial={trial}")
        if not ok: sys.exit(1)
print("all equal")
