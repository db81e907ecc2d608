import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv
from oracle import pyoracle as po
rng = np.random.default_rng(99)
blocks = []
for p in (2, 3, 1024, 4099):
    w = rng.integers(0, 256, p, dtype=np.uint8)
    for n in (899_999, 899_998, 450_000):
        blocks.append(np.tile(w, n // p + 1)[:n].tobytes())
w = np.frombuffer(b"ab" * 500 + b"cd", dtype=np.uint8)
blocks.append(np.tile(w, 900)[:899_999].tobytes())
w = rng.integers(0, 2, 777, dtype=np.uint8) + 97
blocks.append(np.tile(w, 1200)[:899_999].tobytes())
blocks.append(np.tile(w, 1200)[:777 * 1000].tobytes())
w = rng.integers(0, 256, 1024, dtype=np.uint8)
d = np.tile(w, 880)[:899_999].copy(); d[450_000] ^= 1; blocks.append(d.tobytes())
d = np.tile(w, 880)[:899_999].copy(); d[899_990] ^= 1; blocks.append(d.tobytes())
blocks.append((b"\0\0\0\0\xfb" * 180_000)[:899_999])
names = ["p2a","p2b","p2c","p3a","p3b","p3c","p1024a","p1024b","p1024c","p4099a","p4099b","p4099c","abcd","low1","low2","dmg1","dmg2","runs"]
ctx = nv.Context(0, 9, 8)
sets = [s.split("+") for s in sys.argv[1:]]
for st in sets:
    bl = [blocks[names.index(x)] for x in st]
    try:
        got = ctx.bwt_batch(bl)
    except Exception as e:
        print("+".join(st), "ERROR", str(e)[-90:]); ctx.close(); ctx = nv.Context(0, 9, 8); continue
    ok = [g[0] == po.bwt(b)[0] and g[1] == po.bwt(b)[1] for g, b in zip(got, bl)]
    print("+".join(st), ok)
