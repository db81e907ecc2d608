import sys, os, random, time, bz2
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv
from oracle import pyoracle as po
rng = random.Random(21)
bad = 0
def gen(n, mode):
    if mode == 0: return bytes(rng.randrange(256) for _ in range(n))
    if mode == 1:
        d = bytearray()
        while len(d) < n: d += bytes([rng.randrange(3)]) * rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8])
        return bytes(d[:n])
    if mode == 2:
        d = bytearray()
        while len(d) < n: d += bytes([rng.randrange(2)]) * rng.choice([1, 3, 4, 5, 254, 255, 256, 257, 258, 259, 260, 509, 510, 511, 1000, 70000])
        return bytes(d[:n])
    if mode == 3: return (b"It was the best of times, it was the worst of times, " * (n // 50 + 1))[:n]
    return bytes([rng.randrange(256)]) * n
for level in (1, 9):
    ctx = nv.Context(0, level, 8)
    sizes = [0, 1, 2, 3, 4, 5, 255, 256, 1000, 99998, 99999, 100000, 100001, 250000, 431007] if level == 1 else [0, 1, 1000, 899999, 900000, 1000000, 2500000]
    for n in sizes:
        for mode in range(5):
            d = gen(n, mode)
            # stage checks
            infos, chunks = ctx.rle1_split(d)
            off = 0; ob = []
            while off < len(d):
                r, crc, used = po.rle_one(d[off:], level); ob.append(((off, used, len(r), crc), r)); off += used
            if [x[0] for x in ob] != infos or [x[1] for x in ob] != chunks:
                bad += 1; print("RLE1 MISMATCH level", level, "n", n, "mode", mode, infos[:3], [x[0] for x in ob][:3])
                continue
            if n and ctx.crc32(d) != po.crc32(d):
                bad += 1; print("CRC MISMATCH", n, mode)
            g = ctx.encode(d); o = po.encode(d, level)
            if g != o:
                bad += 1
                k = next((i for i in range(min(len(g), len(o))) if g[i] != o[i]), None)
                print("STREAM MISMATCH level", level, "n", n, "mode", mode, "len", len(g), len(o), "first diff", k)
            elif bz2.decompress(g) != d:
                bad += 1; print("ROUNDTRIP FAIL", level, n, mode)
    ctx.close()
print("bad:", bad)
sys.exit(1 if bad else 0)
