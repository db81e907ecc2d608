"""Quick GPU bring-up check of the BWT seam against the oracle (not a pytest)."""
import sys, time, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv
from oracle import pyoracle as po

ctx = nv.Context(0, 9, 8)
rng = random.Random(5)
cases = [b"a", b"ab", b"abab", b"aaaa", b"banana", b"abcabcabcabc", b"mississippi" * 50]
for k in range(30):
    n = rng.choice([1, 2, 3, 5, 63, 64, 65, 257, 4095, 4096, 4097, 10000, 70000])
    sig = rng.choice([1, 2, 4, 256])
    d = bytes(rng.randrange(sig) for _ in range(n))
    if rng.random() < 0.3:
        w = d[:rng.randint(1, max(1, n // 3))]
        d = (w * (n // len(w) + 1))[:n]
    cases.append(d)
bad = 0
for d in cases:
    g = ctx.bwt(d)
    o = po.bwt(d)
    if g[0] != o[0] or g[1] != o[1] or not np.array_equal(g[2], o[2]):
        bad += 1
        print("MISMATCH n=", len(d), "ptr", g[1], o[1], "bwt eq", g[0] == o[0], "hb eq", np.array_equal(g[2], o[2]))
print("small cases:", len(cases), "bad:", bad)
# big: random 899,999 and text-like
big = np.random.default_rng(1).integers(0, 256, 899_999, dtype=np.uint8).tobytes()
words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 9))) for _ in range(3000)]
text = b" ".join(rng.choice(words) for _ in range(200000))[:899_999]
zer = (b"\0\0\0\0\xfb" * 180000)[:899_998]
for name, d in (("random", big), ("text", text), ("nearperiodic", zer)):
    ctx.set_profiling(True)
    t = time.time(); g = ctx.bwt(d); tg = time.time() - t
    t = time.time(); o = po.bwt(d); to = time.time() - t
    ok = g[0] == o[0] and g[1] == o[1] and np.array_equal(g[2], o[2])
    print(name, "ok" if ok else "MISMATCH", f"gpu {tg*1e3:.1f} ms  oracle {to*1e3:.1f} ms", ctx.stats())
    bad += 0 if ok else 1
# batch of 8 text blocks
blocks = [text[i * 1000:] + text[:i * 1000] for i in range(8)]
t = time.time(); res = ctx.bwt_batch(blocks); tg = time.time() - t
okb = all(r[0] == po.bwt(b)[0] for r, b in zip(res, blocks))
print("batch8 text", "ok" if okb else "MISMATCH", f"{tg*1e3:.1f} ms")
sys.exit(1 if bad or not okb else 0)
