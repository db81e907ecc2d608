import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv
from oracle import pyoracle as po
ctx = nv.Context(0, 9, 4)
rng = random.Random(11)
bad = 0
def check(d, name=""):
    global bad
    b, ptr, hb = po.bwt(d)
    s, f, ns = po.mtf_and_rle(b, hb)
    t = time.time(); gbits, gn, glens = ctx.huffman(s, ns, f); tg = time.time() - t
    obits, on, olens = po.huffman_block(s, ns, f)
    ok = gn == on and gbits == obits and np.array_equal(glens[:, :ns], olens[:, :ns])
    if not ok:
        bad += 1
        k = next((i for i in range(min(len(gbits), len(obits))) if gbits[i] != obits[i]), None)
        print("MISMATCH", name, "n", len(d), "m", len(s), "nsyms", ns, "bits", gn, on, "first diff byte", k,
              "lens eq", np.array_equal(glens[:, :ns], olens[:, :ns]))
    elif name: print(name, "ok bits=%d %.2f ms" % (gn, tg * 1e3))
for k in range(60):
    n = rng.choice([1, 2, 3, 49, 50, 51, 100, 1000, 4095, 4096, 4097, 20000, 70000, 200000])
    sig = rng.choice([1, 2, 3, 16, 100, 200, 256])
    mode = rng.randrange(3)
    if mode == 0: d = bytes(rng.randrange(sig) for _ in range(n))
    elif mode == 1:
        d = bytearray()
        while len(d) < n: d += bytes([rng.randrange(sig)]) * rng.choice([1, 2, 5, 40, 300, 3000])
        d = bytes(d[:n])
    else: d = (b"the quick brown fox jumps over the lazy dog. " * (n // 40 + 1))[:n]
    check(d)
print("small bad:", bad)
big = np.random.default_rng(1).integers(0, 256, 899_999, dtype=np.uint8).tobytes()
words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 9))) for _ in range(3000)]
text = b" ".join(rng.choice(words) for _ in range(200000))[:899_999]
zer = (b"\0\0\0\0\xfb" * 180000)[:899_998]
skew = bytes(min(255, int(rng.expovariate(0.05))) for _ in range(300000))
for name, d in (("random", big), ("text", text), ("nearperiodic", zer), ("allsame", b"a" * 500000), ("skew", skew)):
    check(d, name)
sys.exit(1 if bad else 0)
