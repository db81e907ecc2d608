"""bit-exactness of the library in use against the oracle: a mixed 16-block batch (text + runs + periodic) and a small one"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banzai_amd import _native, corpus
from oracle import pyoracle
ok = True
text = corpus.workload(12_000_000)[0].tobytes()
real = corpus.image_corpus("real-text-100MB")[:6_000_000].tobytes() if "real-text-100MB" in corpus.IMAGE_SETS else b""
cases = [text + b"\0" * 70000 + b"abab" * 5000 + real, corpus.enwik_synthetic(2_300_000, seed=7).tobytes() + b"xyz" * 40000]
for data in cases:
    with _native.Context(0, 9, 32) as ctx:
        got = ctx.encode(data)
    ok &= got == pyoracle.encode(data, 9)
print("bit-exact vs oracle:", ok)
