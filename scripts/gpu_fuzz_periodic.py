"""On-box fuzz of the period path (bwt.hip: period_detect / period_expand): blocks that are a word repeated -- and blocks that
almost are -- encoded by the HIP path and compared bit for bit with the CPU oracle's stream.  Time-boxed and seeded
(argv: seconds [seed] [out.json]).  Shapes: period 1 .. 9,000 (the limit is 8,192), alphabets of 2 .. 256 letters, the
repetition cut anywhere, at level 1 .. 9 (blocks of 100 kB .. 900 kB, several per input so that later blocks start at another
phase of the word); "almost": one byte damaged at a random place (start, middle, the last period, the very end), a different
word in front or behind, a run of four equal bytes inside the word (RLE1 changes the period), exactly periodic blocks."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banzai_amd import _native as nv
from oracle import pyoracle as po

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/fuzz_periodic.json"
rng = random.Random(seed)
ctxs = {}
stats = {"seed": seed, "seconds": seconds, "inputs": 0, "bytes": 0, "by_kind": {}, "failures": []}
t0 = time.time()
while time.time() - t0 < seconds:
    level = rng.choice([1, 1, 2, 3, 5, 9, 9])
    M = 100000 * level - 1
    p = rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 31, 64, 255, 256, 257, 1000, 1024, 4096, 8191, 8192, 8193, 9000, rng.randint(2, 9000)])
    sigma = rng.choice([2, 2, 3, 4, 16, 256])
    w = bytearray(rng.randrange(sigma) for _ in range(p))
    kind = rng.choice(["plain", "plain", "plain", "damaged", "prefix", "suffix", "runs", "exact", "short"])
    if kind != "runs":  # no run of four inside the word (or across its seam): RLE1 leaves the repetition alone
        for q in range(len(w) + 3):
            a, b, c, d = (w[(q - k) % len(w)] for k in range(4))
            if a == b == c == d:
                w[q % len(w)] = (a + 1) % max(2, sigma)
    nblocks = rng.choice([1, 1, 2, 3])
    n = rng.randint(max(1, M // 2), M) + (nblocks - 1) * M
    if kind == "short":
        n = rng.randint(1, max(2, 12 * p))
    if kind == "exact":
        n = max(p, (min(n, M) // p) * p)
    data = bytearray((bytes(w) * (n // p + 2))[:n])
    if kind == "damaged" and n > 0:
        pos = rng.choice([0, n - 1, max(0, n - p - 1), rng.randrange(n), rng.randrange(n)])
        data[pos] ^= 1 + rng.randrange(255)
    elif kind == "prefix":
        data = bytearray(rng.randrange(256) for _ in range(rng.randint(1, 300))) + data
    elif kind == "suffix":
        data += bytearray(rng.randrange(256) for _ in range(rng.randint(1, 300)))
    data = bytes(data)
    if level not in ctxs:
        ctxs[level] = nv.Context(0, level, 16)
    got = ctxs[level].encode(data)
    want = po.encode(data, level)
    stats["inputs"] += 1
    stats["bytes"] += len(data)
    stats["by_kind"][kind] = stats["by_kind"].get(kind, 0) + 1
    if got != want:
        stats["failures"].append({"kind": kind, "level": level, "p": p, "sigma": sigma, "n": len(data), "seed": seed, "input": stats["inputs"]})
        if len(stats["failures"]) > 20:
            break
stats["elapsed_s"] = round(time.time() - t0, 1)
stats["ok"] = not stats["failures"]
with open(out, "w") as f:
    json.dump(stats, f)
print(json.dumps(stats))
