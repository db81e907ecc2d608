"""On-box fuzz loop in the spirit of the reference's fuzz/fuzz_targets/round_trip.rs (encode arbitrary bytes, decode,
compare) -- time-boxed, seeded, on cuda:0 through the C ABI.  Every input is
  * encoded by the HIP path and compared bit for bit with the CPU oracle's stream,
  * decoded again by the strict in-repo decoder (and libbz2) and compared with the input,
  * and its blocks go through GPU BWT -> GPU inverse BWT (bzh_unbwt_batch).
Writes a JSON summary (argv: seconds [seed] [out.json])."""
import bz2, json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banzai_amd import _native as nv
from oracle import pyoracle as po
from tests import cases

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/fuzz.json"
rng = random.Random(seed)
ctxs = {lv: nv.Context(0, lv, 8) for lv in (1, 2, 5, 9)}
t0 = time.time()
stats = {"seed": seed, "seconds": seconds, "inputs": 0, "bytes": 0, "blocks_inverted": 0, "by_level": {}, "failures": []}


def arbitrary(rng):
    kind = rng.randrange(8)
    n = rng.choice([0, 1, 2, 5, 300, 5000, 99_999, 100_000, 120_000, 400_000, 1_500_000])
    if kind == 0:
        return cases.mixture(rng, max(1, n))
    if kind == 1:
        return cases.phrase_groups(max(70_000, n), rng.randrange(1000))
    if kind == 2:  # a word repeated: identical rotations, period not dividing the block size
        w = bytes(rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 7, 64, 1000])))
        return (w * (n // len(w) + 1))[:n]
    if kind == 3:  # runs around the RLE1 limits
        d = bytearray()
        while len(d) < n:
            d += bytes([rng.randrange(4)]) * rng.choice([1, 2, 3, 4, 5, 254, 255, 256, 257, 258, 259, 510, 765, 1020, 70_000])
        return bytes(d[:n])
    if kind == 4:
        return bytes(rng.randrange(256) for _ in range(min(n, 200_000)))
    if kind == 5:  # small alphabets: 2-table blocks, long code lengths
        k = rng.choice([2, 3, 5, 17])
        return bytes(rng.choice(range(k)) if rng.random() < 0.9 else rng.randrange(256) for _ in range(min(n, 300_000)))
    return cases.gen(n, rng.choice(["text", "longruns", "shortruns", "same", "random"]), rng.randrange(1 << 30))


while time.time() - t0 < seconds:
    lv = rng.choice([1, 1, 2, 5, 9])
    d = arbitrary(rng)
    ctx = ctxs[lv]
    try:
        g = ctx.encode(d)
        want = po.encode(d, lv)
        ok = g == want and po.decode(g, cap=len(d) + 64) == d and bz2.decompress(g) == d
        infos, chunks = ctx.rle1_split(d)
        if chunks:
            fwd = ctx.bwt_batch(chunks)
            ok = ok and ctx.unbwt_batch([(bw, p) for bw, p, _ in fwd]) == chunks
            stats["blocks_inverted"] += len(chunks)
    except Exception as e:  # noqa: BLE001
        ok = False
        stats["failures"].append({"input": stats["inputs"], "level": lv, "len": len(d), "error": repr(e)})
    if not ok:
        if not stats["failures"] or stats["failures"][-1].get("input") != stats["inputs"]:
            stats["failures"].append({"input": stats["inputs"], "level": lv, "len": len(d)})
        open(f"gpurun_out/fuzz_fail_{stats['inputs']}.bin", "wb").write(d)
        if len(stats["failures"]) >= 5:
            break
    stats["inputs"] += 1
    stats["bytes"] += len(d)
    stats["by_level"][str(lv)] = stats["by_level"].get(str(lv), 0) + 1
stats["elapsed_s"] = round(time.time() - t0, 1)
stats["ok"] = not stats["failures"]
json.dump(stats, open(out, "w"), indent=1)
print(json.dumps(stats))
sys.exit(0 if stats["ok"] else 1)
