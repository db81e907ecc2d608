"""Fuzz of LARGE heterogeneous batches (scripts/gpu_fuzz.py feeds one to two blocks at a time: batches of 12 blocks and
more take other paths -- the bucket-first initial sort next to the 8 passes on the second stream, blocks in SWEEP mode and
near-periodic blocks beside text in one round, the pinned workgroup mapping from 32 blocks on).  Every input is a
concatenation of segments of different kinds, 8-40 MB, encoded at a random level by a context of 128 blocks a batch and
compared bit for bit with the CPU oracle's stream (oracle runs on a thread pool: the C library releases the GIL); libbz2
decodes every 4th stream.  argv: seconds [seed] [out.json]"""
import bz2, json, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from banzai_amd import _native as nv, corpus
from oracle import pyoracle as po
from tests import cases

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/fuzz_batches.json"
rng = random.Random(seed)
text = bytes(corpus.workload(24_000_000)[0])
ctxs = {lv: nv.Context(0, lv, 128) for lv in (2, 5, 9)}
stats = {"seed": seed, "seconds": seconds, "inputs": 0, "bytes": 0, "blocks": 0, "by_level": {}, "segments": {},
         "libbz2_decoded": 0, "failures": []}


def segment(rng, n):
    kind = rng.choice(["text", "text", "text", "text-small-alphabet", "random", "period", "runs", "zeros", "phrases",
                       "repeats", "lowalpha", "mixture"])
    stats["segments"][kind] = stats["segments"].get(kind, 0) + 1
    if kind == "text":
        a = rng.randrange(len(text) - n) if n < len(text) else 0
        return text[a:a + n]
    if kind == "text-small-alphabet":  # text folded onto 16 byte values: large groups, deep rounds
        a = rng.randrange(len(text) - n) if n < len(text) else 0
        return (np.frombuffer(text[a:a + n], dtype=np.uint8) & 15).tobytes()
    if kind == "random":
        return np.random.default_rng(rng.randrange(1 << 30)).integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "period":  # a word repeated (RLE1 leaves it alone unless its bytes repeat): near-periodic blocks
        p = rng.choice([2, 3, 5, 7, 64, 1000, 1024, 4097, 70_001])
        w = np.random.default_rng(rng.randrange(1 << 30)).integers(0, rng.choice([2, 4, 256]), p, dtype=np.uint8).tobytes()
        return (w * (n // p + 1))[:n]
    if kind == "runs":
        return cases.gen(n, rng.choice(["longruns", "shortruns"]), rng.randrange(1 << 30))
    if kind == "zeros":
        return bytes([rng.randrange(256)]) * n
    if kind == "phrases":
        return cases.phrase_groups(n, rng.randrange(1000))
    if kind == "repeats":
        return cases.repeats(n, rng.randrange(1000), copies=rng.choice([2, 6, 20]))
    if kind == "lowalpha":
        return cases.gen(n, "lowalpha", rng.randrange(1 << 30))
    return cases.mixture(rng, n)


def arbitrary(rng):
    total = rng.choice([8, 12, 16, 24, 30, 40]) * 1_000_000 + rng.randrange(1_000_000)
    parts, have = [], 0
    while have < total:
        n = min(total - have, rng.choice([50_000, 200_000, 450_000, 900_000, 1_300_000, 2_000_000, 3_700_000]) + rng.randrange(1000))
        parts.append(segment(rng, n))
        have += len(parts[-1])
    return b"".join(parts)


t0 = time.time()
pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))
pending = []


def settle(item):
    idx, lv, d, g, fut = item
    want, nb = fut.result()
    ok = g == want
    if ok and idx % 4 == 0:
        ok = bz2.decompress(g) == d
        stats["libbz2_decoded"] += 1
    stats["blocks"] += nb
    if not ok:
        stats["failures"].append({"input": idx, "level": lv, "len": len(d)})
        open(f"gpurun_out/fuzz_batches_fail_{idx}.bin", "wb").write(d)


def oracle_job(d, lv):
    s, infos = po.encode(d, lv, want_blocks=True)
    return s, len(infos)


while time.time() - t0 < seconds and len(stats["failures"]) < 3:
    lv = rng.choice([9, 9, 9, 5, 2])
    d = arbitrary(rng)
    try:
        g = ctxs[lv].encode(d)
    except Exception as e:  # noqa: BLE001
        stats["failures"].append({"input": stats["inputs"], "level": lv, "len": len(d), "error": repr(e)})
        open(f"gpurun_out/fuzz_batches_fail_{stats['inputs']}.bin", "wb").write(d)
        g = None
    if g is not None:
        pending.append((stats["inputs"], lv, d, g, pool.submit(oracle_job, d, lv)))
    stats["inputs"] += 1
    stats["bytes"] += len(d)
    stats["by_level"][str(lv)] = stats["by_level"].get(str(lv), 0) + 1
    while len(pending) > 12:  # (bounded memory: the oracle is the slow side)
        settle(pending.pop(0))
for item in pending:
    settle(item)
stats["elapsed_s"] = round(time.time() - t0, 1)
stats["ok"] = not stats["failures"]
json.dump(stats, open(out, "w"), indent=1)
print(json.dumps(stats))
sys.exit(0 if stats["ok"] else 1)
