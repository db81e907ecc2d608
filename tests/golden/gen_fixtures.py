#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference's own debug oracles in the build container.

  /root/reference/debug/bwt.py   (naive doubled-string BWT of one text line; stdin -> stdout)
  /root/reference/debug/rle1.py  (naive unbounded RLE1; imported, rle1(bytearray) -> bytes)

Only the inputs and the outputs those scripts produced are written to
tests/golden/ref_debug_vectors.json -- no reference source text.  The reference does not
exist on the GPU box; tests read the JSON only.  Re-run:  python tests/golden/gen_fixtures.py
"""
import importlib.util
import json
import os
import random
import subprocess
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def ref_bwt(line: str):
    out = subprocess.run([sys.executable, os.path.join(REF, "debug", "bwt.py")], input=line + "\n",
                         capture_output=True, text=True, check=True).stdout.split("\n")
    # prints "'" + bwt + "'", then ptr, then len
    assert out[0][0] == "'" and out[0][-1] == "'"
    return out[0][1:-1], int(out[1]), int(out[2])


def main():
    rng = random.Random(0xBA27A1)
    spec = importlib.util.spec_from_file_location("ref_rle1", os.path.join(REF, "debug", "rle1.py"))
    ref_rle1 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_rle1)

    bwt_cases = []
    lines = [
        "He served fire and smoke; these denizens of the fields served vegetation, weather, frost, and sun.",
        "a", "aa", "ab", "ba", "abab", "aaaa", "aaaaaaaaaaaaaaaa", "abcabcabcabc", "abcabcabcabca",
        "banana", "mississippi", "abracadabra abracadabra abracadabra", "zyxwvutsrqponmlkjihgfedcba",
        "the quick brown fox jumps over the lazy dog " * 4, "ab" * 33, "aab" * 21 + "a", "xyxyxyxyxyxyxyxz",
    ]
    alph = "abcdefghijklmnopqrstuvwxyz ABCDEFG,.;"
    for k in range(40):
        n = rng.choice([2, 3, 5, 8, 13, 31, 64, 100, 257, 600])
        sigma = rng.choice([1, 2, 3, 4, 8, len(alph)])
        lines.append("".join(rng.choice(alph[:sigma]) for _ in range(n)))
    for k in range(8):  # exactly / nearly periodic
        w = "".join(rng.choice("abc") for _ in range(rng.choice([1, 2, 3, 7])))
        reps = rng.choice([2, 5, 16, 40])
        lines.append(w * reps)
        lines.append(w * reps + rng.choice("abc"))
    for line in lines:
        b, ptr, n = ref_bwt(line)
        assert n == len(line)
        bwt_cases.append({"input": line, "bwt": b, "ptr": ptr})

    rle_cases = []
    datas = [bytes([7] * L) for L in (1, 2, 3, 4, 5, 6, 254, 255, 256, 257, 258, 259, 260, 509, 510, 511, 512, 513,
                                      765, 766, 1020, 1024)]
    datas += [b"abc", b"aaab", b"aaaab", b"aaaabaaaa", b"aaaaaaaabbbbbbbbcccc", bytes(range(256)),
              b"\x00" * 300 + b"\x01" * 300, b"xyz" + b"q" * 259 + b"xyz" + b"q" * 4 + b"r" * 5]
    for k in range(60):
        out = bytearray()
        for _ in range(rng.randint(1, 40)):
            v = rng.choice([0, 1, 2, 251, 255, rng.randrange(256)])
            L = rng.choice([1, 1, 1, 2, 3, 4, 5, 6, 7, 100, 254, 255, 256, 257, 258, 259, 300, 511, 700])
            out += bytes([v]) * L
        datas.append(bytes(out))
    for d in datas:
        r = ref_rle1.rle1(bytearray(d))
        rle_cases.append({"input_hex": d.hex(), "rle1_hex": bytes(r).hex()})

    # multi-kilobyte RLE1 vectors (255 / 256 / 510 / 765-runs back to back): inputs come from the deterministic
    # generators of stream_cases.py, so only digests are stored
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.golden import stream_cases
    large = {}
    for name, make in stream_cases.RLE1_LARGE.items():
        d = make()
        r = bytes(ref_rle1.rle1(bytearray(d)))
        large[name] = {"input_len": len(d), "input_sha256": hashlib.sha256(d).hexdigest(), "rle1_len": len(r),
                       "rle1_sha256": hashlib.sha256(r).hexdigest()}
    with open(os.path.join(HERE, "ref_rle1_large.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_fixtures.py", "source": "reference debug/rle1.py on the inputs of "
                   "tests/golden/stream_cases.py:RLE1_LARGE", "cases": large}, f, indent=1)
    print(len(large), "large rle1 cases")

    with open(os.path.join(HERE, "ref_debug_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_fixtures.py", "source": "reference debug/bwt.py + debug/rle1.py",
                   "bwt": bwt_cases, "rle1": rle_cases}, f, indent=0)
    print(len(bwt_cases), "bwt cases,", len(rle_cases), "rle1 cases")


if __name__ == "__main__":
    main()
