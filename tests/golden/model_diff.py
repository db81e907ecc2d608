"""Differential fuzz of the two independent restatements of the reference: tests/golden/pymodel.py (plain Python,
written from lib/*.rs) against oracle/banzai_oracle.c (C, the parity oracle).  The Rust binary cannot run in the
build image, so what pins rle_one's bounded cut (lib/rle.rs:179-203) and build_table_from_freqs' tie-breaking
(lib/huffman.rs:271-298) is that two separate readings of the source agree -- this widens that agreement from seven
streams to thousands of inputs, stage seam by stage seam.

    python tests/golden/model_diff.py            # the full run (several minutes), writes model_diff.json
    tests/test_model_diff.py                     # asserts the committed record, re-runs a slice of it

Inputs are a pure function of (kind, index): SMALL = mixtures of at most 8 kB at a random level 1..9 (whole stream,
RLE1 bytes and cut, code lengths compared); CUT = level-1 inputs of 100-330 kB of run-heavy mixtures whose RLE1
output crosses the 99,999-byte budget (every block: RLE1 bytes, bytes consumed, CRC; then, on the oracle's BWT of
that block, MTF symbols, frequencies and the code lengths of both models).
"""
import hashlib
import json
import os
import random
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cases  # noqa: E402
from tests.golden import pymodel  # noqa: E402

N_SMALL = 5200
N_CUT = 210
OUT = os.path.join(ROOT, "tests", "golden", "model_diff.json")


def small_case(k):
    rng = random.Random(0x5EED0000 + k)
    level = rng.randrange(1, 10)
    kind = rng.randrange(8)
    if kind <= 3:
        data = cases.mixture(rng, 8192)
    elif kind == 4:  # tiny alphabets: long MTF zero runs, two-table streams with few symbols
        a = rng.choice([1, 2, 3, 5])
        data = bytes(rng.randrange(a) + 65 for _ in range(rng.randrange(0, 8193)))
    elif kind == 5:  # skewed frequencies: deep Huffman trees, the rescaling loop
        ratio = rng.choice([1.3, 1.618, 2.0, 3.0])
        nsym = rng.randrange(2, 40)
        w = [ratio ** -j for j in range(nsym)]
        data = bytes(rng.choices(range(nsym), weights=w, k=rng.randrange(1, 8193)))
    elif kind == 6:  # many distinct bytes: three tables
        data = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8193)))
    else:  # runs around the RLE1 thresholds only
        out = bytearray()
        target = rng.randrange(1, 8193)
        while len(out) < target:
            out += bytes([rng.randrange(4)]) * rng.choice([1, 2, 3, 4, 5, 6, 254, 255, 256, 257, 258, 259, 260, 510, 515])
        data = bytes(out[:target])
    return level, data


def cut_case(k):
    """level 1, run-heavy, long enough for two to four blocks; the literal/run mix moves the cut through every phase"""
    rng = random.Random(0xC0750000 + k)
    target = rng.randrange(120_000, 330_001)  # RLE1 bytes, so that the budget of 99,999 is crossed one to three times
    out = bytearray()
    est = 0
    style = rng.randrange(4)
    prev = -1
    while est < target:
        b = rng.randrange(6)
        if b == prev:
            b = (b + 1) % 6
        prev = b
        if style == 0:
            ln = rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8])
        elif style == 1:
            ln = rng.choice([1, 3, 4, 5, 254, 255, 256, 257, 258, 259, 260, 509, 510, 511])
        elif style == 2:
            ln = rng.choice([4, 4, 4, 5, 3, 1, 259, 260])
        else:
            ln = rng.choice([1, 2, 3, 4, 5, 255, 256, 1000, 20000])
        out += bytes([b]) * ln
        q, r = divmod(ln, 255)
        est += 5 * q + (r if r < 4 else 5)
        if rng.random() < 0.2:
            lit = rng.randrange(1, 12)
            out += bytes(rng.randrange(6, 250) for _ in range(lit))
            est += lit
            prev = -1
    return 1, bytes(out)


def check_small(k, po):
    level, data = small_case(k)
    info = {}
    sm = pymodel.encode(data, level, info)
    so = po.encode(data, level)
    bad = []
    if sm != so:
        bad.append("stream")
    rm, cm = pymodel.rle_one(data, level)
    ro, crc, co = po.rle_one(data, level)
    if rm != ro or cm != co:
        bad.append("rle_one")
    if data and crc != pymodel.checksum(data[:co]):
        bad.append("crc")
    if rm:
        col, ptr, hb = po.bwt(rm)
        colm, ptrm = pymodel.bwt(rm)
        if colm != col or ptrm != ptr:
            bad.append("bwt")
        symm, nsm, fm = pymodel.mtf_and_rle(col, hb)
        symo, fo, nso = po.mtf_and_rle(col, hb)
        if list(symo) != list(symm) or nsm != nso or list(fo) != list(fm):
            bad.append("mtf")
        if list(po.build_table_from_freqs(nso, fo)) != list(pymodel.build_table_from_freqs(nsm, fm)):
            bad.append("code_lengths")
    return bad, so


def check_cut(k, po):
    level, data = cut_case(k)
    bad = []
    pos = 0
    blocks = 0
    cuts_at_m1 = 0
    digest = hashlib.sha256()
    while pos < len(data):
        rm, cm = pymodel.rle_one(data[pos:], level)
        ro, crc, co = po.rle_one(data[pos:], level)
        if rm != ro or cm != co:
            bad.append(f"rle_one@block{blocks}")
            break
        if crc != pymodel.checksum(data[pos:pos + co]):
            bad.append(f"crc@block{blocks}")
        if len(ro) == 100_000 * level - 2:
            cuts_at_m1 += 1
        col, ptr, hb = po.bwt(ro)
        symm, nsm, fm = pymodel.mtf_and_rle(col, hb)
        symo, fo, nso = po.mtf_and_rle(col, hb)
        if list(symo) != list(symm) or nsm != nso or list(fo) != list(fm):
            bad.append(f"mtf@block{blocks}")
        lo = po.build_table_from_freqs(nso, fo)
        if list(lo) != list(pymodel.build_table_from_freqs(nsm, fm)):
            bad.append(f"code_lengths@block{blocks}")
        digest.update(ro)
        digest.update(bytes(lo))
        pos += co
        blocks += 1
    return bad, blocks, cuts_at_m1, digest.digest()


def run(small_ids, cut_ids, verbose=False):
    from oracle import pyoracle as po
    res = {"small": len(small_ids), "cut": len(cut_ids), "mismatches": [], "blocks_in_cut_cases": 0,
           "cuts_at_M_minus_1": 0, "levels": {}}
    h_small = hashlib.sha256()
    for k in small_ids:
        bad, so = check_small(k, po)
        h_small.update(zlib.crc32(so).to_bytes(4, "little"))
        lv = small_case(k)[0]
        res["levels"][str(lv)] = res["levels"].get(str(lv), 0) + 1
        if bad:
            res["mismatches"].append({"small": k, "what": bad})
        if verbose and k % 500 == 0:
            print("small", k, len(res["mismatches"]), flush=True)
    h_cut = hashlib.sha256()
    for k in cut_ids:
        bad, blocks, m1, dg = check_cut(k, po)
        res["blocks_in_cut_cases"] += blocks
        res["cuts_at_M_minus_1"] += m1
        h_cut.update(dg)
        if bad:
            res["mismatches"].append({"cut": k, "what": bad})
        if verbose and k % 20 == 0:
            print("cut", k, len(res["mismatches"]), flush=True)
    res["oracle_stream_crcs_sha256"] = h_small.hexdigest()
    res["cut_blocks_sha256"] = h_cut.hexdigest()
    return res


if __name__ == "__main__":
    full = run(range(N_SMALL), range(N_CUT), verbose=True)
    # the slice the CPU test re-runs: every 26th small input, every 30th cut input
    sl = run(range(0, N_SMALL, 26), range(0, N_CUT, 30))
    rec = {"note": "pymodel.py (Python, from lib/*.rs) vs oracle/banzai_oracle.c on the same inputs; "
                   "not produced by the Rust binary (no toolchain in the build image)",
           "full": full, "slice": sl}
    json.dump(rec, open(OUT, "w"), indent=1)
    print(json.dumps({k: v for k, v in full.items() if k != "levels"}))
