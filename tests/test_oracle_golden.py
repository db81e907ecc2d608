"""CPU: pin the oracle (oracle/banzai_oracle.c) to every known answer the reference holds for the
path, to the vectors produced by the reference's own debug scripts, and to libbz2 as decoder."""
import bz2
import json
import os
import random

import numpy as np
import pytest

from tests import cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_bwt_kat(oracle):
    """reference lib/bwt.rs:758-772"""
    t = b"He served fire and smoke; these denizens of the fields served vegetation, weather, frost, and sun."
    b, ptr, hb = oracle.bwt(t)
    assert b == b"e,eed,sesddf;d,trnne.  etenne lrshHkwvvvidzhsshgo   etttftfnoesouaaee mireifeende   o se a asrr  i"
    assert ptr == 20
    assert sorted(np.flatnonzero(hb)) == sorted(set(t))


def test_mtf_kat(oracle):
    """reference lib/mtf.rs:139-158 (dead test in the crate, vector from Joe Tsai's bzip2 spec)"""
    test = [153, 45, 45, 38, 135, 179, 26, 154, 165, 170, 170, 170, 170, 18, 109, 240, 174, 150, 87, 164, 30, 30,
            30, 30, 30, 30, 30, 148, 190, 10, 60, 13, 13, 13, 13, 13, 6, 81, 200, 13, 225, 32, 17, 43, 22, 179, 13,
            13, 17, 236, 236, 236, 236, 236, 236, 236, 121, 211, 2, 211, 185, 54, 16] + [5] * 22 + [50] + [5] * 22 + [40]
    expected = [27, 17, 0, 15, 25, 33, 15, 29, 31, 32, 0, 0, 17, 28, 40, 34, 33, 31, 34, 25, 1, 1, 34, 36, 23, 33, 25,
                1, 0, 25, 34, 37, 4, 39, 32, 31, 34, 33, 26, 7, 0, 5, 40, 1, 1, 38, 40, 34, 2, 40, 40, 38, 38, 0, 1,
                1, 0, 40, 2, 0, 1, 1, 0, 40, 41]
    hb = np.zeros(256, np.uint8)
    hb[list(set(test))] = 1
    syms, freqs, ns = oracle.mtf_and_rle(bytes(test), hb)
    assert list(syms) == expected
    assert ns == len(set(test)) + 2
    assert freqs[ns - 1] == 1 and int(freqs.sum()) == len(expected)


def test_bitsink_kat(oracle):
    """reference lib/out.rs:107-133"""
    out = oracle.bitsink_run([(0, 6, 3), (1, 200, 8), (0, 0, 1), (3, 4, 0), (0, 1, 7)], bytes([0xCA, 0xFE, 0xBA, 0xBE]))
    assert out == bytes([0xD9, 0x0C, 0xAF, 0xEB, 0xAB, 0xE0, 0x20])


def test_crc_check_value(oracle):
    """CRC-32/BZIP2 catalogue check value (crc 3.0.0 / crc-catalog 2.1.0, lib/crc32.rs:31-48)"""
    assert oracle.crc32(b"123456789") == 0xFC891918
    assert oracle.crc32(b"") == 0


def test_reference_debug_vectors(oracle):
    """vectors produced by running the reference's debug/bwt.py and debug/rle1.py (gen_fixtures.py)"""
    v = json.load(open(os.path.join(GOLDEN, "ref_debug_vectors.json")))
    assert len(v["bwt"]) >= 50 and len(v["rle1"]) >= 50
    for c in v["bwt"]:
        b, ptr, _ = oracle.bwt(c["input"].encode())
        assert b.decode() == c["bwt"] and ptr == c["ptr"], c["input"][:40]
    for c in v["rle1"]:
        d = bytes.fromhex(c["input_hex"])
        r, _, used = oracle.rle_one(d, 9)
        assert r.hex() == c["rle1_hex"] and used == len(d)


def test_golden_streams(oracle):
    """whole streams: the empty stream follows from lib/lib.rs:18-22,66-70 alone; the others were
    derived independently at survey time from a literal Python restatement."""
    g = json.load(open(os.path.join(GOLDEN, "streams.json")))
    for c in g["streams"]:
        data = bytes.fromhex(c["input_hex"]) if "input_hex" in c else bytes([c["fill"]]) * c["count"]
        s = oracle.encode(data, c["level"])
        assert s.hex() == c["stream_hex"], c["name"]
        assert bz2.decompress(s) == data


def test_sais_matches_definition(oracle):
    rng = random.Random(1)
    for _ in range(300):
        n = rng.choice([2, 3, 4, 5, 7, 16, 33, 100, 1000, 4000])
        sig = rng.choice([1, 2, 3, 4, 16, 256])
        d = bytes(rng.randrange(sig) for _ in range(n))
        if rng.random() < 0.3:
            w = d[:rng.randint(1, max(1, n // 3))]
            d = (w * (n // len(w) + 1))[:n]
        a, b = oracle.bwt(d), oracle.bwt(d, naive=True)
        assert a[0] == b[0] and a[1] == b[1]


@pytest.mark.parametrize("mode", cases.MODES)
def test_libbz2_roundtrip_level1(oracle, mode):
    """the reference's fuzz/fuzz_targets/round_trip.rs check: libbz2 must reproduce the input
    (this also pins block CRCs, the stream CRC, headers, tables and the block split)"""
    for n in cases.SIZES_L1:
        d = cases.gen(n, mode, 3)
        assert bz2.decompress(oracle.encode(d, 1)) == d, (mode, n)


def test_libbz2_roundtrip_level9(oracle):
    for mode in ("random", "text", "longruns"):
        d = cases.gen(1_800_001, mode, 5)
        s, blocks = oracle.encode(d, 9, want_blocks=True)
        assert bz2.decompress(s) == d
        assert sum(b.in_len for b in blocks) == len(d)
        assert all(b.rle_len <= 899_999 for b in blocks)


def test_config1_plumbing(oracle):
    """BASELINE.json configs[0]: level 1, 1 MB of 0x00 -> one block, 50-byte stream"""
    s, blocks = oracle.encode(b"\0" * 1_000_000, 1, want_blocks=True)
    assert len(s) == 50 and len(blocks) == 1 and blocks[0].rle_len == 19610
    assert bz2.decompress(s) == b"\0" * 1_000_000


def test_huffman_table_quirks(oracle):
    """2 tables for <= 199 symbols else 3; all lengths in 1..17"""
    for n, sig in ((5000, 10), (300000, 256)):
        d = np.random.default_rng(n).integers(0, sig, n, dtype=np.uint8).tobytes()
        b, _, hb = oracle.bwt(d)
        syms, freqs, ns = oracle.mtf_and_rle(b, hb)
        _, bits, tables = oracle.huffman_block(syms, ns, freqs)
        assert tables.shape[0] == (2 if ns <= 199 else 3)
        assert tables[:, :ns].min() >= 1 and tables[:, :ns].max() <= 17
        assert bits > 0


def test_model_streams_pin_the_oracle(oracle):
    """whole streams from the independent Python restatement (tests/golden/pymodel.py, run by gen_streams.py in the
    build container): three tables, the `scaling <<= 1` loop (lib/huffman.rs:293-296), block cuts at M-1
    (lib/rle.rs:179-203), 255-chunk runs back to back, identical rotations"""
    import hashlib
    from tests.golden import stream_cases
    v = json.load(open(os.path.join(GOLDEN, "model_streams.json")))["cases"]
    assert set(v) == set(stream_cases.CASES)
    assert any(c["tables"] == 3 for c in v.values()) and any(c["rescaled_to"] > 1 for c in v.values())
    assert any(r == 100_000 * c["level"] - 2 for c in v.values() for _, r in c["blocks_consumed_rle"][:-1])
    for name, c in v.items():
        level, data = stream_cases.CASES[name]()
        assert level == c["level"] and len(data) == c["input_len"], name
        assert hashlib.sha256(data).hexdigest() == c["input_sha256"], name
        got, infos = oracle.encode(data, level, want_blocks=True)
        assert len(got) == c["stream_len"] and hashlib.sha256(got).hexdigest() == c["stream_sha256"], name
        assert [(int(b.in_len), int(b.rle_len)) for b in infos] == [tuple(x) for x in c["blocks_consumed_rle"]], name


def test_reference_rle1_large_vectors(oracle):
    """multi-kilobyte outputs of the reference's debug/rle1.py (digests; inputs from stream_cases.RLE1_LARGE)"""
    import hashlib
    from tests.golden import stream_cases
    v = json.load(open(os.path.join(GOLDEN, "ref_rle1_large.json")))["cases"]
    assert set(v) == set(stream_cases.RLE1_LARGE)
    for name, c in v.items():
        d = stream_cases.RLE1_LARGE[name]()
        assert len(d) == c["input_len"] and hashlib.sha256(d).hexdigest() == c["input_sha256"], name
        r, _, used = oracle.rle_one(d, 9)  # every output is shorter than a level-9 block: the bound never bites
        assert used == len(d) and len(r) == c["rle1_len"] and hashlib.sha256(r).hexdigest() == c["rle1_sha256"], name


def test_oracle_round_trips_under_sanitizers():
    """`make -C oracle san`: the oracle and the in-repo decoder built with -fsanitize=address,undefined, 300 seeded
    encode -> decode round trips over empty / tiny / run-heavy / periodic / random inputs at levels 1-9
    (the reference's fuzz/fuzz_targets/round_trip.rs:8-22 loop; GPU sanitizers do not exist on this pool)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "-s", "san"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "300 round trips clean" in r.stdout


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz_corpus():
    import zipfile
    with zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", "fuzz_corpus.zip")) as z:
        return [(n, z.read(n)) for n in sorted(z.namelist())]


def test_coverage_guided_corpus_round_trips_and_agrees_with_the_second_model(oracle):
    """the minimised corpus of the coverage-guided fuzzing session of the oracle (oracle/covfuzz.c, scripts/covfuzz.sh;
    the reference fuzzes the same property with libFuzzer, fuzz/fuzz_targets/round_trip.rs:8-22): every input round-trips
    through the strict decoder and libbz2, the small ones also give the same stream in the independent Python model"""
    import bz2
    import json
    from tests.golden import pymodel
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "fuzz_corpus.json")))
    items = _fuzz_corpus()
    assert len(items) == rec["kept_inputs"] >= 100 and rec["round_trip_failures"] == 0 and rec["executions"] >= 100_000
    levels, modelled = set(), 0
    for name, blob in items:
        level, data = 1 + blob[0] % 9, blob[1:]
        levels.add(level)
        stream = oracle.encode(data, level)
        assert oracle.decode(stream, cap=len(data) + 64) == data, name
        assert bz2.decompress(stream) == data, name
        if len(data) <= 1500:
            assert pymodel.encode(data, level) == stream, name
            modelled += 1
    assert levels == set(range(1, 10)) and modelled >= 30
