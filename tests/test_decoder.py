"""The in-repo bzip2 decoder (oracle/bz2_decode.c, SURVEY 8f row f3: verification tooling that makes
the round-trip property independent of the system's libbz2; the reference checks itself the same way
with libbz2 in fuzz/fuzz_targets/round_trip.rs:8-22).  Pinned here against libbz2 in BOTH directions:
it decodes what libbz2's own encoder writes (different table counts / selectors than banzai's), it
agrees with libbz2's decoder on the oracle's streams, and it rejects what libbz2 rejects."""
import bz2
import json
import os

import pytest

from tests import cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_decodes_golden_streams(oracle):
    v = json.load(open(os.path.join(GOLDEN, "streams.json")))["streams"]
    assert v
    for c in v:
        s = bytes.fromhex(c["stream_hex"])
        assert oracle.decode(s) == bz2.decompress(s), c.get("name")
        if c.get("input_hex") is not None:
            assert oracle.decode(s) == bytes.fromhex(c["input_hex"]), c.get("name")


@pytest.mark.parametrize("mode", cases.MODES)
def test_agrees_with_libbz2_on_oracle_streams(oracle, mode):
    for n in (1, 2, 49, 50, 51, 4096, 99_999, 100_000, 250_000):
        d = cases.gen(n, mode, 31)
        s = oracle.encode(d, 1)
        assert oracle.decode(s) == d == bz2.decompress(s), (mode, n)


def test_decodes_libbz2_encoder_output(oracle):
    """streams written by an independent encoder: up to 6 tables, real selector choices, level 1..9"""
    for mode in ("text", "random", "longruns", "lowalpha"):
        d = cases.gen(300_000, mode, 5) + cases.repeats(200_000, 9)
        for level in (1, 5, 9):
            assert oracle.decode(bz2.compress(d, level)) == d, (mode, level)
    assert oracle.decode(bz2.compress(b"", 9)) == b""


def test_level9_multi_block_and_long_runs(oracle):
    d = cases.gen(1_900_000, "text", 3) + b"\xff" * 3_000_000 + cases.gen(50_000, "shortruns", 3)
    assert oracle.decode(oracle.encode(d, 9)) == d


def test_rejects_damaged_streams(oracle):
    d = cases.gen(120_000, "text", 7)
    s = bytearray(oracle.encode(d, 1))

    def status(stream):
        with pytest.raises(oracle.DecodeError) as e:
            oracle.decode(bytes(stream))
        return e.value.status

    assert status(b"BZx9" + s[4:]) == -1
    assert status(s[:len(s) // 2]) == -2                 # truncated
    assert status(s + b"\0") == -8                       # trailing byte
    bad = bytearray(s)
    bad[12] ^= 0x40                                      # inside the first block's stored CRC
    assert status(bad) == -4
    bad = bytearray(s)
    bad[-2] ^= 0x01                                      # stream CRC (last 32 bits before the padding)
    assert status(bad) in (-5, -8, -3)
    payload = bytearray(s)
    payload[len(s) // 2] ^= 0x10                         # somewhere in a Huffman payload
    with pytest.raises(oracle.DecodeError):
        oracle.decode(bytes(payload))
    with pytest.raises(Exception):                       # libbz2 rejects it too
        bz2.decompress(bytes(payload))
