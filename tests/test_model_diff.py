"""The committed record of the differential fuzz of the two restatements of the reference (tests/golden/model_diff.py:
pymodel.py vs oracle/banzai_oracle.c, >= 5,000 small inputs at every level and >= 200 level-1 inputs that cross a block
cut) says "no mismatch", and a slice of it, re-run here, still does and still hashes to the recorded digests."""
import json
import os

from tests.golden import model_diff as md

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record():
    with open(os.path.join(ROOT, "tests", "golden", "model_diff.json")) as f:
        return json.load(f)


def test_committed_record_is_clean_and_wide():
    rec = _record()["full"]
    assert rec["mismatches"] == []
    assert rec["small"] >= 5000 and rec["cut"] >= 200
    assert set(rec["levels"]) == {str(k) for k in range(1, 10)}      # every level saw small inputs
    assert rec["blocks_in_cut_cases"] >= 2 * rec["cut"]               # every cut input really crossed a cut
    assert rec["cuts_at_M_minus_1"] >= 20                             # the "4th literal needs its count" rule fired


def test_slice_reproduces():
    want = _record()["slice"]
    got = md.run(range(0, md.N_SMALL, 26), range(0, md.N_CUT, 30))
    assert got["mismatches"] == []
    assert got["oracle_stream_crcs_sha256"] == want["oracle_stream_crcs_sha256"]
    assert got["cut_blocks_sha256"] == want["cut_blocks_sha256"]
    assert got["blocks_in_cut_cases"] == want["blocks_in_cut_cases"]
