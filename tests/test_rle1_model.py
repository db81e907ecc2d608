"""CPU: the closed-form block split used by the HIP RLE1 stage equals the reference state machine
(oracle.rle_one, lib/rle.rs:102-253) -- randomized multi-block inputs plus every boundary residue."""
import pytest

from tests import cases
from tests.rle_model import split


def oracle_split(oracle, data, level):
    off, res = 0, []
    while off < len(data):
        r, _, used = oracle.rle_one(data[off:], level)
        res.append((off, used, len(r)))
        off += used
    return res


@pytest.mark.parametrize("mode", cases.MODES)
def test_model_equals_state_machine(oracle, mode):
    for n in (1, 5, 1000, 99998, 99999, 100000, 100005, 250000, 400001):
        for seed in (1, 2):
            d = cases.gen(n, mode, seed)
            assert split(d, 99999) == oracle_split(oracle, d, 1), (mode, n, seed)


def test_model_boundary_residues(oracle):
    """runs of 3,4,5,6,255,256,259,600 equal bytes placed at every offset around the block budget:
    covers 'block ends at M-1 with three literals' (lib/rle.rs:179-182, :193-203)"""
    saw_short_block = False
    for d in cases.boundary_cases():
        m, o = split(d, 99999), oracle_split(oracle, d, 1)
        assert m == o
        saw_short_block |= any(b[2] == 99998 for b in o[:-1])
    assert saw_short_block


def test_model_level9(oracle):
    d = cases.gen(2_000_000, "longruns", 9)
    assert split(d, 899_999) == oracle_split(oracle, d, 9)
