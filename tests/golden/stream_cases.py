"""Inputs of the model-derived whole-stream fixtures (model_streams.json): name -> (level, bytes).
Deterministic: the generator script and the tests both build the inputs from here; the fixture pins their
SHA-256 as well, so a drifting generator shows up as such and not as a parity failure."""
import numpy as np

from tests import cases


def _geometric(nsym, ratio, n, seed):
    rng = np.random.default_rng(seed)
    p = np.array([ratio ** -k for k in range(nsym)])
    p /= p.sum()
    return rng.choice(np.arange(nsym, dtype=np.uint8) + 65, size=n, p=p).tobytes()


def _runs_255(seed):
    """runs of 255 / 256 / 510 / 765 / 1020 and their neighbours back to back, between short literals"""
    rng = np.random.default_rng(seed)
    out = bytearray()
    lens = [255, 256, 510, 765, 1020, 254, 257, 509, 511, 764, 766, 4, 3, 5, 259, 260]
    prev = -1
    for k in range(400):
        b = int(rng.integers(0, 5))
        if b == prev:
            b = (b + 1) % 5
        prev = b
        out += bytes([b]) * lens[int(rng.integers(0, len(lens)))]
        if rng.random() < 0.3:
            out += bytes(rng.integers(5, 60, int(rng.integers(1, 9)), dtype=np.uint8))
            prev = -1
    return bytes(out)


def _enwik_like(n, seed):
    from banzai_amd import corpus
    return corpus.enwik_synthetic(n, seed=seed).tobytes()


# multi-kilobyte inputs for the reference's debug/rle1.py (unbounded RLE1): name -> bytes, all shorter than a level-9 block
RLE1_LARGE = {
    "runs-255-a": lambda: _runs_255(11),
    "runs-255-b": lambda: _runs_255(12),
    "runs-255-c": lambda: _runs_255(13),
    "shortruns-200k": lambda: cases.gen(200_000, "shortruns", 21),
    "longruns-200k": lambda: cases.gen(200_000, "longruns", 22),
    "same-500k": lambda: cases.gen(500_000, "same", 23),
}


CASES = {
    # three tables (258 symbols), two blocks at level 1
    "random-3-tables-L1": lambda: (1, cases.gen(120_000, "random", 4)),
    # block cuts at M-1 = 99,998 RLE1 bytes (the 4th literal of a run needs its count byte, lib/rle.rs:179-203)
    "cut-at-M-1-first-block-L1": lambda: (1, cases.gen(330_000, "shortruns", 1)),
    "cut-at-M-1-second-block-L1": lambda: (1, cases.gen(330_000, "shortruns", 5)),
    # code lengths beyond 17 on the first attempt: build_table_from_freqs doubles `scaling` twice (lib/huffman.rs:293-296)
    "rescale-loop-L9": lambda: (9, _geometric(30, 1.618, 400_000, 5)),
    # RLE1 chunk limits back to back
    "runs-255-chunks-L1": lambda: (1, _runs_255(3)),
    # natural-text-like, several blocks, level 2
    "text-3-blocks-L2": lambda: (2, _enwik_like(450_000, 3)),
    # a word repeated (identical rotations: descending-index tie rule) with a ragged end
    "periodic-L1": lambda: (1, (b"banzai!" * 9000)[:60_001]),
}
