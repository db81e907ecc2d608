"""Seeded inputs shared by the CPU and GPU tests (edge cases the reference's fuzzers reach:
empty and ragged inputs, block-boundary offsets, run lengths around 4 / 255 / 256, periodic blocks)."""
import random

import numpy as np


def gen(n, mode, seed):
    rng = random.Random(seed * 1000003 + n * 7 + sum(mode.encode()))
    if mode == "random":
        return np.random.default_rng(seed + n).integers(0, 256, n, dtype=np.uint8).tobytes()
    if mode == "lowalpha":
        return np.random.default_rng(seed + n).integers(0, 3, n, dtype=np.uint8).tobytes()
    if mode == "shortruns":
        d = bytearray()
        while len(d) < n:
            d += bytes([rng.randrange(3)]) * rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8])
        return bytes(d[:n])
    if mode == "longruns":
        d = bytearray()
        while len(d) < n:
            d += bytes([rng.randrange(2)]) * rng.choice([1, 3, 4, 5, 254, 255, 256, 257, 258, 259, 260, 509, 510,
                                                         511, 1000, 70000])
        return bytes(d[:n])
    if mode == "text":
        return (b"It was the best of times, it was the worst of times, it was the age of wisdom. " * (n // 70 + 1))[:n]
    if mode == "same":
        return bytes([rng.randrange(256)]) * n
    if mode == "periodic":
        w = bytes(rng.randrange(4) for _ in range(rng.choice([1, 2, 3, 5, 7, 64])))
        return (w * (n // len(w) + 1))[:n]
    raise ValueError(mode)


MODES = ["random", "lowalpha", "shortruns", "longruns", "text", "same", "periodic"]

# sizes around the level-1 block boundary (M = 99,999) and tiny / ragged inputs
SIZES_L1 = [0, 1, 2, 3, 4, 5, 49, 50, 51, 63, 64, 65, 127, 128, 129, 255, 256, 257, 4095, 4096, 4097, 8191, 8192,
            99998, 99999, 100000, 100001, 250000]
SIZES_L9 = [1, 1000, 899999, 900000, 1800001]


def boundary_cases(M=99999):
    """Inputs whose runs straddle the block budget in every residue the cut rule distinguishes."""
    out = []
    for pre in range(M - 8, M + 3):
        for runlen in (3, 4, 5, 6, 255, 256, 259, 600):
            rng = np.random.default_rng(pre * 31 + runlen)
            body = rng.integers(1, 200, pre, dtype=np.uint8)
            # forbid accidental runs in the literal part
            body[1:][body[1:] == body[:-1]] += 1
            tail = rng.integers(1, 200, 3000, dtype=np.uint8)
            out.append(body.tobytes() + bytes([250]) * runlen + tail.tobytes())
    return out


def repeats(n, seed, copies=6):
    """text with verbatim repeated passages: long pair-groups that live in the BWT's TAIL rounds"""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, rng.integers(2, 9)).astype(np.uint8)) for _ in range(500)]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(0, 500))] + b" "
    out = bytearray(out[:n])
    for _ in range(copies):
        ln = int(rng.integers(50, max(51, n // 8)))
        a = int(rng.integers(0, max(1, n - 2 * ln)))
        b = int(rng.integers(a + ln, max(a + ln + 1, n - ln)))
        out[b:b + ln] = out[a:a + ln]
    return bytes(out)


def phrase_groups(n, seed, counts=(700, 1500, 2800)):
    """random lower-case text with a few 40-byte phrases pasted in hundreds to thousands of times:
    suffix groups of those sizes survive to depth 32 (too large for the TAIL window), then dissolve"""
    rng = np.random.default_rng(seed)
    out = rng.integers(97, 123, n, dtype=np.uint8)
    for k, c in enumerate(counts):
        phrase = rng.integers(65, 91, 40, dtype=np.uint8)  # upper case: never occurs by chance
        pos = rng.choice(max(1, n - 50), size=min(c, max(1, n // 60)), replace=False)
        for p0 in pos:
            out[p0:p0 + 40] = phrase
    return out.tobytes()


def mixture(rng, max_len):
    """one random input built from random segments (the shapes the reference's fuzzers stumble on: runs
    around the RLE1 thresholds, periodic pieces, noise, text, single bytes), total length <= max_len"""
    out = bytearray()
    target = rng.randrange(0, max_len + 1)
    while len(out) < target:
        kind = rng.randrange(7)
        ln = min(target - len(out), rng.choice([1, 2, 3, 5, 50, 255, 256, 1000, 5000, 40000]))
        if kind == 0:
            out += bytes([rng.randrange(256)]) * ln
        elif kind == 1:
            w = bytes(rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 4, 5, 7, 16, 255, 256, 257])))
            out += (w * (ln // len(w) + 1))[:ln]
        elif kind == 2:
            out += bytes(rng.randrange(256) for _ in range(min(ln, 3000)))
        elif kind == 3:
            out += (b"the quick brown fox jumps over the lazy dog; " * (ln // 40 + 1))[:ln]
        elif kind == 4:
            b = rng.randrange(256)
            for _ in range(min(ln, 400) // 4 + 1):
                out += bytes([b]) * rng.choice([3, 4, 5, 254, 255, 256, 259]) + bytes([(b + 1) & 255])
        elif kind == 5:
            out += bytes(rng.choice(b"ab") for _ in range(min(ln, 2000)))
        else:
            out += out[-min(len(out), ln):] if out else b"x"
    return bytes(out[:max_len])
