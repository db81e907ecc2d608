"""The rule behind period_detect / period_expand (banzai_amd/csrc/bwt.hip): the rotations of a near-periodic block
S = w^k w[0:r] (|w| = p its minimal period, 0 < r < p) are ordered as the rotations of S' = w^m w[0:r] are, with the k - m
extra rotations of every phase inserted next to rotation c of S' -- in front of it in ascending order if a rotation that keeps
reading (phase r) sorts before one that has wrapped (phase 0), behind it in descending order otherwise.  Checked here against
a naive sort of all rotations (the definition the reference's debug/bwt.py uses: reference lib/bwt.rs:564-573 sorts S||S)."""
import random


def sa_rot(s):
    n = len(s)
    d = s + s
    return sorted(range(n), key=lambda i: d[i:i + n])


def minimal_period(s):
    n = len(s)
    for p in range(1, n + 1):
        if all(s[u] == s[u + p] for u in range(n - p)):
            return p
    return n


def expand(S, p, m):
    n = len(S)
    k, r = divmod(n, p)
    npr = m * p + r
    D = n - npr
    sa_p = sa_rot(S[:npr])
    rank_p = [0] * npr
    for pos, i in enumerate(sa_p):
        rank_p[i] = pos
    winf = S[:p] * 3
    asc = winf[r:r + p] < winf[0:p]
    before = {rank_p[c]: c for c in range(p)} if asc else {}
    after = {} if asc else {rank_p[c]: c for c in range(p)}
    out = []
    for pos, i in enumerate(sa_p):
        if pos in before:
            out += [before[pos] + j * p for j in range(k - m)]
        out.append(i + D)
        if pos in after:
            out += [after[pos] + j * p for j in range(k - m - 1, -1, -1)]
    return out


def test_shortened_block_expands_to_the_full_order():
    rng = random.Random(20261003)
    done = 0
    while done < 1500:
        sigma = rng.choice([2, 2, 3, 4])
        p = rng.randint(2, 24)
        w = [rng.randrange(sigma) for _ in range(p)]
        m = rng.choice([3, 4, 8])
        k = rng.randint(m + 1, m + 5)
        r = rng.randint(1, p - 1)
        S = bytes(w * k + w[:r])
        if minimal_period(S) != p:
            continue
        assert expand(S, p, m) == sa_rot(S), (S, p, m)
        done += 1
