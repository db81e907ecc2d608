"""Deterministic synthetic inputs for the configs of BASELINE.json (there is no network, so
enwik8/enwik9 themselves are only used when a local copy is pointed to by $BZH_ENWIK8).

  xorshift_bytes      C2: uniform random bytes (xorshift64*, seed 0x9E3779B97F4A7C15), SURVEY 8(d)
  enwik_synthetic     C3/C4: "enwik8-synthetic" -- Zipfian word + reusable-phrase model with wiki-ish
                      markup tokens and a few percent of long verbatim repeats (real wiki dumps contain
                      duplicated passages; they are what drives the number of prefix-doubling rounds)
  pathological        C5: long single-byte runs + periodic repeats
"""
import os

import numpy as np

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_P = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0,
                      1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
_MARKUP = [b"<page>\n", b"</page>\n", b"<title>", b"</title>\n", b"<text xml:space=\"preserve\">", b"</text>\n",
           b"[[", b"]]", b"{{", b"}}", b"&quot;", b"&amp;", b"==", b"'''", b"\n\n", b"\n* ", b"|", b"<ref>",
           b"</ref>", b"[[Category:", b"<id>", b"</id>\n", b"<timestamp>2006-03-0", b"http://www."]
_PUNCT = [b". ", b", ", b"; ", b": ", b".\n", b" (", b") ", b" - ", b"1", b"2", b"19", b"200", b"0", b"The ", b"In "]


def xorshift_bytes(n, seed=0x9E3779B97F4A7C15):
    """Top byte of xorshift64* outputs; the same generator is trivial to restate in C."""
    out = np.empty(n, dtype=np.uint8)
    x = np.uint64(seed)
    # vectorised in lanes: 4096 independent streams seeded by splitting the seed sequence
    lanes = 4096
    st = (np.arange(1, lanes + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ x
    mul = np.uint64(0x2545F4914F6CDD1D)
    pos = 0
    with np.errstate(over="ignore"):
        while pos < n:
            st ^= st >> np.uint64(12)
            st ^= st << np.uint64(25)
            st ^= st >> np.uint64(27)
            v = ((st * mul) >> np.uint64(56)).astype(np.uint8)
            k = min(lanes, n - pos)
            out[pos:pos + k] = v[:k]
            pos += k
    return out


def _vocab(rng, nwords, nphrases):
    """Token table: single words (Zipf rank = index) followed by multi-word phrases."""
    lens = np.clip(1 + rng.poisson(4.2, nwords), 1, 14)
    total = int(lens.sum())
    letters = rng.choice(_LETTERS, size=total, p=_LETTER_P / _LETTER_P.sum())
    words = []
    pos = 0
    for ln in lens:
        words.append(letters[pos:pos + ln].tobytes() + b" ")
        pos += ln
    # sprinkle markup and punctuation tokens through the frequent ranks
    for k, tok in enumerate(_PUNCT):
        words.insert(3 + 4 * k, tok)
    for k, tok in enumerate(_MARKUP):
        words.insert(40 + 23 * k, tok)
    nw = len(words)
    pw = 1.0 / (np.arange(nw) + 2.7) ** 1.07
    cdf_w = np.cumsum(pw / pw.sum())
    # phrases: 2..9 words drawn from the same Zipf law; reused phrases are what gives natural text
    # its long common prefixes (and the BWT its deep sorts)
    plen = rng.integers(2, 10, nphrases)
    ids = np.minimum(np.searchsorted(cdf_w, rng.random(int(plen.sum())), side="right"), nw - 1)
    phrases = []
    pos = 0
    for ln in plen:
        phrases.append(b"".join(words[i] for i in ids[pos:pos + ln]))
        pos += ln
    toks = words + phrases
    vlen = np.array([len(w) for w in toks], dtype=np.int64)
    voff = np.concatenate(([0], np.cumsum(vlen)[:-1]))
    vflat = np.frombuffer(b"".join(toks), dtype=np.uint8)
    return vflat, voff, vlen, nw, cdf_w


def enwik_synthetic(n, seed=20061, repeat_fraction=0.03, phrase_prob=0.4):
    """n bytes of enwik-like text (deterministic for a given (n, seed))."""
    rng = np.random.default_rng(seed)
    nphrases = 200_000
    vflat, voff, vlen, nw, cdf_w = _vocab(rng, 50000, nphrases)
    pp = 1.0 / (np.arange(nphrases) + 5.0) ** 0.9
    cdf_p = np.cumsum(pp / pp.sum())
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    chunk_toks = 400_000
    while pos < n:
        u = rng.random(chunk_toks)
        is_phrase = rng.random(chunk_toks) < phrase_prob
        ids = np.where(is_phrase,
                       nw + np.minimum(np.searchsorted(cdf_p, u, side="right"), nphrases - 1),
                       np.minimum(np.searchsorted(cdf_w, u, side="right"), nw - 1))
        lens = vlen[ids]
        ends = np.cumsum(lens)
        total = int(ends[-1])
        src = np.repeat(voff[ids] - (ends - lens), lens) + np.arange(total)
        take = min(total, n - pos)
        out[pos:pos + take] = vflat[src[:take]]
        pos += take
    # verbatim repeats: copy passages a short distance forward (lengths log-uniform in 200 .. 60000,
    # distance log-uniform up to 600 kB so that most copies share a 900 kB block with their source,
    # as templated / duplicated wiki passages do)
    budget = int(n * repeat_fraction)
    while budget > 0 and n > 200_000:
        ln = int(np.exp(rng.uniform(np.log(200), np.log(60000))))
        dist = int(np.exp(rng.uniform(np.log(ln + 1), np.log(600_000))))
        src0 = int(rng.integers(0, max(1, n - ln - dist - 1)))
        dst0 = src0 + dist
        if dst0 + ln <= n:
            out[dst0:dst0 + ln] = out[src0:src0 + ln].copy()
        budget -= ln
    return out


def _vocab_v2(rng, nwords, nphrases):
    """v2 token table: v1's Zipfian words, plus what real wiki text has and v1 lacks -- capitalised forms (names,
    sentence starts), numbers, and words in other scripts as UTF-8 (Latin-1 supplement, Greek/Cyrillic, CJK), so that
    the byte alphabet is enwik8-like (about 200 distinct values, 1-2 % of the bytes >= 0x80) instead of 57 values."""
    lens = np.clip(1 + rng.poisson(4.2, nwords), 1, 14)
    total = int(lens.sum())
    letters = rng.choice(_LETTERS, size=total, p=_LETTER_P / _LETTER_P.sum())
    words = []
    pos = 0
    for k, ln in enumerate(lens):
        w = letters[pos:pos + ln].tobytes()
        pos += ln
        r = k % 97
        if r < 9:                       # ~9 % of the vocabulary: capitalised (proper nouns keep their form)
            w = w[:1].upper() + w[1:]
        elif r == 11:                   # numbers / years
            w = str(int(rng.integers(0, 2100))).encode()
        elif r == 13:                   # another script, UTF-8
            kind = int(rng.integers(0, 3))
            lo, hi = ((0xC0, 0x100), (0x391, 0x450), (0x4E00, 0x9FA6))[kind]
            cps = rng.integers(lo, hi, size=max(1, min(int(ln), 6)))
            w = "".join(chr(int(c)) for c in cps if not (0xD7 <= c <= 0xD7)).encode("utf-8")
        words.append(w + b" ")
    for k, tok in enumerate(_PUNCT):
        words.insert(3 + 4 * k, tok)
    for k, tok in enumerate(_MARKUP):
        words.insert(40 + 23 * k, tok)
    nw = len(words)
    pw = 1.0 / (np.arange(nw) + 2.7) ** 1.07
    cdf_w = np.cumsum(pw / pw.sum())
    plen = rng.integers(2, 10, nphrases)
    ids = np.minimum(np.searchsorted(cdf_w, rng.random(int(plen.sum())), side="right"), nw - 1)
    phrases = []
    pos = 0
    for ln in plen:
        phrases.append(b"".join(words[i] for i in ids[pos:pos + ln]))
        pos += ln
    toks = words + phrases
    vlen = np.array([len(w) for w in toks], dtype=np.int64)
    voff = np.concatenate(([0], np.cumsum(vlen)[:-1]))
    vflat = np.frombuffer(b"".join(toks), dtype=np.uint8)
    return vflat, voff, vlen, nw, cdf_w


V2_PHRASE_PROB = 0.15     # calibrated so that bzip2 -9 of the 100,000,000-byte stream is 0.289 of it (enwik8: 29,008,758 bytes)
V2_REPEAT_FRACTION = 0.03


def enwik_synthetic_v2(n, seed=20061, repeat_fraction=None):
    """"enwik8-synthetic-v2": the stand-in for enwik8 the bench quotes its headline on.

    Stated model: tokens are drawn i.i.d. -- with probability V2_PHRASE_PROB a reusable phrase (2..9 words, phrase
    rank ~ Zipf exponent 0.9 over 200,000 phrases), otherwise a single word or markup token (rank ~ Zipf exponent
    1.07 over ~50,000 tokens, _vocab_v2); then V2_REPEAT_FRACTION of the bytes are overwritten by verbatim copies of
    earlier passages whose LENGTH is log-uniform in [200, 60000] bytes and whose DISTANCE to the source is
    log-uniform in (length, 600000] bytes (most copies share a 900 kB block with their source: templated and
    duplicated wiki passages are what drives the depth of the suffix sort).  Calibration target: compressed size
    0.29 +- 0.01 of the input under bzip2 -9 (enwik8's published figure 29,008,758 / 100,000,000)."""
    rng = np.random.default_rng(seed + 7_000_000)
    nphrases = 200_000
    vflat, voff, vlen, nw, cdf_w = _vocab_v2(rng, 50000, nphrases)
    pp = 1.0 / (np.arange(nphrases) + 5.0) ** 0.9
    cdf_p = np.cumsum(pp / pp.sum())
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    chunk_toks = 400_000
    while pos < n:
        u = rng.random(chunk_toks)
        is_phrase = rng.random(chunk_toks) < V2_PHRASE_PROB
        ids = np.where(is_phrase,
                       nw + np.minimum(np.searchsorted(cdf_p, u, side="right"), nphrases - 1),
                       np.minimum(np.searchsorted(cdf_w, u, side="right"), nw - 1))
        lens = vlen[ids]
        ends = np.cumsum(lens)
        total = int(ends[-1])
        src = np.repeat(voff[ids] - (ends - lens), lens) + np.arange(total)
        take = min(total, n - pos)
        out[pos:pos + take] = vflat[src[:take]]
        pos += take
    # (repeat_fraction: the bench's sensitivity rows -- how the throughput depends on the share of copied bytes, which sets
    # the depth of the suffix sort; the headline uses the calibrated V2_REPEAT_FRACTION)
    budget = int(n * (V2_REPEAT_FRACTION if repeat_fraction is None else repeat_fraction))
    while budget > 0 and n > 200_000:
        ln = int(np.exp(rng.uniform(np.log(200), np.log(60000))))
        dist = int(np.exp(rng.uniform(np.log(ln + 1), np.log(600_000))))
        src0 = int(rng.integers(0, max(1, n - ln - dist - 1)))
        dst0 = src0 + dist
        if dst0 + ln <= n:
            out[dst0:dst0 + ln] = out[src0:src0 + ln].copy()
        budget -= ln
    return out


def pathological(n, seed=5):
    """C5: 1/4 zeros, 1/4 a 1024-byte random tile repeated, 1/4 'ab' repeated, 1/4 alternating runs
    of lengths cycling {255, 256, 257, 3, 4, 5} over two byte values."""
    rng = np.random.default_rng(seed)
    q = n // 4
    parts = [np.zeros(q, dtype=np.uint8)]
    tile = rng.integers(0, 256, 1024, dtype=np.uint8)
    parts.append(np.tile(tile, q // 1024 + 1)[:q])
    parts.append(np.tile(np.frombuffer(b"ab", dtype=np.uint8), q // 2 + 1)[:q])
    cyc = [255, 256, 257, 3, 4, 5]
    reps = (n - 3 * q) // sum(cyc) + 1
    lens = np.tile(np.array(cyc), reps)
    vals = np.tile(np.array([0x41, 0x7A], dtype=np.uint8), len(lens) // 2 + 1)[:len(lens)]
    parts.append(np.repeat(vals, lens)[:n - 3 * q])
    return np.concatenate(parts)


def workload(n=100_000_000, segment=0):
    """The bench workload: real enwik8 if $BZH_ENWIK8 names a file, else enwik8-synthetic-v2 (the stand-in calibrated
    to enwik8's bzip2 -9 ratio; round 1-3's generator stays available as enwik_synthetic, "enwik8-synthetic").
    -> (uint8 array of n bytes, name)"""
    path = os.environ.get("BZH_ENWIK8")
    if path and os.path.exists(path):
        data = np.fromfile(path, dtype=np.uint8)
        if data.size >= n:
            return data[:n].copy(), "enwik8"
    return enwik_synthetic_v2(n, seed=20061 + segment), "enwik8-synthetic-v2"


def corpus_digest(data):
    """SHA-256 of a workload's bytes (the bench line carries it: image-derived corpora differ between images)"""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(data).tobytes()).hexdigest()


# ---- real (non-synthetic) bytes that ship with the image: the same files exist on every box ----------------
IMAGE_SETS = {
    "python-sources": (["/usr/lib/python3.10/**/*.py", "/usr/lib/python3/dist-packages/**/*.py"], 60_000_000),
    "shared-libs": (["/usr/lib/x86_64-linux-gnu/*.so*"], 60_000_000),
    "real-text-100MB": ([], 100_000_000),
}


# Real text / markup / source text of the image for a workload of the headline's size: documentation, licences,
# Perl pods and modules, the Python standard library, C and C++ headers -- walked in sorted order (the same files,
# hence the same bytes, on every box of this image), every file once (symlinks skipped), cut at 100,000,000 bytes.
REAL_TEXT = [("/usr/share/doc", None), ("/usr/share/common-licenses", None), ("/usr/share/perl", (".pod", ".pm", ".pl")),
             ("/usr/lib/python3.10", (".py", ".txt")), ("/usr/include", (".h", ".hpp")),
             ("/opt/rocm/include", (".h", ".hpp", ".inc")),
             ("/usr/local/lib/python3.10/dist-packages", (".py", ".md", ".txt", ".rst"))]
REAL_TEXT_SKIP = (".gz", ".xz", ".bz2", ".png", ".jpg", ".pyc", ".so", ".a", ".o", ".bin", ".pdf", ".ico", ".gif")


LAST_FILE_COUNT = {}


def _walk_text(roots, limit):
    buf = bytearray()
    nfiles = 0
    for root, exts in roots:
        for dirpath, dirnames, filenames in os.walk(root):
            dirnames.sort()
            for fn in sorted(filenames):
                if fn.endswith(REAL_TEXT_SKIP) or (exts is not None and not fn.endswith(exts)):
                    continue
                f = os.path.join(dirpath, fn)
                try:
                    if os.path.islink(f) or not os.path.isfile(f):
                        continue
                    with open(f, "rb") as fh:
                        data = fh.read()
                except OSError:
                    continue
                if b"\0" in data[:4096]:  # not text
                    continue
                buf += data
                nfiles += 1
                if len(buf) >= limit:
                    LAST_FILE_COUNT["real-text-100MB"] = nfiles
                    return np.frombuffer(bytes(buf[:limit]), dtype=np.uint8)
    LAST_FILE_COUNT["real-text-100MB"] = nfiles
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def image_corpus(name):
    """Concatenation of the image's files matching IMAGE_SETS[name], in sorted order, cut at the limit.
    -> uint8 array (possibly short or empty when the files are not there)"""
    if name == "real-text-100MB":
        return _walk_text(REAL_TEXT, 100_000_000)
    import glob
    patterns, limit = IMAGE_SETS[name]
    buf = bytearray()
    nfiles = 0
    for pat in patterns:
        for f in sorted(glob.glob(pat, recursive=True)):
            try:
                if os.path.isfile(f):  # symlinked names repeat their target's bytes (as round 1's script did)
                    with open(f, "rb") as fh:
                        buf += fh.read()
                    nfiles += 1
            except OSError:
                pass
            if len(buf) >= limit:
                LAST_FILE_COUNT[name] = nfiles
                return np.frombuffer(bytes(buf[:limit]), dtype=np.uint8)
    LAST_FILE_COUNT[name] = nfiles
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def c5_parts(n=100_000_000):
    """The four quarters of the C5 workload as separate inputs: [(name, uint8 array)]"""
    data = pathological(n)
    q = n // 4
    names = ["c5-zeros", "c5-tile1024", "c5-abab", "c5-cycling-runs"]
    return [(names[k], data[k * q:(k + 1) * q if k < 3 else n]) for k in range(4)]
