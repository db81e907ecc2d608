"""A second, independent restatement of the reference's encode path, in plain Python -- test infrastructure.

Written from the Rust sources (reference lib/{lib,rle,crc32,bwt,mtf,huffman,out}.rs, cited per function), NOT from
oracle/banzai_oracle.c: different language, different data structures (Python lists / ints, numpy only for the
suffix sort, zlib for the CRC), so that an error of reading in one restatement shows up as a mismatch with the
other.  It is slow (seconds per block) and is only run in the build container by gen_streams.py, which stores
SHA-256 digests of its streams as fixtures (tests/golden/model_streams.json); the tests then hold the C oracle and
the HIP path to those digests.  The BWT is the definition (sort the rotations, identical rotations by descending
start index, lib/bwt.rs:564-573 sorts S||S), not SA-IS.
"""
import zlib

import numpy as np


# ---- lib/out.rs:7-105: MSB-first bit sink --------------------------------------------------------------------------
class Bits:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        assert nbits == 0 or 0 <= value < (1 << nbits)
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xFF)
            self.acc &= (1 << self.n) - 1

    def put_bytes(self, bs):
        for b in bs:
            self.put(b, 8)

    def close(self):  # lib/out.rs:22-28: partial byte padded with zeros
        if self.n:
            self.buf.append((self.acc << (8 - self.n)) & 0xFF)
            self.acc = self.n = 0
        return bytes(self.buf)


# ---- lib/crc32.rs:31-48 ------------------------------------------------------------------------------------------------
_REV = bytes(int(f"{b:08b}"[::-1], 2) for b in range(256))


def checksum(raw):
    chk = zlib.crc32(bytes(raw).translate(_REV)) & 0xFFFFFFFF  # CRC_32_ISO_HDLC of the bit-reversed bytes
    return int(f"{chk:032b}"[::-1], 2)


# ---- lib/rle.rs:102-253 (reader = a slice: everything is available at once) ---------------------------------------------
def rle_one(raw, level):
    """raw: the bytes not yet encoded -> (rle output, consumed)"""
    n = len(raw)
    if n == 0:
        return b"", 0
    bound = 100_000 * level - 1
    out = bytearray()

    def push(x):
        nonlocal bound
        assert bound > 0
        out.append(x)
        bound -= 1

    floor = 0
    i = 0
    b = raw[0]
    while True:
        if bound == 0:
            break
        if bound == 1:
            push(b)
            i += 1
            break
        push(b)
        avail = min(n - i, 256)  # margin_call (:58-91) with the whole input in `raw`
        if avail == 1:
            i += 1
            break
        if avail == 2:
            push(raw[i + 1])
            i += 2
            break
        hop = raw[i + 2]
        push(raw[i + 1])
        if b == hop and b == raw[i + 1]:
            run = False
            if i > floor and b == raw[i - 1]:
                if bound < 2:
                    i += 2
                    break
                push(hop)
                i += 3
                run = True
            if not run and i + 3 < n:
                step = raw[i + 3]
                if b == step:
                    if bound == 0:
                        i += 2
                        break
                    push(hop)
                    if bound < 2:
                        i += 3
                        break
                    push(step)
                    i += 4
                    run = True
            if run:
                rep = 0
                while rep < 251 and i < n and raw[i] == b:
                    rep += 1
                    i += 1
                push(rep)
                floor = i
                if i >= n:
                    break
                b = raw[i]
                continue
        i += 2
        b = hop
    return bytes(out), i


# ---- lib/bwt.rs:526-756, by its contract ------------------------------------------------------------------------------
def bwt(block):
    s = np.frombuffer(block, dtype=np.uint8).astype(np.int64)
    n = s.size
    if n == 1:
        return bytes(block), 0
    idx = np.arange(n, dtype=np.int64)
    rank = s.copy()
    k = 1
    while True:
        key = rank * (n + 257) + rank[(idx + k) % n]  # order by (first k bytes, next k bytes)
        order = np.argsort(key, kind="stable")
        ks = key[order]
        rank = np.empty(n, dtype=np.int64)
        rank[order] = np.concatenate(([0], np.cumsum(ks[1:] != ks[:-1])))
        if int(rank.max()) == n - 1:  # every rotation distinct
            sa = order
            break
        k *= 2
        if k >= n:  # what still ties is identical as a rotation: larger start index first
            sa = np.lexsort((-idx, rank))
            break
    last = s[(sa - 1) % n].astype(np.uint8).tobytes()
    return last, int(np.nonzero(sa == 0)[0][0])


# ---- lib/mtf.rs:14-121 ---------------------------------------------------------------------------------------------------
def mtf_and_rle(col, present):
    names = {}
    for byte in range(256):
        if present[byte]:
            names[byte] = len(names)
    num_names = len(names)
    eob = num_names + 1
    freqs = [0] * 258
    out = []
    recency = list(range(num_names))

    def zero_run(count):
        code = count + 1
        while True:
            bit = code & 1
            code >>= 1
            if code == 0:
                break
            out.append(bit)  # RUNA = 0, RUNB = 1
            freqs[bit] += 1

    zeros = 0
    for byte in col:
        name = names[byte]
        if name == recency[0]:
            zeros += 1
            continue
        if zeros:
            zero_run(zeros)
            zeros = 0
        r = recency.index(name)
        out.append(r + 1)
        freqs[r + 1] += 1
        del recency[r]
        recency.insert(0, name)
    if zeros:
        zero_run(zeros)
    out.append(eob)
    freqs[eob] = 1
    return out, num_names + 2, freqs


# ---- lib/huffman.rs:161-298: the reference's own heap, tie-breaks and all -------------------------------------------------
def build_table_from_freqs(num_syms, freqs, info=None):
    scaling = 1
    while True:
        heap = []  # 1-indexed through helper functions; entries (id, (weight, depth))

        def insert(sym, pr):
            heap.append((sym, pr))
            this = len(heap)
            init = this
            if init == 1:
                return
            while True:
                above = this >> 1
                asym, apr = heap[above - 1]
                if pr < apr:
                    heap[this - 1] = (asym, apr)
                    this = above
                    if this == 1:
                        break
                else:
                    break
            if this != init:
                heap[this - 1] = (sym, pr)

        def extract():
            sym, pr = heap.pop()
            if not heap:
                return sym, pr
            root = heap[0]
            heap[0] = (sym, pr)
            size = len(heap)
            this = 1
            while True:
                left = this << 1
                if left > size:
                    break
                right = left + 1
                if right <= size and heap[right - 1][1] < heap[left - 1][1]:
                    below = right
                else:
                    below = left
                bsym, bpr = heap[below - 1]
                if pr < bpr:
                    break
                heap[this - 1] = (bsym, bpr)
                this = below
            heap[this - 1] = (sym, pr)
            return root

        for s in range(num_syms):
            insert(s + 1, (freqs[s] // scaling + 1, 0))
        children = {}
        nnodes = num_syms + 1
        while True:
            a, pa = extract()
            c, pc = extract()
            if nnodes == 2 * num_syms - 1:
                children[0] = (a, c)
                break
            parent = nnodes
            nnodes += 1
            children[parent] = (a, c)
            insert(parent, (pa[0] + pc[0], max(pa[1], pc[1]) + 1))
        lengths = [0] * num_syms
        stack = [(0, 0)]
        maxlen = 0
        while stack:
            node, d = stack.pop()
            if node in children:
                stack.append((children[node][0], d + 1))
                stack.append((children[node][1], d + 1))
            else:
                lengths[node - 1] = d
                maxlen = max(maxlen, d)
        if maxlen <= 17:
            if info is not None and scaling > 1:
                info["rescaled"] = max(info.get("rescaled", 1), scaling)
            return lengths
        scaling <<= 1


# ---- lib/huffman.rs:313-575, literally (including the refinement that zeroes the tables) -----------------------------
def huffman_encode(bits, syms, num_syms, freqs, info=None):
    m = len(syms)
    if num_syms <= 199:
        num_tables = 2
    elif num_syms <= 599:
        num_tables = 3
    else:
        raise AssertionError("unreachable: num_syms <= 258")
    tables = []
    remaining = m
    left = 0
    for t in range(num_tables):
        target = remaining // (num_tables - t)
        acc = 0
        right = left
        while True:
            acc += freqs[right]
            if acc >= target or right + 1 == num_syms:
                break
            right += 1
        if right > left and t != 0 and t != num_tables - 1 and t % 2 == 1:
            acc -= freqs[right]
            right -= 1
        tables.append([15 if left <= s <= right else 0 for s in range(num_syms)])
        left = right + 1
        remaining -= acc
    table_freqs = [[0] * num_syms for _ in range(num_tables)]
    selectors = []
    for it in range(4):
        if it != 0:
            tables = [[0] * num_syms for _ in range(num_tables)]
        for lo in range(0, m, 50):
            seg = syms[lo:lo + 50]
            best, best_cost = 0, None
            for t, table in enumerate(tables):
                cost = sum(table[s] for s in seg)
                if best_cost is None or cost < best_cost:
                    best, best_cost = t, cost
            tf = table_freqs[best]
            for s in seg:
                tf[s] += 1
            if it == 3:
                selectors.append(best)
        tables = [build_table_from_freqs(num_syms, table_freqs[t], info) for t in range(num_tables)]
    if info is not None:
        info["tables"] = max(info.get("tables", 0), num_tables)
        info["maxlen"] = max(info.get("maxlen", 0), max(max(t) for t in tables))
    bits.put(num_tables, 3)
    bits.put(len(selectors), 15)
    order = list(range(num_tables))
    for sel in selectors:
        j = order.index(sel)
        bits.put((1 << (j + 1)) - 2 if j else 0, j + 1)
        del order[j]
        order.insert(0, sel)
    codings = []
    for table in tables:
        bits.put(table[0], 5)
        acc = table[0]
        for ln in table:
            while ln != acc:
                if ln > acc:
                    bits.put(2, 2)
                    acc += 1
                else:
                    bits.put(3, 2)
                    acc -= 1
            bits.put(0, 1)
        coding = [None] * num_syms
        word = 0
        for ln in range(min(table), max(table) + 1):
            for s in range(num_syms):
                if table[s] == ln:
                    coding[s] = (ln, word)
                    word += 1
            word <<= 1
        codings.append(coding)
    for i, s in enumerate(syms):
        ln, word = codings[selectors[i // 50]][s]
        bits.put(word, ln)


# ---- lib/lib.rs:18-132 ------------------------------------------------------------------------------------------------------
def encode(data, level, info=None):
    assert 1 <= level <= 9
    bits = Bits()
    bits.put_bytes(b"BZh" + bytes([48 + level]))
    stream_crc = 0
    raw = bytes(data)
    blocks = []
    while True:
        rle, consumed = rle_one(raw, level)
        if consumed == 0:
            break
        chk = checksum(raw[:consumed])
        stream_crc = chk ^ (((stream_crc << 1) | (stream_crc >> 31)) & 0xFFFFFFFF)
        col, ptr = bwt(rle)
        present = [False] * 256
        for byte in set(col):
            present[byte] = True
        bits.put_bytes(bytes.fromhex("314159265359"))
        bits.put(chk, 32)
        bits.put(0, 1)
        bits.put(ptr, 24)
        sector_map = 0
        sectors = []
        for a in range(16):
            sector = 0
            for b in range(16):
                sector = (sector << 1) | (1 if present[(a << 4) | b] else 0)
            sector_map = (sector_map << 1) | (1 if sector else 0)
            if sector:
                sectors.append(sector)
        bits.put(sector_map, 16)
        for sct in sectors:
            bits.put(sct, 16)
        syms, num_syms, freqs = mtf_and_rle(col, present)
        huffman_encode(bits, syms, num_syms, freqs, info)
        blocks.append((consumed, len(rle)))
        raw = raw[consumed:]
        if not raw:
            break
    bits.put_bytes(bytes.fromhex("177245385090"))
    bits.put(stream_crc, 32)
    if info is not None:
        info["blocks"] = blocks
    return bits.close()
