"""GPU (-m gpu): the HIP path, called through the C ABI (include/bzhip.h), against the oracle on the
same seeded inputs -- bit-exact for every stage seam and for whole .bz2 streams -- plus the
committed golden vectors and, at BASELINE.json's full size, size-independent properties
(libbz2 reproduces the input; block table tiles the input; sharded == monolithic)."""
import bz2
import io
import json
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def oracle_split(oracle, data, level):
    off, res, chunks = 0, [], []
    while off < len(data):
        r, crc, used = oracle.rle_one(data[off:], level)
        res.append((off, used, len(r), crc))
        chunks.append(r)
        off += used
    return res, chunks


# ---- stage seams -----------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", cases.MODES)
def test_bwt_seam(oracle, ctx9, mode):
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 17, 63, 64, 65, 257, 4095, 4096, 4097, 10000, 70000):
        d = cases.gen(n, mode, 2)
        g, o = ctx9.bwt(d), oracle.bwt(d)
        assert g[0] == o[0] and g[1] == o[1] and np.array_equal(g[2], o[2]), (mode, n)


def test_bwt_all_short_binary_strings(oracle, ctx9):
    """every string over {a, b} of length 1..9 (all periods, all wrap-arounds of the 8-byte initial
    sort key, every tie pattern of T6), in batches through bzh_bwt_batch"""
    blocks = [bytes(97 + ((v >> k) & 1) for k in range(n)) for n in range(1, 10) for v in range(1 << n)]
    step = 8  # the fixture context is created with max_batch = 8
    for k in range(0, len(blocks), step):
        part = blocks[k:k + step]
        for blk, r in zip(part, ctx9.bwt_batch(part)):
            o = oracle.bwt(blk)
            assert r[0] == o[0] and r[1] == o[1], blk


def test_bwt_golden_vectors(ctx9):
    """reference KAT (lib/bwt.rs:758-772) and the debug/bwt.py vectors, straight on the GPU"""
    v = json.load(open(os.path.join(GOLDEN, "ref_debug_vectors.json")))
    for c in v["bwt"]:
        b, ptr, _ = ctx9.bwt(c["input"].encode())
        assert b.decode() == c["bwt"] and ptr == c["ptr"], c["input"][:40]


def test_bwt_mid_size_groups(oracle, ctx9):
    """groups of 600..3000 equal 40-byte phrases that survive to depth 32: too large for the TAIL window
    (512), so these blocks stay on the radix path (ACTIVE rounds) while others in the batch are in TAIL
    mode at their own depth"""
    for n, seed, counts in ((600_000, 3, (700, 1500, 2800)), (899_000, 4, (513, 1025, 3072)), (120_000, 5, (600, 900))):
        d = cases.phrase_groups(n, seed, counts)
        g, o = ctx9.bwt(d), oracle.bwt(d)
        assert g[0] == o[0] and g[1] == o[1], (n, seed)
    blocks = [cases.phrase_groups(500_000, 7), cases.gen(400_000, "text", 1),
              cases.phrase_groups(880_000, 8, counts=(3000, 600, 1200)), cases.repeats(300_000, 2)]
    for r, blk in zip(ctx9.bwt_batch(blocks), blocks):
        o = oracle.bwt(blk)
        assert r[0] == o[0] and r[1] == o[1]


def test_bwt_full_block_config2(oracle, ctx9):
    """BASELINE.json configs[1]: one 899,999-byte block of uniform-random bytes, BWT only"""
    from banzai_amd import corpus
    d = corpus.xorshift_bytes(899_999).tobytes()
    g, o = ctx9.bwt(d), oracle.bwt(d)
    assert g[0] == o[0] and g[1] == o[1] and np.array_equal(g[2], o[2])


def test_bwt_batch_and_worst_cases(oracle, ctx9):
    near = (b"\0\0\0\0\xfb" * 180000)[:899_998]          # RLE1 image of a zero-filled input: ~18 rounds
    exact = (b"abcab" * 180000)[:899_995]                 # S = w^k: ties resolved by descending index (T6)
    text = cases.gen(600_000, "text", 1)
    blocks = [near, exact, text, b"z", text[:77777]]
    res = ctx9.bwt_batch(blocks)
    for r, b in zip(res, blocks):
        o = oracle.bwt(b)
        assert r[0] == o[0] and r[1] == o[1] and np.array_equal(r[2], o[2])


@pytest.mark.parametrize("mode", cases.MODES)
def test_mtf_seam(oracle, ctx9, mode):
    for n in (1, 2, 15, 16, 17, 1023, 1024, 1025, 2047, 2048, 2049, 4096, 4097, 20000, 300000):
        d = cases.gen(n, mode, 4)
        b, _, hb = oracle.bwt(d)
        gs, gf, gn = ctx9.mtf(b, hb)
        os_, of, on = oracle.mtf_and_rle(b, hb)
        assert gn == on and np.array_equal(gs, os_) and np.array_equal(gf, of), (mode, n)


def test_mtf_kat_on_gpu(ctx9):
    """reference lib/mtf.rs:139-158"""
    test = [153, 45, 45, 38, 135, 179, 26, 154, 165, 170, 170, 170, 170, 18, 109, 240, 174, 150, 87, 164, 30, 30,
            30, 30, 30, 30, 30, 148, 190, 10, 60, 13, 13, 13, 13, 13, 6, 81, 200, 13, 225, 32, 17, 43, 22, 179, 13,
            13, 17, 236, 236, 236, 236, 236, 236, 236, 121, 211, 2, 211, 185, 54, 16] + [5] * 22 + [50] + [5] * 22 + [40]
    expected = [27, 17, 0, 15, 25, 33, 15, 29, 31, 32, 0, 0, 17, 28, 40, 34, 33, 31, 34, 25, 1, 1, 34, 36, 23, 33, 25,
                1, 0, 25, 34, 37, 4, 39, 32, 31, 34, 33, 26, 7, 0, 5, 40, 1, 1, 38, 40, 34, 2, 40, 40, 38, 38, 0, 1,
                1, 0, 40, 2, 0, 1, 1, 0, 40, 41]
    hb = np.zeros(256, np.uint8)
    hb[list(set(test))] = 1
    syms, _, ns = ctx9.mtf(bytes(test), hb)
    assert list(syms) == expected and ns == len(set(test)) + 2


@pytest.mark.parametrize("mode", cases.MODES)
def test_huffman_seam(oracle, ctx9, mode):
    for n in (1, 3, 49, 50, 51, 1000, 4096, 4097, 70000, 300000):
        d = cases.gen(n, mode, 6)
        b, _, hb = oracle.bwt(d)
        s, f, ns = oracle.mtf_and_rle(b, hb)
        gbits, gn, glens = ctx9.huffman(s, ns, f)
        obits, on, olens = oracle.huffman_block(s, ns, f)
        assert gn == on and gbits == obits, (mode, n)
        assert np.array_equal(glens[:, :ns], olens[:, :ns])


def test_huffman_three_tables_and_rescale(oracle, ctx9):
    """>= 200 symbols -> 3 tables (T10); a geometric histogram forces the > 17-bit rescale loop (T13)"""
    rng = np.random.default_rng(3)
    d = np.minimum(rng.geometric(0.35, 400_000) - 1 + rng.integers(0, 2, 400_000) * 128, 255).astype(np.uint8).tobytes()
    b, _, hb = oracle.bwt(d)
    s, f, ns = oracle.mtf_and_rle(b, hb)
    gbits, gn, glens = ctx9.huffman(s, ns, f)
    obits, on, olens = oracle.huffman_block(s, ns, f)
    assert gn == on and gbits == obits and np.array_equal(glens[:, :ns], olens[:, :ns])
    # direct check of the code-length builder on Fibonacci-like weights (needs scaling > 1)
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    freqs = np.zeros(258, np.uint32)
    freqs[:40] = np.minimum(fib, 2 ** 31)
    assert oracle.build_table_from_freqs(40, freqs).max() <= 17


def test_huffman_scaling_attempts_of_both_halves(oracle, ctx9):
    """huff_build runs the scaling attempts 1, 2, 4, ... of a table side by side, the lower and the upper exponents in two
    workgroups a block, and huff_header takes the lower half's table if it has one: symbol streams whose counts double from
    symbol to symbol (codes up to 19 bits at scaling 1) need scaling 8 .. 64 -- tables decided by the lower half, by the
    upper half, with 2 tables and with 3 (>= 200 symbols)"""
    rng = np.random.default_rng(17)
    for top, extra in ((13, 0), (15, 0), (17, 0), (18, 0), (18, 100), (18, 230)):
        counts = np.array([1 << i for i in range(top + 1)] + [1] * extra, dtype=np.int64)
        ns = len(counts) + 1
        s = np.repeat(np.arange(len(counts), dtype=np.uint16), counts)
        rng.shuffle(s)
        s = np.concatenate([s, np.array([ns - 1], np.uint16)])
        f = np.zeros(258, np.uint32)
        f[:ns] = np.bincount(s, minlength=ns)
        gbits, gn, glens = ctx9.huffman(s, ns, f)
        obits, on, olens = oracle.huffman_block(s, ns, f)
        assert gn == on and gbits == obits and np.array_equal(glens[:, :ns], olens[:, :ns]), (top, extra)
        assert glens[0, :ns].max() <= 17


@pytest.mark.parametrize("level,ctxname", [(1, "ctx1"), (9, "ctx9")])
def test_rle1_split_and_crc_seam(oracle, request, level, ctxname):
    ctx = request.getfixturevalue(ctxname)
    sizes = cases.SIZES_L1 if level == 1 else cases.SIZES_L9
    for mode in cases.MODES:
        for n in sizes:
            d = cases.gen(n, mode, 8)
            infos, chunks = ctx.rle1_split(d)
            oi, oc = oracle_split(oracle, d, level)
            assert infos == oi and chunks == oc, (level, mode, n)
            if n:
                assert ctx.crc32(d) == oracle.crc32(d)


def test_rle1_boundary_residues(oracle, ctx1):
    for d in cases.boundary_cases()[::3]:
        infos, chunks = ctx1.rle1_split(d)
        oi, oc = oracle_split(oracle, d, 1)
        assert infos == oi and chunks == oc


def test_rle1_golden_vectors(ctx9):
    """debug/rle1.py vectors (unbounded RLE1; inputs < M so the bound never bites)"""
    v = json.load(open(os.path.join(GOLDEN, "ref_debug_vectors.json")))
    for c in v["rle1"]:
        d = bytes.fromhex(c["input_hex"])
        infos, chunks = ctx9.rle1_split(d)
        assert len(infos) == 1 and chunks[0].hex() == c["rle1_hex"] and infos[0][1] == len(d)


def test_crc_check_value(ctx9):
    assert ctx9.crc32(b"123456789") == 0xFC891918


# ---- whole streams ---------------------------------------------------------------------------------------
def test_golden_streams(ctx1, ctx9):
    g = json.load(open(os.path.join(GOLDEN, "streams.json")))
    for c in g["streams"]:
        data = bytes.fromhex(c["input_hex"]) if "input_hex" in c else bytes([c["fill"]]) * c["count"]
        ctx = ctx1 if c["level"] == 1 else ctx9
        assert ctx.encode(data).hex() == c["stream_hex"], c["name"]


def test_model_streams_on_gpu(native):
    """the HIP path against the digests of the independent Python model (tests/golden/model_streams.json)"""
    import hashlib
    from tests.golden import stream_cases
    v = json.load(open(os.path.join(GOLDEN, "model_streams.json")))["cases"]
    ctxs = {}
    try:
        for name, c in v.items():
            level, data = stream_cases.CASES[name]()
            assert hashlib.sha256(data).hexdigest() == c["input_sha256"], name
            ctx = ctxs.setdefault(level, native.Context(0, level, 8))
            got = ctx.encode(data)
            assert len(got) == c["stream_len"] and hashlib.sha256(got).hexdigest() == c["stream_sha256"], name
            infos, _ = ctx.rle1_split(data, want_bytes=False)
            assert [(i[1], i[2]) for i in infos] == [tuple(x) for x in c["blocks_consumed_rle"]], name
    finally:
        for ctx in ctxs.values():
            ctx.close()


def test_rle1_large_reference_vectors_on_gpu(ctx9):
    import hashlib
    from tests.golden import stream_cases
    v = json.load(open(os.path.join(GOLDEN, "ref_rle1_large.json")))["cases"]
    for name, c in v.items():
        d = stream_cases.RLE1_LARGE[name]()
        assert hashlib.sha256(d).hexdigest() == c["input_sha256"], name
        infos, chunks = ctx9.rle1_split(d)
        assert len(infos) == 1 and infos[0][1] == len(d)
        assert len(chunks[0]) == c["rle1_len"] and hashlib.sha256(chunks[0]).hexdigest() == c["rle1_sha256"], name


@pytest.mark.parametrize("mode", cases.MODES)
def test_stream_bit_exact_level1(oracle, ctx1, mode):
    for n in cases.SIZES_L1:
        d = cases.gen(n, mode, 12)
        g = ctx1.encode(d)
        assert g == oracle.encode(d, 1), (mode, n)
        assert bz2.decompress(g) == d and oracle.decode(g) == d  # libbz2 and the in-repo decoder


@pytest.mark.parametrize("mode", ["random", "text", "longruns", "shortruns", "same"])
def test_stream_bit_exact_level9(oracle, ctx9, mode):
    for n in cases.SIZES_L9:
        d = cases.gen(n, mode, 13)
        g = ctx9.encode(d)
        assert g == oracle.encode(d, 9), (mode, n)


def test_fuzz_streams(oracle, ctx1):
    """the reference's fuzz target (fuzz/fuzz_targets/round_trip.rs: level 1, arbitrary bytes), here
    with the oracle as the bit-exact judge: random lengths, alphabets and run structures"""
    import random
    rng = random.Random(20260)
    for it in range(150):
        n = rng.choice([rng.randrange(0, 300), rng.randrange(300, 5000), rng.randrange(5000, 130000)])
        kind = rng.randrange(5)
        if kind == 0:
            d = bytes(rng.randrange(256) for _ in range(n))
        elif kind == 1:
            d = bytes(rng.randrange(rng.choice([1, 2, 3, 5])) for _ in range(n))
        elif kind == 2:
            d = bytearray()
            while len(d) < n:
                d += bytes([rng.randrange(256)]) * rng.choice([1, 1, 2, 3, 4, 5, 6, 250, 255, 256, 260, 1000])
            d = bytes(d[:n])
        elif kind == 3:
            w = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
            d = (w * (n // len(w) + 1))[:n]
        else:
            d = cases.repeats(n, it) if n > 200 else bytes(n)
        g = ctx1.encode(d)
        assert g == oracle.encode(d, 1), (it, kind, n)


def test_tail_rounds_long_repeats(oracle, ctx9):
    """verbatim repeats of up to n/8 bytes: thousands of two-member groups resolved in TAIL rounds,
    block boundaries in the middle of repeats; plus exact and near periodicity at full block size"""
    for n, seed in ((400_000, 1), (899_999, 2), (2_000_000, 3)):
        d = cases.repeats(n, seed, copies=10)
        assert ctx9.encode(d) == oracle.encode(d, 9), (n, seed)
    per = cases.gen(899_999, "periodic", 4)
    b, ptr, hb = ctx9.bwt(per)
    ob = oracle.bwt(per)
    assert (b, ptr) == (ob[0], ob[1])


def test_other_levels(oracle, native):
    d = cases.gen(1_234_567, "text", 2) + cases.gen(300_000, "longruns", 2)
    for level in (2, 5, 8):
        with native.Context(0, level, 4) as ctx:
            assert ctx.encode(d) == oracle.encode(d, level), level


def test_public_api_matches_reference_surface(oracle, tmp_path):
    """encode(reader, writer, level) -> consumed; encode_file(in, out) at level 9 (lib/lib.rs:84-153)"""
    import banzai_amd
    d = cases.gen(250_000, "text", 3)
    out = io.BytesIO()
    assert banzai_amd.encode(io.BytesIO(d), out, 1) == len(d)
    assert out.getvalue() == oracle.encode(d, 1)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bz2"
    src.write_bytes(d)
    assert banzai_amd.encode_file(str(src), str(dst)) == len(d)
    assert dst.read_bytes() == oracle.encode(d, 9)


def test_pathological_config5(oracle, ctx9):
    """BASELINE.json configs[4] (scaled to 8 MB): long single-byte runs + periodic repeats"""
    from banzai_amd import corpus
    d = corpus.pathological(8_000_000).tobytes()
    g = ctx9.encode(d)
    assert g == oracle.encode(d, 9)
    assert bz2.decompress(g) == d and oracle.decode(g) == d


def test_fuzz_lite(oracle, native):
    """seeded stand-in for the reference's fuzz targets (fuzz/fuzz_targets/encode.rs, round_trip.rs):
    random mixtures at random levels must equal the oracle's stream and decode back to the input"""
    import random
    rng = random.Random(20240611)
    ctxs = {}
    try:
        for k in range(70):
            level = rng.choice([1, 1, 1, 2, 3, 9])
            d = cases.mixture(rng, rng.choice([0, 10, 1000, 99_999, 100_001, 350_000]))
            if level not in ctxs:
                ctxs[level] = native.Context(0, level, 8)
            g = ctxs[level].encode(d)
            assert g == oracle.encode(d, level), (k, level, len(d))
            assert oracle.decode(g) == d, (k, level, len(d))
    finally:
        for c in ctxs.values():
            c.close()


# ---- device-resident and sharded paths ------------------------------------------------------------------
def test_device_path_and_sharded_equals_monolithic(oracle, native):
    """bzh_encode_device == oracle, and plan + 3 x encode_range + assemble (the N>1 protocol, here
    on one GPU) gives the identical stream."""
    import torch
    from banzai_amd import corpus, sharded
    n = 12_345_678
    data = corpus.enwik_synthetic(n, seed=3)
    want = oracle.encode(data.tobytes(), 9)
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    d_in[:n] = torch.from_numpy(data).to(dev)
    cap = (n // 2 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    with native.Context(0, 9, 8) as ctx:  # max_batch 8 < 14 blocks: exercises batching
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        assert d_out[:ln].cpu().numpy().tobytes() == want
        blocks = ctx.plan_device(d_in.data_ptr(), n)
        assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == n
        assert all(blocks[k][0] + blocks[k][1] == blocks[k + 1][0] for k in range(len(blocks) - 1))
        world, segs, keep = 3, [], []
        for r in range(world):
            b0, b1 = sharded.block_range(len(blocks), r, world)
            buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
            nbits = ctx.encode_range_device(b0, b1, buf.data_ptr(), cap)
            keep.append(buf)
            segs.append((buf.data_ptr(), nbits))
        d_out.zero_()
        ln2 = ctx.assemble_device(segs, [b[3] for b in blocks], d_out.data_ptr(), cap)
        assert d_out[:ln2].cpu().numpy().tobytes() == want


def test_plan_without_crcs_then_per_range(oracle, native):
    """the sharded path's plan: cuts without CRCs; encode_range computes those of its range, the rest
    stay 0 until asked for; assembling with the collected CRCs gives the stream; DeviceEngine end to end"""
    import torch
    from banzai_amd import corpus, sharded
    n = 6_000_123
    data = corpus.enwik_synthetic(n, seed=11)
    want = oracle.encode(data.tobytes(), 9)
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    d_in[:n] = torch.from_numpy(data).to(dev)
    cap = (n // 2 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    with native.Context(0, 9, 8) as ctx:
        full = ctx.plan_device(d_in.data_ptr(), n)
        blocks = ctx.plan_device(d_in.data_ptr(), n, crc=False)
        assert [b[:3] for b in blocks] == [b[:3] for b in full] and all(b[3] == 0 for b in blocks)
        nb = len(blocks)
        mid = nb // 2
        buf0 = torch.zeros(cap, dtype=torch.uint8, device=dev)
        bits0 = ctx.encode_range_device(0, mid, buf0.data_ptr(), cap)
        got = ctx.plan_blocks()
        assert [b[3] for b in got[:mid]] == [b[3] for b in full[:mid]] and all(b[3] == 0 for b in got[mid:])
        assert ctx.plan_crc_range(mid, nb) == [b[3] for b in full[mid:]]
        buf1 = torch.zeros(cap, dtype=torch.uint8, device=dev)
        bits1 = ctx.encode_range_device(mid, nb, buf1.data_ptr(), cap)
        crcs = [b[3] for b in ctx.plan_blocks()]
        ln = ctx.assemble_device([(buf0.data_ptr(), bits0), (buf1.data_ptr(), bits1)], crcs, d_out.data_ptr(), cap)
        assert d_out[:ln].cpu().numpy().tobytes() == want
        d_out.zero_()
        eng = sharded.DeviceEngine(ctx, d_in, n, d_out, cap)
        ln = sharded.encode_sharded(eng)
        assert d_out[:ln].cpu().numpy().tobytes() == want


def test_offset_ownership_with_prefix_plans_on_device(oracle, native):
    """the N > 1 protocol rank by rank on one GPU: every rank plans only a prefix of the input (small
    look-ahead, so cuts come back open and the margin has to grow inside a 3 MB run), encodes the blocks
    that start in its byte range, and the assembled stream is the single-GPU stream"""
    import torch
    from banzai_amd import corpus, sharded
    data = np.concatenate([corpus.enwik_synthetic(5_000_000, seed=13), np.zeros(3_000_000, np.uint8),
                           corpus.enwik_synthetic(2_500_000, seed=14)])
    n = data.size
    want = oracle.encode(data.tobytes(), 9)
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    d_in[:n] = torch.from_numpy(data).to(dev)
    cap = (n // 2 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    with native.Context(0, 9, 8) as ctx:
        for world in (2, 3, 7):
            segs, keep, crcs, start = [], [], [], 0
            for r in range(world):  # rank r sees its own range plus a look-ahead, and the start rank r-1 hands over
                # (the block that holds the 3 MB zero run ends up to 3.9 MB behind the range it starts in)
                lo, hi = sharded.resident_range(n, r, world, lookahead=5_000_000)
                d_r = torch.zeros(hi - lo + 16, dtype=torch.uint8, device=dev)
                d_r[:hi - lo] = d_in[lo:hi]
                eng = sharded.DeviceEngine(ctx, d_r, n, d_out, cap, resident=hi - lo, lo=lo)
                eng.tables()
                blocks, b0, b1, start = sharded.own_blocks(eng, r, world, start)
                part, nbits = eng.encode_range(b0, b1)
                keep.append(part.clone())
                segs.append((keep[-1], nbits))
                crcs += eng.crcs(b0, b1)
            assert start == n
            d_out.zero_()
            ln = eng.assemble(segs, crcs)
            assert d_out[:ln].cpu().numpy().tobytes() == want, world


def test_two_lanes_same_bytes(oracle, native):
    """bzh_set_lanes(2): half-batches prepared concurrently on two internal streams, packed in order"""
    d = cases.gen(3_000_001, "text", 9) + cases.repeats(1_500_000, 9) + cases.gen(400_000, "longruns", 9)
    with native.Context(0, 1, 8) as ctx:  # level 1: ~49 blocks -> 13 jobs over two lanes
        ctx.set_lanes(2)
        assert ctx.encode(d) == oracle.encode(d, 1)
        ctx.set_lanes(1)
        assert ctx.encode(d) == oracle.encode(d, 1)


def test_output_capacity_error(native):
    import torch
    dev = torch.device("cuda", 0)
    data = np.random.default_rng(0).integers(0, 256, 500_000, dtype=np.uint8)
    d_in = torch.zeros(data.size + 16, dtype=torch.uint8, device=dev)
    d_in[:data.size] = torch.from_numpy(data).to(dev)
    d_out = torch.zeros(1024, dtype=torch.uint8, device=dev)
    with native.Context(0, 9, 4) as ctx:
        with pytest.raises(native.BzhError) as e:
            ctx.encode_device(d_in.data_ptr(), data.size, d_out.data_ptr(), 1024)
        assert e.value.status == -4


def test_output_capacity_boundary_of_a_framed_stream(oracle, native):
    """a one-batch stream gets its header and footer on the device (frame_stream), the capacity check included (pack_gate counts
    the footer's 80 bits): with exactly the room include/bzhip.h asks for -- the stream rounded up to 4 bytes, plus 4 -- the call
    succeeds and nothing is written behind it; with 8 bytes less it fails with BZH_E_CAP and the context stays usable"""
    import torch
    dev = torch.device("cuda", 0)
    for data in (cases.repeats(700_000, 8), cases.gen(1_000_000, "random", 8)[:300_001], b"q"):
        want = oracle.encode(data, 9)
        n = len(data)
        d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
        d_in[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        need = (len(want) + 3) // 4 * 4 + 4
        d_out = torch.full((need + 64,), 0xA5, dtype=torch.uint8, device=dev)
        with native.Context(0, 9, 4) as ctx:
            ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), need)
            got = d_out.cpu().numpy()
            assert got[:ln].tobytes() == want and bool((got[need:] == 0xA5).all())
            if need > 16:
                with pytest.raises(native.BzhError) as e:
                    ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), need - 8)
                assert e.value.status == -4
            assert ctx.encode(data) == want


def test_full_size_properties_config3(native, oracle):
    """BASELINE.json configs[2] at full size (100,000,000 bytes): properties that do not need the
    encoding oracle -- libbz2 and the in-repo decoder reproduce the input (pins every CRC, table and block cut), the block table
    tiles the input, RLE1 lengths respect the level, and two runs give identical bytes."""
    import torch
    from banzai_amd import corpus
    n = 100_000_000
    data, _ = corpus.workload(n)
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    d_in[:n] = torch.from_numpy(data).to(dev)
    cap = (n // 2 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    with native.Context(0, 9, 128) as ctx:
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        first = d_out[:ln].cpu().numpy().tobytes()
        blocks = ctx.plan_device(d_in.data_ptr(), n)
        d_out.zero_()
        ln2 = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        assert d_out[:ln2].cpu().numpy().tobytes() == first
    assert sum(b[1] for b in blocks) == n and all(0 < b[2] <= 899_999 for b in blocks)
    assert first[:4] == b"BZh9"
    assert bz2.decompress(first) == data.tobytes() and oracle.decode(first) == data.tobytes()


def test_full_size_config4_one_gigabyte_and_world8_simulation(native, oracle):
    """BASELINE.json configs[3] (enwik9-sized: 1,000,000,000 bytes, ~1,112 blocks) on one GPU: the block table
    tiles the input, two runs give identical bytes, the in-repo decoder reproduces the input (pins every
    CRC, table and block cut), and the 8-rank protocol run rank by rank -- each rank restricted to the input
    prefix bench.py would make resident on it -- assembles to the very same stream."""
    import torch
    from banzai_amd import corpus, sharded
    seg, world = 100_000_000, 8
    n = 1_000_000_000
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    host = np.empty(n, dtype=np.uint8)
    for k in range(n // seg):  # the segments bench.py's ranks would generate
        host[k * seg:(k + 1) * seg] = corpus.workload(seg, segment=k)[0]
        d_in[k * seg:(k + 1) * seg] = torch.from_numpy(host[k * seg:(k + 1) * seg]).to(dev)
    cap = (n // 3 + n // 8 + (1 << 20)) & ~3
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    with native.Context(0, 9, 128) as ctx:
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        mono = d_out[:ln].clone()
        blocks = ctx.plan_device(d_in.data_ptr(), n)
        d_out.zero_()
        ln2 = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        assert ln2 == ln and torch.equal(d_out[:ln], mono)
        assert sum(b[1] for b in blocks) == n and all(0 < b[2] <= 899_999 for b in blocks)
        assert all(blocks[k][0] + blocks[k][1] == blocks[k + 1][0] for k in range(len(blocks) - 1))
        assert 1100 <= len(blocks) <= 1130
        assert ctx.bwt_roundtrip_device(0, len(blocks)) == 0  # every block: BWT -> inverse BWT on the device
        # world = 8, rank by rank: a rank holds its own range + 64 MiB and gets its first block's start from the rank before
        slab = sharded.worst_case_slab(n, world, 9)
        segs, keep, crcs, nblk, start = [], [], [], 0, 0
        for r in range(world):
            lo, hi = sharded.resident_range(n, r, world)
            assert hi - lo <= (sharded.offsets(n, world)[r + 1] - lo) + (64 << 20)
            d_r = torch.zeros(hi - lo + 16, dtype=torch.uint8, device=dev)
            d_r[:hi - lo] = d_in[lo:hi]
            eng = sharded.DeviceEngine(ctx, d_r, n, d_out, slab, resident=hi - lo, lo=lo)
            eng.tables()
            _, b0, b1, start = sharded.own_blocks(eng, r, world, start)
            part, nbits = eng.encode_range(b0, b1)
            keep.append(part[:(nbits + 31) // 32 * 4 + 4].clone())
            segs.append((keep[-1], nbits))
            crcs += eng.crcs(b0, b1)
            nblk += b1 - b0
            del eng, d_r
        assert start == n and nblk == len(blocks) and crcs == [b[3] for b in blocks]
        d_out.zero_()
        eng = sharded.DeviceEngine(ctx, d_in, n, d_out, 16)
        ln3 = eng.assemble(segs, crcs)
        assert ln3 == ln and torch.equal(d_out[:ln], mono)
    stream = mono.cpu().numpy().tobytes()
    assert stream[:4] == b"BZh9"
    assert oracle.decode(stream, cap=n + 64) == host.tobytes()


def test_gpu_inverse_bwt_round_trip(oracle, ctx9, ctx1):
    """f3: the GPU inverse transform undoes the GPU BWT (and the oracle's): periodic blocks (several cycles in the
    LF mapping), runs, text, random, n = 1, 2; and equals what the reference's KAT says"""
    import random
    rng = random.Random(12)
    blocks = [cases.gen(n, mode, 3) for mode in ("text", "longruns", "shortruns", "random", "same")
              for n in (1, 2, 3, 255, 4096, 70_001)]
    blocks += [b"ab" * 5000, b"abc" * 3333 + b"ab", bytes(rng.randrange(256) for _ in range(1024)) * 80, b"a" * 99_999,
               cases.gen(899_999, "text", 4)]
    fwd = ctx9.bwt_batch(blocks)
    back = ctx9.unbwt_batch([(bw, p) for bw, p, _ in fwd])
    assert back == [bytes(b) for b in blocks]
    for blk in blocks[:12]:  # the oracle's transform is undone just the same
        bw, p, _ = oracle.bwt(blk)
        assert ctx1.unbwt_batch([(bw, p)]) == [bytes(blk)]
    kat_in = b"If Peter Piper picked a peck of pickled peppers, where's the peck of pickled peppers Peter Piper picked?????"
    bw, p, _ = ctx9.bwt(kat_in)
    assert ctx9.unbwt_batch([(bw, p)]) == [kat_in]


def test_bwt_round_trip_on_device_config3(native):
    """BASELINE.json configs[2] at full size without leaving the GPU: every block's RLE1 bytes -> BWT -> inverse
    BWT compare equal on the device (the reference's round_trip property for the dominant stage)"""
    import torch
    from banzai_amd import corpus
    n = 100_000_000
    data, _ = corpus.workload(n)
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
    d_in[:n] = torch.from_numpy(data).to(dev)
    with native.Context(0, 9, 128) as ctx:
        blocks = ctx.plan_device(d_in.data_ptr(), n)
        assert ctx.bwt_roundtrip_device(0, len(blocks)) == 0


def test_fixed_huffman_mode(oracle, native):
    """f4 (opt-in, bzh_set_mode): 2..6 tables from the symbol count, real refinement iterations, per-segment
    selectors.  Not the reference's bits -- but every stream must decode (libbz2 and the strict in-repo decoder)
    to the input, must not be larger than the default mode's (beyond tiny streams), and switching back restores the
    reference's bytes."""
    from banzai_amd import corpus
    inputs = [b"", b"x", b"abab", cases.gen(60_000, "text", 1), cases.gen(250_000, "random", 2),
              cases.gen(300_000, "longruns", 3), cases.gen(180_000, "shortruns", 4), cases.gen(99_999, "same", 5),
              corpus.enwik_synthetic(2_700_000, seed=9).tobytes(), bytes([1, 2, 3, 2, 1] * 700)]
    gains = []
    for level in (1, 9):
        with native.Context(0, level, 8) as ctx:
            for d in inputs:
                ref = ctx.encode(d)
                assert ref == oracle.encode(d, level)
                ctx.set_mode(True)
                fx = ctx.encode(d)
                ctx.set_mode(False)
                assert bz2.decompress(fx) == d and oracle.decode(fx, cap=len(d) + 64) == d
                # (six table headers and real selectors cost a few hundred bytes per block: a block of a few hundred
                # symbols, or incompressible bytes, can lose that much -- as with libbz2)
                assert len(fx) <= len(ref) + 16 + len(ref) // 200, (level, len(d), len(fx), len(ref))
                assert ctx.encode(d) == ref  # the default path is untouched by the detour
                if len(d) > 100_000:
                    gains.append(len(fx) / len(ref))
    assert min(gains) < 0.97  # several tables do pay on text-like data


def test_two_contexts_concurrently(oracle, native):
    """include/bzhip.h: a context is single-threaded, distinct contexts may run concurrently -- two host threads,
    one context each, same GPU, different inputs and levels, several encodes in a row"""
    import threading
    from banzai_amd import corpus
    jobs = [(9, corpus.enwik_synthetic(5_000_000, seed=51).tobytes() + cases.gen(300_000, "longruns", 5)),
            (1, cases.gen(2_000_000, "text", 6) + cases.repeats(800_000, 6) + cases.gen(300_000, "shortruns", 6))]
    want = [oracle.encode(d, lv) for lv, d in jobs]
    got, errs = [None, None], []

    def work(k):
        try:
            lv, d = jobs[k]
            with native.Context(0, lv, 8) as ctx:
                for _ in range(4):
                    got[k] = ctx.encode(d)
                    if got[k] != want[k]:
                        errs.append((k, "mismatch"))
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and got == want


def test_two_lanes_level9_several_batches(oracle, native):
    """bzh_set_lanes(2) at level 9 with more than two batches per lane (max_batch 4 -> lanes of 2 blocks)"""
    from banzai_amd import corpus
    d = corpus.enwik_synthetic(9_300_000, seed=41).tobytes() + cases.gen(400_000, "longruns", 2)
    want = oracle.encode(d, 9)
    with native.Context(0, 9, 4) as ctx:
        ctx.set_lanes(2)
        assert ctx.encode(d) == want
        assert ctx.encode(d) == want


def test_empty_plan_after_a_non_empty_one(native):
    """plan state is reset for n = 0 as well (blocks, open flags)"""
    import torch
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(300_016, dtype=torch.uint8, device=dev)
    d_in[:300_000] = torch.from_numpy(np.frombuffer(cases.gen(300_000, "text", 1), dtype=np.uint8).copy()).to(dev)
    with native.Context(0, 1, 4) as ctx:
        assert len(ctx.plan_device(d_in.data_ptr(), 300_000)) >= 3
        assert len(ctx.plan_open()) >= 3
        assert ctx.plan_device(d_in.data_ptr(), 0) == []
        assert ctx.plan_open() == []


# ---- streaming --------------------------------------------------------------------------
def _stream(ctx, data, cuts, chunk_bytes):
    ctx.stream_begin(chunk_bytes)
    out, pos = [], 0
    for c in cuts:
        out.append(ctx.stream_feed(data[pos:pos + c]))
        pos += c
    out.append(ctx.stream_feed(data[pos:], eof=True))
    assert ctx.stream_consumed() == len(data)
    return b"".join(out)


def test_streaming_equals_one_shot(oracle, ctx1):
    """any chunking of the reader gives the same stream (the reference's BufRead contract), including
    feeds that end inside runs, GPU passes whose last blocks are not final yet, and empty input"""
    import random
    rng = random.Random(5)
    assert _stream(ctx1, b"", [], 1 << 20) == oracle.encode(b"", 1)
    for mode in ("text", "longruns", "shortruns", "same", "random"):
        d = cases.gen(700_001, mode, 21)
        want = oracle.encode(d, 1)
        for chunk_bytes in (150_000, 1 << 20):
            cuts, left = [], len(d)
            while left > 0 and len(cuts) < 40:
                c = min(left, rng.choice([1, 7, 4096, 99_999, 100_000, 250_000, 333_333]))
                cuts.append(c)
                left -= c
            assert _stream(ctx1, d, cuts, chunk_bytes) == want, (mode, chunk_bytes)
    # a zero run far longer than a block: blocks inside it must be released before the run ends
    d = b"\0" * 30_000_000 + b"tail"
    ctx1.stream_begin(4 << 20)
    pieces = [ctx1.stream_feed(d[k:k + (3 << 20)]) for k in range(0, len(d), 3 << 20)]
    assert sum(len(p) for p in pieces) > 0  # output appeared before eof although the run was still open
    pieces.append(ctx1.stream_feed(b"", eof=True))
    assert b"".join(pieces) == oracle.encode(d, 1)


def test_streaming_small_buffer_is_refused_before_anything_is_consumed(native, oracle, ctx1):
    """BZH_E_CAP comes back before the feed is taken: the same feed can be repeated with a buffer of
    bzh_stream_bound() bytes, and the stream is the one-shot stream (passes overlap with feeding, so
    the bound depends on what is pending and in flight)"""
    import ctypes
    lib = native.lib()
    d = cases.gen(900_000, "text", 8)
    ctx1.stream_begin(200_000)
    out = bytearray()
    got = ctypes.c_size_t(0)
    small = np.zeros(16, np.uint8)
    for k in range(0, len(d) + 1, 300_000):
        piece = np.frombuffer(d[k:k + 300_000], dtype=np.uint8) if k < len(d) else np.zeros(1, np.uint8)
        n = min(300_000, len(d) - k) if k < len(d) else 0
        eof = 1 if k + 300_000 >= len(d) else 0
        assert lib.bzh_stream_feed(ctx1.handle, native.ptr(piece), n, eof, native.ptr(small), small.size,
                                   ctypes.byref(got)) == -4  # BZH_E_CAP, nothing consumed
        buf = np.zeros(int(lib.bzh_stream_bound(ctx1.handle, n)), np.uint8)
        ctx1.check(lib.bzh_stream_feed(ctx1.handle, native.ptr(piece), n, eof, native.ptr(buf), buf.size, ctypes.byref(got)))
        out += buf[:got.value].tobytes()
        if eof:
            break
    assert bytes(out) == oracle.encode(d, 1)


def test_public_api_level1_default_batch(oracle):
    """banzai_amd.encode at level 1: the default batch is 1024 blocks there (same bytes per batch as
    128 level-9 blocks), which uses the copied-gates path of the round setup"""
    import banzai_amd
    d = cases.gen(4_000_001, "text", 17) + cases.repeats(1_000_000, 17) + cases.gen(700_000, "longruns", 17)
    w = io.BytesIO()
    assert banzai_amd.encode(io.BytesIO(d), w, 1) == len(d)
    assert w.getvalue() == oracle.encode(d, 1)


def test_streaming_public_api_chunked_reader(oracle):
    import banzai_amd

    class Dribble(io.RawIOBase):  # a reader that hands out odd-sized pieces
        def __init__(self, data):
            self.d, self.p, self.k = data, 0, 0

        def read(self, n=-1):
            self.k += 1
            take = min(n if n >= 0 else len(self.d), 1 + (self.k * 7919) % 300_000)
            out = self.d[self.p:self.p + take]
            self.p += len(out)
            return out

    d = cases.gen(2_000_003, "text", 4) + cases.gen(500_000, "longruns", 4)
    w = io.BytesIO()
    assert banzai_amd.encode(Dribble(d), w, 9) == len(d)
    assert w.getvalue() == oracle.encode(d, 9)


def test_public_api_retaining_writer(oracle):
    """A writer that KEEPS what write() is handed (a list sink, a queue, a transport) must not see its pieces change
    under it: encode() hands the context's reusable output buffer only to sinks known to copy (BytesIO, real files)."""
    import banzai_amd

    class Keeper:  # several feeds with output each: level 1, 16 MiB reads over 108 MB
        def __init__(self):
            self.parts = []

        def write(self, b):
            self.parts.append(b)
            return len(b)

    d = cases.gen(36_000_000, "text", 23) * 3  # (a pass starts every 32 MiB of input: three of them hand out bytes before eof)
    k = Keeper()
    assert banzai_amd.encode(io.BytesIO(d), k, 1) == len(d)
    assert len(k.parts) >= 2 and all(isinstance(p, bytes) for p in k.parts)
    assert b"".join(k.parts) == oracle.encode(d, 1)


def test_near_periodic_blocks_vs_oracle(oracle, ctx9):
    """BASELINE config 5, the near-periodic part: a word repeated and cut off inside a repetition (periods 2, 3, 1024,
    4099; lengths 899,999 / 899,998 / 450,000), plus words whose own structure mixes phases in the early groups, words
    with a damaged repetition (no period: the probe must refuse) and exactly periodic blocks: last column and origin
    pointer equal the oracle's"""
    rng = np.random.default_rng(99)
    blocks = []
    for p in (2, 3, 1024, 4099):
        w = rng.integers(0, 256, p, dtype=np.uint8)
        while p > 1 and len(set(w.tolist())) < 2:
            w = rng.integers(0, 256, p, dtype=np.uint8)
        for n in (899_999, 899_998, 450_000):
            blocks.append(np.tile(w, n // p + 1)[:n].tobytes())
    # "ab" x 500 + "cd": the first groups hold many phases of the word
    w = np.frombuffer(b"ab" * 500 + b"cd", dtype=np.uint8)
    blocks.append(np.tile(w, 900)[:899_999].tobytes())
    # a low-entropy word: 8-byte prefixes recur inside it
    w = rng.integers(0, 2, 777, dtype=np.uint8) + 97
    blocks.append(np.tile(w, 1200)[:899_999].tobytes())
    blocks.append(np.tile(w, 1200)[:777 * 1000].tobytes())  # exactly periodic
    # one damaged byte in the middle / near the end: not periodic any more
    w = rng.integers(0, 256, 1024, dtype=np.uint8)
    d = np.tile(w, 880)[:899_999].copy()
    d[450_000] ^= 1
    blocks.append(d.tobytes())
    d = np.tile(w, 880)[:899_999].copy()
    d[899_990] ^= 1
    blocks.append(d.tobytes())
    # what RLE1 leaves of one long run: "aaaa" + count, period 5
    blocks.append((b"\0\0\0\0\xfb" * 180_000)[:899_999])
    got = ctx9.bwt_batch(blocks)
    for k, blk in enumerate(blocks):
        bw, ptr, _ = oracle.bwt(blk)
        assert got[k][0] == bw and got[k][1] == ptr, (k, len(blk))


def test_lookback_give_up_is_an_error_not_a_hang(oracle, native):
    """a tile that never publishes its look-back status (injected): the tiles behind it give up after a bounded wait,
    the call returns an error status, and the context encodes correctly afterwards"""
    import time
    d = cases.repeats(400_000, 3)  # (not a repeated sentence: a near-periodic block is sorted as eight of its periods, one tile)
    with native.Context(0, 9, 8) as ctx:
        ctx.debug_fault(1)
        t0 = time.perf_counter()
        with pytest.raises(native.BzhError) as ei:
            ctx.encode(d)
        assert time.perf_counter() - t0 < 60
        assert "look-back gave up" in str(ei.value)
        assert ctx.encode(d) == oracle.encode(d, 9)


def test_lookback_give_up_of_a_shared_gpu_is_recovered(oracle, native):
    """what several processes computing on one GPU do to a small batch (a look-back gives up: its predecessor is queued on
    an XCD whose slots another process holds), injected: the suffix sort runs again with every block on one XCD, the call
    succeeds with the oracle's bytes, and the context keeps that mapping"""
    d = cases.repeats(400_000, 5) + cases.repeats(2_000_000, 6)  # (see above: blocks that take the general sort)
    with native.Context(0, 9, 8) as ctx:
        ctx.debug_fault(2)
        assert ctx.encode(d) == oracle.encode(d, 9)
        assert ctx.encode(d[:700_000]) == oracle.encode(d[:700_000], 9)


def test_stream_beyond_four_gib(native):
    """more than 2^32 input bytes through bzh_stream_feed (the reference's encode has no length limit, lib/lib.rs:84-132;
    one plan here has 32-bit positions, so the stream is cut into plans): 16 x 256 MiB of zeros, 64 MiB more, and a
    text tail; libbz2 decodes the stream back to exactly that, piece by piece"""
    import bz2
    from banzai_amd import corpus
    zeros = np.zeros(256 << 20, dtype=np.uint8)
    tail = corpus.enwik_synthetic(3_000_000, seed=77)
    total = 16 * zeros.size + (64 << 20) + tail.size
    assert total > 1 << 32
    out = bytearray()
    with native.Context(0, 9, 64) as ctx:
        ctx.stream_begin()
        for _ in range(16):
            out += ctx.stream_feed(zeros)
        out += ctx.stream_feed(zeros[:64 << 20])
        out += ctx.stream_feed(tail, eof=True)
        assert ctx.stream_consumed() == total
    assert bytes(out[:4]) == b"BZh9"
    dec = bz2.BZ2Decompressor()
    pos, nz = 0, 16 * zeros.size + (64 << 20)
    data = bytes(out)
    got_tail = bytearray()
    feed = data
    while not dec.eof:
        piece = dec.decompress(feed, 64 << 20)
        feed = b""
        if not piece and dec.needs_input:
            break
        if pos + len(piece) <= nz:
            assert piece.count(0) == len(piece), pos
        else:
            k = max(0, nz - pos)
            assert piece[:k].count(0) == k
            got_tail += piece[k:]
        pos += len(piece)
    assert dec.eof and pos == total and bytes(got_tail) == tail.tobytes()


def test_rust_facade_twin(oracle, native, tmp_path):
    """tests/abi_facade.c = the calling sequence of rust/src/lib.rs (banzai::encode over the C ABI) executed in C: 8 KiB
    fill_buf slices coalesced into a 4 MiB stage, large slices passed through only while the stage is empty, eof fed with
    the staged remainder (also when it is empty), bzh_stream_bound before every feed.  Streams equal the oracle's."""
    import subprocess
    from banzai_amd import corpus
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "abi_facade"
    lib_dir = os.path.join(ROOT, "banzai_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", str(exe), os.path.join(ROOT, "tests", "abi_facade.c"),
                           "-L" + lib_dir, "-lbzhip", "-Wl,-rpath," + lib_dir])
    text = corpus.enwik_synthetic(9_500_000, seed=61).tobytes()
    jobs = [(9, 8192, text),                                   # BufReader default: every slice goes through the stage
            (9, 8192, text[:8 * (4 << 20)]),                   # the input ends exactly on a stage boundary: eof with an empty remainder
            (9, 16 << 20, text),                               # encode_file: slices pass straight through
            (9, (4 << 20) + 12345, text[:9_000_001]),          # large slices, ragged last one (staged, then eof)
            (1, 8192, cases.gen(700_000, "longruns", 6)),      # run-heavy, level 1
            (5, 8192, b""), (9, 8192, b"x"), (2, 1 << 20, cases.gen(1_234_567, "shortruns", 2))]
    for level, slice_bytes, data in jobs:
        fin, fout = tmp_path / "in.bin", tmp_path / "out.bz2"
        fin.write_bytes(data)
        r = subprocess.run([str(exe), str(level), str(slice_bytes), str(fin), str(fout)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip() == f"consumed {len(data)}"
        assert fout.read_bytes() == oracle.encode(data, level), (level, slice_bytes, len(data))
    # the context pool: three calls in one process (the second and third find the context, its output vector and its stage in
    # the pool), through the in-memory reader (slice 0: fill_buf hands out all that is left, as Rust's &[u8] does) and through
    # 8 KiB slices; every call writes the same bytes
    for slice_bytes in (0, 8192):
        fin, fout = tmp_path / "in.bin", tmp_path / "out.bz2"
        fin.write_bytes(text)
        r = subprocess.run([str(exe), "9", str(slice_bytes), str(fin), str(fout), "3"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        assert len(lines) == 4 and lines[-1] == f"consumed {len(text)}" and all(ln.startswith("call ") for ln in lines[:3])
        assert fout.read_bytes() == oracle.encode(text, 9)


def test_full_size_headline_bit_exact_vs_oracle(oracle, native):
    """BASELINE config 3 at full size, the whole 100,000,000-byte workload of bench.py against the oracle (not only
    its size-independent properties): a mismatch shows up as a red test, not as a failed bench run."""
    import torch
    from banzai_amd import corpus
    n = 100_000_000
    data, _ = corpus.workload(n)
    want = oracle.encode(data.tobytes(), 9)
    with native.Context(0, 9, 128) as ctx:
        d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
        d_in[:n] = torch.from_numpy(data).cuda()
        cap = (n // 2 + (1 << 20)) & ~3
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        assert d_out[:ln].cpu().numpy().tobytes() == want


def test_full_size_other_workloads_bit_exact_vs_oracle(oracle, native):
    """every other workload of the bench line at the size bench.py runs it -- BASELINE config 5 as its four 25 MB quarters
    (zeros, a 1,024-byte tile, "ab", runs around 255), the corpora taken from the image (Python sources, shared libraries,
    100 MB of real text) -- whole streams against the oracle (which runs on a thread per workload: the C library releases
    the GIL).  Corpora the box does not hold are skipped one by one."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from banzai_amd import corpus
    sets = [(name, data) for name, data in corpus.c5_parts(100_000_000)]
    sets += [(name, corpus.image_corpus(name)) for name in corpus.IMAGE_SETS]
    sets = [(name, np.ascontiguousarray(data)) for name, data in sets if data.size >= 1_000_000]
    assert len(sets) >= 4
    with ThreadPoolExecutor(max_workers=len(sets)) as pool:
        wants = [pool.submit(oracle.encode, data.tobytes(), 9) for _, data in sets]
        with native.Context(0, 9, 128) as ctx:
            for (name, data), want in zip(sets, wants):
                n = int(data.size)
                d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
                d_in[:n] = torch.from_numpy(data).cuda()
                cap = (n + n // 4 + (1 << 20)) & ~3
                d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
                ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
                assert d_out[:ln].cpu().numpy().tobytes() == want.result(), name
                del d_in, d_out


@pytest.mark.parametrize("init", ["msd", "lsd", "msd nomid"])
def test_both_initial_sorts(native, init):
    """the bucket-first initial sort of bwt_msd.h (the default for text-like blocks) and the 8-pass sort forced on every
    block (BZH_INIT=lsd) are both bit-exact against the oracle on blocks and streams that reach every part of them
    (scripts/gpu_msd_check.py, a process of its own: the switch is read once per process); "msd nomid": the bucket-first sort
    with the large groups on the global radix passes (BZH_MID=0) instead of mid_sort -- the path blocks with a group that
    spans several units always take"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_msd_check.py")] + init.split(), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.parametrize("init", ["msd", "lsd", "default"])
def test_exactly_periodic_text_blocks(native, init):
    """a stretch of text repeated k times (2 .. 113), shorter than a block: groups of k identical rotations that only the tie
    rule orders, on both initial sorts and through both paths of the large groups (scripts/gpu_exact_period.py, a process of
    its own: the switch is read once per process)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_exact_period.py"), init], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "bad: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_coverage_guided_corpus(oracle, native):
    """the corpus a coverage-guided fuzzing session of the CPU oracle left (tests/golden/fuzz_corpus.zip, oracle/covfuzz.c)
    through the HIP path: every stream equal to the oracle's, bit for bit, at the level the input names"""
    import zipfile
    with zipfile.ZipFile(os.path.join(os.path.dirname(__file__), "golden", "fuzz_corpus.zip")) as z:
        items = [(n, z.read(n)) for n in sorted(z.namelist())]
    ctxs = {}
    try:
        for name, blob in items:
            level, data = 1 + blob[0] % 9, blob[1:]
            if level not in ctxs:
                ctxs[level] = native.Context(0, level, 8)
            assert ctxs[level].encode(data) == oracle.encode(data, level), (name, level, len(data))
    finally:
        for c in ctxs.values():
            c.close()


@pytest.mark.gpu
def test_bench_multi_rank_flow_over_rccl_with_one_rank():
    """The N > 1 flow of `bench.py` with its real transport, as far as one GPU can run it: BZH_BENCH_DIST_WORLD1=1 makes
    the single rank initialise the process group on RCCL (backend "nccl") with the gloo side group beside it and take the
    sharded path -- broadcast, all_gather, gather, all_reduce and all_gather_into_tensor on device tensors,
    `encode_sharded` with the chain hand-off over the side group -- and the stream must equal the plain one-GPU encode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BZH_BENCH_DIST_WORLD1="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BZH_BENCH_SHARED_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "4000000", "--no-extra", "--bytes", "30000000"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["collective_backend"] == "nccl" and line["ranks_seen"] == 1 and len(line["per_rank"]) == 1
    assert line["checks"]["sharded_equals_single_gpu"] and line["checks"]["libbz2_roundtrip"]
    assert line["checks"]["bit_exact_vs_oracle_sample"] and all(line["checks"].values())
    assert line["per_rank"][0]["encoded_input_bytes"] == 30000000


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--bytes", "30000000"], ["--total-bytes", "70000000"]])
def test_bench_multi_rank_flow_on_one_gpu(extra):
    """`bench.py --gpus 3` end to end on a box with ONE GPU: three real processes, every rank computing on cuda:0, the
    collectives over gloo on host tensors (BZH_BENCH_SHARED_GPU=1; RCCL refuses several ranks on one device) -- the
    script's multi-rank flow (self-launch, residency, chain over the side group, all-gather + gather, assembly, timing
    reductions, per-rank rows) and its checks: the sharded stream equals one GPU's, libbz2 decodes it, the oracle agrees"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BZH_BENCH_SHARED_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "4000000", "--no-extra"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["ranks_seen"] == 3 and len(line["per_rank"]) == 3
    assert line["scaling"] == ("strong" if extra[0] == "--total-bytes" else "weak")
    assert line["checks"]["sharded_equals_single_gpu"] and line["checks"]["libbz2_roundtrip"]
    assert line["checks"]["bit_exact_vs_oracle_sample"] and all(line["checks"].values())
    assert sum(r["encoded_input_bytes"] for r in line["per_rank"]) == (70000000 if extra[0] == "--total-bytes" else 90000000)


@pytest.mark.gpu
def test_bench_world_eight_on_one_gpu():
    """World 8 as far as one GPU allows, so that the driver's 8-GPU run is not the first time eight ranks meet: EIGHT real
    processes through `bench.py --gpus 8 --total-bytes 400000000` (BASELINE config 4's shape at 0.4 x: one stream over
    eight ranks, 55 blocks a rank), all computing on cuda:0 with the collectives over gloo on host tensors
    (BZH_BENCH_SHARED_GPU=1) -- the chain of eight 8-byte hand-offs over the side group, the meta row of the
    all-gather at its world-8 width, eight slabs gathered and funnel-shifted on rank 0, the per-rank rows, and the
    script's checks: the sharded stream equals one GPU's encode of the same 400 MB and the oracle agrees on its sample."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BZH_BENCH_SHARED_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                        "--cpu-sample", "4000000", "--no-extra", "--total-bytes", "400000000"], capture_output=True, text=True,
                       timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and len(line["per_rank"]) == 8 and line["scaling"] == "strong"
    assert line["checks"]["sharded_equals_single_gpu"] and line["checks"]["bit_exact_vs_oracle_sample"] and all(line["checks"].values())
    assert sum(r["encoded_input_bytes"] for r in line["per_rank"]) == 400000000
    # ranges shrink with the rank (each rank waits for the splits before it) and rank 0's is discounted for the gather
    own = [r["encoded_input_bytes"] for r in line["per_rank"]]
    assert max(own) < 1.15 * min(own) and all(r["ms_encode"] > 0 for r in line["per_rank"])
    # the chain's arithmetic from rank 0's calibration stands beside the clocks: later ranks wait longer, in the model too
    cal = line["chain_calibration"]
    assert 0 < cal["split_ms_per_block"] < cal["encode_ms_per_block"] and 0 <= cal["plan_cost"] <= 0.05
    waits = [r["model"]["wait"] for r in line["per_rank"]]
    assert waits[0] == 0 and all(a < b for a, b in zip(waits, waits[1:]))


def test_multi_device_handle_bit_exact(oracle, native):
    """bzh_create_multi with ONE GPU listed several times (one context and host thread per entry: what an 8-GPU node runs with
    eight different devices): the chained split over host variables, the strings copied to the first entry's device and
    assembled there.  Streams equal the oracle's -- text, runs that make a block span several workers' ranges, inputs shorter
    than the number of workers, the empty input; the three-step form (load / run / fetch) gives the same bytes."""
    from banzai_amd import corpus
    text = corpus.enwik_synthetic(30_000_000, seed=77).tobytes()
    runs = cases.gen(3_000_000, "text", 9) + b"\0" * 60_000_000 + cases.gen(2_000_000, "shortruns", 9) + b"ab" * 1_500_000
    with native.MultiContext([0, 0, 0], 9) as m:
        for data in (text, runs, b"", b"x", b"xy" * 3, text[:2_500_000]):
            want = oracle.encode(data, 9)
            assert m.encode(data) == want, len(data)
        m.load(text)
        n1 = m.run()
        n2 = m.run()  # (a loaded input can be encoded again: bench.py --single-process times this call)
        assert n1 == n2 and m.fetch(n1) == oracle.encode(text, 9)
        assert len(m.times()) == 3
    with native.MultiContext([0, 0], 1) as m:  # level 1: a hundred blocks a worker
        data = cases.gen(9_000_000, "text", 3) + cases.repeats(1_000_000, 3)
        assert m.encode(data) == oracle.encode(data, 1)
    with native.MultiContext([0] * 8, 9) as m:  # eight workers, fewer blocks than workers in some ranges
        data = text[:5_000_000]
        assert m.encode(data) == oracle.encode(data, 9)


def test_multi_device_handle_400mb(oracle, native):
    """the same at 400 MB over three workers: equal to the one-device stream AND to the oracle's"""
    import numpy as np
    import torch
    from banzai_amd import corpus
    data = np.concatenate([corpus.workload(100_000_000, segment=k)[0] for k in range(4)])
    n = int(data.size)
    with native.MultiContext([0, 0, 0], 9) as m:
        got = m.encode(data)
    with native.Context(0, 9) as ctx:
        d_in = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
        d_in[:n] = torch.from_numpy(data).cuda()
        cap = (n // 2 + (1 << 20)) & ~3
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        ln = ctx.encode_device(d_in.data_ptr(), n, d_out.data_ptr(), cap)
        single = d_out[:ln].cpu().numpy().tobytes()
    assert got == single
    assert got == oracle.encode(data.tobytes(), 9)


def test_multi_device_public_api_and_errors(oracle, native):
    """banzai_amd.encode(reader, writer, level, devices=[...]) and $BZHIP_DEVICES; a device that does not exist fails the
    creation with a status, not a crash"""
    import io
    import banzai_amd
    data = cases.gen(4_000_000, "text", 12) + b"\0" * 300_000
    want = oracle.encode(data, 9)
    out = io.BytesIO()
    assert banzai_amd.encode(io.BytesIO(data), out, 9, devices=[0, 0]) == len(data) and out.getvalue() == want
    os.environ["BZHIP_DEVICES"] = "0,0,0"
    try:
        out = io.BytesIO()
        assert banzai_amd.encode(io.BytesIO(data), out, 9) == len(data) and out.getvalue() == want
    finally:
        del os.environ["BZHIP_DEVICES"]
    with pytest.raises(native.BzhError):
        native.MultiContext([0, 63], 9)
    with pytest.raises(native.BzhError):
        native.MultiContext([], 9)
    # a slab that turns out too small (the slab size is a heuristic): the worker repeats its encode once with twice the room;
    # a slab that is still too small then fails the call with BZH_E_CAP on every worker's behalf, and the handle stays usable
    from banzai_amd import corpus
    text = corpus.enwik_synthetic(24_000_000, seed=91).tobytes()
    want = oracle.encode(text, 9)
    with native.MultiContext([0, 0, 0], 9) as m:
        m.debug_slab(1_600_000)  # (a worker's 8 MB of text need about 2.2 MB)
        assert m.encode(text) == want
        m.debug_slab(400_000)
        with pytest.raises(native.BzhError) as ei:
            m.encode(text)
        assert ei.value.status == -4
        m.debug_slab(0)
        assert m.encode(text) == want


def test_near_periodic_from_the_start(oracle, native):
    """period_detect / period_expand (bwt.hip): a block that is a word repeated, cut off inside a repetition, is sorted as eight
    of its periods and expanded.  Whole streams against the oracle: periods from 2 to the limit of 8,192 and just beyond it,
    small alphabets (partial matches between phases: the tails that read like another phase after their wrap), what RLE1
    leaves of one enormous run (period 5), blocks with barely enough repetitions and one too few, level 1 and level 9, periodic
    stretches between text (blocks that are periodic only in part must take the general sort)."""
    rng = np.random.default_rng(606)
    inputs = []
    for p, sigma, reps in ((2, 2, 700_000), (3, 2, 400_000), (7, 3, 200_000), (31, 2, 40_000), (1024, 256, 1_500), (1000, 4, 1_200),
                           (8191, 2, 160), (8192, 256, 130), (8193, 256, 130), (4096, 3, 9), (4096, 3, 11), (60_000, 256, 14)):
        w = rng.integers(0, sigma, p, dtype=np.uint8)
        # (no run of four equal bytes: RLE1 must leave the repetition alone, or the block's period is another one)
        for q in range(3, p):
            if w[q] == w[q - 1] == w[q - 2] == w[q - 3]:
                w[q] = (int(w[q]) + 1) % max(2, sigma)
        data = np.tile(w, reps).tobytes()
        inputs.append(data[: len(data) - int(rng.integers(1, p))] if p > 1 else data)
    inputs.append(b"\0" * 30_000_000)                                  # RLE1 output: 00 00 00 00 FB repeated
    inputs.append(b"A" * 255 + b"z" * 256 + b"A" * 257 + b"zzz" + b"AAAA" + b"z" * 5)
    inputs[-1] = inputs[-1] * 12_000
    text = cases.gen(1_200_000, "text", 4)
    inputs.append(text + b"ab" * 900_000 + text[:500_000] + b"xyz" * 400_000)
    for level in (9, 1):
        with native.Context(0, level, 32) as ctx:
            for data in inputs:
                if level == 1 and len(data) > 3_000_000:
                    data = data[:3_000_000]
                assert ctx.encode(data) == oracle.encode(data, level), (level, len(data), data[:16])
