"""CPU, world_size 2 over gloo: the block-sharded protocol of banzai_amd/sharded.py (range
partition, gather order, bit-level assembly, stream-CRC fold) reproduces the single stream.
The per-rank compute is a test engine built on the oracle (no GPU here); on a GPU box the same
protocol runs with sharded.DeviceEngine (covered by tests/test_gpu_parity.py on one device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases


def _strip_framing(stream):
    """block bits of a .bz2 stream: drop 'BZh9' (32 bits) and footer (48+32 bits) + padding."""
    bits = np.unpackbits(np.frombuffer(stream, dtype=np.uint8))
    # footer magic 0x177245385090 starts 80 bits before the (padded) end; find the last occurrence
    magic = np.unpackbits(np.frombuffer(bytes.fromhex("177245385090"), dtype=np.uint8))
    for pad in range(8):
        end = len(bits) - pad - 32
        if end - 48 >= 32 and np.array_equal(bits[end - 48:end], magic):
            return bits[32:end - 48]
    raise AssertionError("footer not found")


class OracleEngine:
    def __init__(self, oracle, data, level, cap, lo=0, resident=None):
        self.o, self.data, self.level, self.cap = oracle, data, level, cap
        self.n = len(data)
        self.lo = lo
        self.resident = self.n - lo if resident is None else resident
        self.min_block = (100000 * level - 1) * 4 // 5
        self.stream = None
        self.times = {}
        self.blocks = []
        _, infos = self.o.encode(self.data, self.level, want_blocks=True)
        self.full = [(int(b.in_off), int(b.in_len), int(b.rle_len), int(b.crc)) for b in infos]

    def tables(self):
        pass

    def split(self, start, stop=None):
        """what a splitter that only sees data[lo : lo + resident] can know when it starts at `start`: the blocks that
        end inside those bytes are exact; whatever is left forms one more block whose cut is still open.  Like the
        device engine, hand out the cuts without CRCs (a rank only knows the CRCs of the blocks it encodes)."""
        assert start in [b[0] for b in self.full], "the chain handed over something that is not a block start"
        end_seen = self.lo + self.resident
        if end_seen >= self.n:
            self.blocks = [b for b in self.full if b[0] >= start]
            return [(o, ln, r, 0) for o, ln, r, _ in self.blocks], [False] * len(self.blocks)
        self.blocks = [b for b in self.full if b[0] >= start and b[0] + b[1] <= end_seen]
        flags = [False] * len(self.blocks)
        end = self.blocks[-1][0] + self.blocks[-1][1] if self.blocks else start
        out = [(o, ln, r, 0) for o, ln, r, _ in self.blocks]
        if end < end_seen:
            out.append((end, end_seen - end, 0, 0))
            flags.append(True)
        return out, flags

    def crcs(self, b0, b1):
        return [b[3] for b in self.blocks[b0:b1]]

    def encode_range(self, b0, b1):
        buf = torch.zeros(self.cap, dtype=torch.uint8)
        if b0 == b1:
            return buf, 0
        lo, hi = self.blocks[b0][0], self.blocks[b1 - 1][0] + self.blocks[b1 - 1][1]
        bits = _strip_framing(self.o.encode(self.data[lo:hi], self.level))
        packed = np.packbits(bits)
        buf[:len(packed)] = torch.from_numpy(packed)
        return buf, int(len(bits))

    def assemble(self, segments, crcs):
        allbits = [np.unpackbits(np.frombuffer(b"BZh" + bytes([48 + self.level]), dtype=np.uint8))]
        for t, nb in segments:
            allbits.append(np.unpackbits(t.numpy())[:nb])
        s = 0
        for c in crcs:
            s = (c ^ (((s << 1) | (s >> 31)) & 0xFFFFFFFF)) & 0xFFFFFFFF
        allbits.append(np.unpackbits(np.frombuffer(bytes.fromhex("177245385090") + s.to_bytes(4, "big"), dtype=np.uint8)))
        self.stream = np.packbits(np.concatenate(allbits)).tobytes()
        return len(self.stream)


def _worker(rank, world, port, data, level, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from banzai_amd import sharded
    from oracle import pyoracle
    # every rank holds its own range plus a look-ahead only (long runs need a long one: a block inside a run of
    # 70,000 equal bytes eats them all)
    lo, hi = sharded.resident_range(len(data), rank, world, lookahead=400_000)
    eng = OracleEngine(pyoracle, data, level, cap=len(data) + 4096, lo=lo, resident=hi - lo)
    # the product's two-group layout: the chain hand-off on its own gloo group (CPU tensors), the collectives on the
    # default group (nccl on a GPU node, gloo here)
    side = sharded.side_group(dist)
    n = sharded.encode_sharded(eng, dist, rank, world, side=side)
    if rank == 0:
        q.put((n, eng.stream))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,n,world", [("text", 450_000, 2), ("longruns", 700_001, 2), ("random", 99_000, 2),
                                          ("shortruns", 1_000_003, 3)])
def test_ranks_reproduce_single_stream(oracle, mode, n, world):
    data = cases.gen(n, mode, 11)
    want = oracle.encode(data, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, 1, q)) for r in range(world)]
    for p in procs:
        p.start()
    n_out, stream = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n_out == len(want) and stream == want


def _failing_worker(rank, world, port, data, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from banzai_amd import sharded
    from oracle import pyoracle
    eng = OracleEngine(pyoracle, data, 1, cap=len(data) + 4096)
    if rank == 1:  # this rank's encode fails (as BZH_E_CAP / a plan error would on a GPU)
        def boom(b0, b1):
            raise RuntimeError("slab too small")
        eng.encode_range = boom
    try:
        sharded.encode_sharded(eng, dist, rank, world, side=sharded.side_group(dist))
        q.put((rank, "returned"))
    except sharded.ShardError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


class _SlabTooSmall(RuntimeError):
    is_cap = True  # what sharded.is_cap_error looks for (libbzhip.so: BZH_E_CAP)


def _retry_worker(rank, world, port, data, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from banzai_amd import sharded
    from oracle import pyoracle
    lo, hi = sharded.resident_range(len(data), rank, world, lookahead=400_000)
    small = 2048  # far too small for this rank's bit string: the first encode_range must fail with "cap"
    eng = OracleEngine(pyoracle, data, 1, cap=small if rank == 1 else len(data) + 4096, lo=lo, resident=hi - lo)
    grown = []
    if rank == 1:
        plain = eng.encode_range

        def capped(b0, b1):
            if eng.cap < len(data):
                raise _SlabTooSmall("slab too small")
            return plain(b0, b1)

        def grow():
            eng.cap = max(eng.cap * 2, len(data) + 4096)
            grown.append(eng.cap)
        eng.encode_range, eng.grow = capped, grow
    side = sharded.side_group(dist)
    n = sharded.encode_sharded(eng, dist, rank, world, side=side)
    q.put((rank, n, eng.stream if rank == 0 else None, len(grown), eng.times.get("retries", 0)))
    dist.barrier()
    dist.destroy_process_group()


def test_cap_error_is_retried_once_with_a_larger_slab(oracle):
    """a slab that is too small on one rank (BZH_E_CAP) costs one retry on that rank, not the job: the stream is
    still the single stream, and rank 0 copes with a bit string longer than its own slab"""
    data = cases.gen(450_000, "text", 8)
    want = oracle.encode(data, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_retry_worker, args=(r, 2, port, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r[0]: r for r in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == len(want) and got[0][2] == want
    assert got[1][3] == 1 and got[1][4] == 1 and got[0][3] == 0


def test_failure_on_one_rank_raises_on_all(oracle):
    """a rank that fails before the collectives reports through the status word: nobody hangs, everybody raises"""
    data = cases.gen(450_000, "text", 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "rank(s) [1]" in got[0] and "rank(s) [1]" in got[1] and "slab too small" in got[1]


def test_resident_range_and_slab_sizes():
    from banzai_amd import sharded
    n = 800_000_000
    b = sharded.offsets(n, 8)
    for r in range(8):
        lo, hi = sharded.resident_range(n, r, 8)
        assert lo == b[r] and hi == min(n, b[r + 1] + (64 << 20))
        # a rank holds its own range plus the look-ahead, never a prefix of the stream
        assert sharded.resident_bytes(n, r, 8) <= (b[r + 1] - b[r]) + (64 << 20)
    assert max(sharded.resident_bytes(n, r, 8) for r in range(8)) < n // 4
    assert sharded.worst_case_slab(n, 8) >= (max(b[k + 1] - b[k] for k in range(8)) * 5) // 4


def test_chained_ownership_covers_every_block_once(oracle):
    """the chain of own_blocks calls, every rank seeing only its own range plus a look-ahead, tiles the
    whole-input plan exactly; a look-ahead that cannot settle the last cut is refused, not guessed"""
    from banzai_amd import sharded
    for mode, n in (("text", 1_234_567), ("longruns", 900_000), ("same", 400_000), ("random", 50)):
        data = cases.gen(n, mode, 5)
        full = OracleEngine(oracle, data, 1, cap=16)
        for world in (1, 2, 3, 5, 8):
            b = sharded.offsets(n, world)
            assert b[0] == 0 and b[-1] == n and all(b[k] <= b[k + 1] for k in range(world))
            got, start = [], 0
            for r in range(world):
                # (a level-1 block inside long runs eats up to 5.1 MB of input: those inputs are held to their end)
                lo, hi = sharded.resident_range(n, r, world, lookahead=450_000 if mode in ("text", "random") else n)
                eng = OracleEngine(oracle, data, 1, cap=16, lo=lo, resident=hi - lo)
                blocks, b0, b1, start = sharded.own_blocks(eng, r, world, start)
                got += [blk[:3] for blk in blocks[b0:b1]]
            assert start == n and got == [blk[:3] for blk in full.full], (mode, world)
    data = cases.gen(900_000, "longruns", 5)  # blocks inside runs of 70,000 bytes: 10 kB of look-ahead settle nothing
    lo, hi = sharded.resident_range(len(data), 0, 4, lookahead=10_000)
    eng = OracleEngine(oracle, data, 1, cap=16, lo=lo, resident=hi - lo)
    with pytest.raises(sharded.ShardError):
        sharded.own_blocks(eng, 0, 4, 0)


def test_lookahead_is_validated_up_front():
    """a DeviceEngine whose look-ahead cannot settle every possible cut is refused by name before any input
    shows it (no GPU needed: the check only reads sizes)"""
    from banzai_amd import sharded

    class Ctx:
        level = 9

    n = 800_000_000
    need = sharded.min_lookahead(9)
    assert 45_000_000 < need < (64 << 20)  # the default look-ahead covers the worst case at level 9
    for look, ok in ((64 << 20, True), (need, True), (need - 1, False), (1 << 20, False)):
        lo, hi = sharded.resident_range(n, 2, 8, lookahead=look)
        eng = sharded.DeviceEngine.__new__(sharded.DeviceEngine)
        eng.ctx, eng.n, eng.lo, eng.resident = Ctx(), n, lo, hi - lo
        if ok:
            eng.check_lookahead(2, 8)
        else:
            with pytest.raises(sharded.ShardError, match="min_lookahead"):
                eng.check_lookahead(2, 8)
    lo, hi = sharded.resident_range(n, 7, 8, lookahead=0)  # the last rank sees the end of the input: nothing to hold
    eng = sharded.DeviceEngine.__new__(sharded.DeviceEngine)
    eng.ctx, eng.n, eng.lo, eng.resident = Ctx(), n, lo, hi - lo
    eng.check_lookahead(7, 8)


def test_block_range_partition():
    from banzai_amd.sharded import block_range
    for nb in (0, 1, 7, 111, 1112):
        for world in (1, 2, 4, 8):
            got = [block_range(nb, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == nb
            assert all(got[k][1] == got[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in got) - min(b - a for a, b in got) <= 1


def test_plan_cost_from_a_measurement_and_the_chain_model():
    """sharded.set_plan_cost / model_finish_ms (bench.py calibrates on rank 0 and broadcasts): the ranges follow the
    measured split / encode ratio, every rank computes the same offsets, and the model's wait of rank r is the sum of
    the splits before it."""
    from banzai_amd import sharded
    old = sharded.PLAN_COST
    try:
        assert sharded.set_plan_cost(0.002, 0.086) == pytest.approx(0.5 * 0.002 / 0.086)
        b = sharded.offsets(800_000_000, 8)
        lens = [b[k + 1] - b[k] for k in range(8)]
        assert b[0] == 0 and b[-1] == 800_000_000 and all(x > y for x, y in zip(lens[1:], lens[2:]))
        m = sharded.model_finish_ms(800_000_000, 8, 0.002, 0.086, 1_000_000)
        assert m[0][0] == 0 and m[3][0] == pytest.approx(sum(x[1] for x in m[:3]))
        assert all(abs(x[3] - (x[0] + x[1] + x[2])) < 1e-9 for x in m)
        assert sharded.set_plan_cost(1.0, 0.086) == 0.05 and sharded.set_plan_cost(0.0, 0.086) == 0.0  # clamped
        assert sharded.set_plan_cost(0.002, 0.0) == 0.0  # (no measurement: unchanged)
    finally:
        sharded.PLAN_COST = old
