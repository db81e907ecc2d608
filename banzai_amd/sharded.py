"""Block-sharded encode across ranks (one process per GPU), SURVEY.md section 8(e).

bzip2 blocks are independent once the sequential split is known, so the path shards with no
data-path collective.  Ownership goes by START OFFSET: the input is divided at fixed byte offsets
B_0 = 0 < B_1 < ... < B_world = n and rank r encodes the blocks that start in [B_r, B_{r+1}).

The split is CHAINED, not repeated: the loop of the reference (lib/lib.rs:101-126) carries only `raw`, the
stream CRC, `consumed` and the bit cursor from one block to the next, and a cut depends only on the bytes
from its block's start on -- so all rank r needs from rank r-1 is the offset s_r at which the first block at or
after B_r starts (one 8-byte send/recv).  Rank r holds input[B_r, B_{r+1} + look-ahead) and nothing else, builds
the split's run tables over that while s_r is on its way (they describe runs, not blocks), cuts blocks from s_r
until one starts at or after B_{r+1}, and forwards that start.  What remains sequential is the chain of the
splits themselves (about 2.4 us per block on one wavefront): rank r waits for the r splits before it, so the
ranges shrink slightly with the rank (`offsets`).
Each rank encodes its blocks into a bit string that starts at bit 0; one small all-gather (bit
counts, block CRCs) and one gather of the bit strings bring everything to rank 0, which funnel-shifts
them into the stream and folds the stream CRC in block order.

The `engine` is the per-rank compute: on a GPU box it is DeviceEngine (libbzhip.so through the
C ABI); the CPU tests pass their own engine so that partitioning, chain, gather order and assembly are
exercised under gloo without a GPU.  An engine provides
    n                              -> length of the whole input in bytes
    lo, resident                   -> the engine holds input[lo, lo + resident)
    tables()                       -> start whatever the split needs over the resident bytes (may do nothing)
    split(start, stop)             -> ([(in_off, in_len, rle_len, crc)], [open]): the blocks cut from ABSOLUTE offset
                                      `start` (a block start, >= lo) on, at least up to and including the first
                                      one that starts at or after `stop` (or to the end of the resident bytes),
                                      in_off absolute; the crc field may be 0; open[k] = the cut of block k could
                                      still move if the input went on (the streaming rule)
    encode_range(b0, b1)           -> (buffer tensor uint8 [cap], nbits) for blocks [b0, b1) of the last split
    crcs(b0, b1)                   -> [crc] of those blocks, valid after encode_range(b0, b1)
    assemble(segments, crcs)       -> stream length; segments = [(tensor, nbits)] in rank order,
                                      crcs = all block CRCs in block order
    cap                            -> fixed gather slab size in bytes (same on every rank)
    min_block                      -> least number of input bytes a block consumes (bounds blocks per range)
"""
import time

# what a rank waits for the split of the rank before it, relative to encoding the same bytes (one wavefront cuts
# a 900 kB block in about 2.4 us, the encode takes about 105 us, i.e. 0.023): rank r+1's range is this much shorter
# than rank r's -- half of what would even out wait + split + encode, so that the ranks' own work (split + encode)
# stays within a few per cent of each other
PLAN_COST = 0.012
LOOKAHEAD = 64 << 20  # bytes held beyond the end of the own range (a block inside one enormous run eats < 52 MB)
# rank 0 also receives the gather and assembles the stream (~4 % of a step): its range is shortened by that much
ROOT_DISCOUNT = 0.96


def block_range(nblocks, rank, world):
    """Contiguous share of rank by block COUNT: blocks [b0, b1) (used where a full plan is at hand)."""
    return rank * nblocks // world, (rank + 1) * nblocks // world


def set_plan_cost(split_ms_per_block, encode_ms_per_block):
    """Chooses PLAN_COST from a measurement (bench.py times one split and one encode of the first rank's bytes and
    broadcasts the two figures, so that every rank computes the same offsets): half the ratio, as above, kept inside
    [0, 0.05].  Must be called on every rank, with the same values, before anything asks for `offsets`."""
    global PLAN_COST
    if encode_ms_per_block > 0:
        PLAN_COST = min(0.05, max(0.0, 0.5 * split_ms_per_block / encode_ms_per_block))
    return PLAN_COST


def model_finish_ms(n, world, split_ms_per_block, encode_ms_per_block, block_bytes):
    """The chain's arithmetic, per rank: rank r starts cutting when the r ranks before it have cut (their split times add
    up), then cuts and encodes its own range -> [(wait, split, encode, finish)] in ms.  (Not in it: the 8-byte hops,
    the gather and the assembly on rank 0.)"""
    b = offsets(n, world)
    out, wait = [], 0.0
    for r in range(world):
        blocks = (b[r + 1] - b[r]) / max(1, block_bytes)
        sp, en = blocks * split_ms_per_block, blocks * encode_ms_per_block
        out.append((wait, sp, en, wait + sp + en))
        wait += sp
    return out


def offsets(n, world):
    """Range boundaries B_0..B_world (bytes): geometric lengths, ratio 1/(1 + PLAN_COST)."""
    q = 1.0 / (1.0 + PLAN_COST)
    w = [q ** r for r in range(world)]
    if world > 1:
        w[0] *= ROOT_DISCOUNT
    tot = sum(w)
    out, acc = [0], 0.0
    for r in range(world - 1):
        acc += w[r]
        out.append(int(n * acc / tot))
    out.append(n)
    return out


def resident_range(n, rank, world, lookahead=LOOKAHEAD):
    """[lo, hi): the input bytes rank `rank` holds -- its own range plus the look-ahead its last block may need."""
    b = offsets(n, world)
    return b[rank], min(n, b[rank + 1] + lookahead)


def resident_bytes(n, rank, world, lookahead=LOOKAHEAD):
    lo, hi = resident_range(n, rank, world, lookahead)
    return hi - lo


def own_blocks(engine, rank, world, start):
    """Cut this rank's blocks given the ABSOLUTE offset `start` its first block begins at.
    -> (blocks, b0, b1, next_start): the rank's blocks are blocks[b0:b1] of the engine's split; next_start = where
    the first block of the next range begins (n if there is none)."""
    import numpy as np

    n = engine.n
    bounds = offsets(n, world)
    hi = bounds[rank + 1]
    if start >= hi or start >= n:  # a block of an earlier rank runs over this whole range
        return [], 0, 0, start
    blocks, is_open = engine.split(start, hi)
    offs = blocks["in_off"] if isinstance(blocks, np.ndarray) else np.fromiter((b[0] for b in blocks), dtype=np.int64,
                                                                               count=len(blocks))
    b1 = int(np.searchsorted(offs, hi, side="left"))  # first block that starts at or after hi
    sees_end = engine.lo + engine.resident >= n
    # exact if the split saw the end of the input, or if a block starting at/after `hi` exists whose predecessor's
    # cut is final (a cut is final once its block is not open; the cuts before a final cut are final too)
    if not (sees_end or (b1 < len(blocks) and (b1 == 0 or not bool(is_open[b1 - 1])))):
        raise ShardError(f"rank {rank}: the look-ahead of {engine.lo + engine.resident - hi} bytes behind its range does "
                         "not settle its last cut")
    nxt = int(offs[b1]) if b1 < len(blocks) else n
    return blocks, 0, b1, nxt


class ShardError(RuntimeError):
    """Raised on EVERY rank when any rank's plan / encode failed (nobody is left waiting in a collective)."""


def side_group(dist):
    """The chain's own process group: `gloo`, CPU tensors.  The 8-byte hand-off from rank to rank (and the -1 that
    announces a failure) is latency, not bandwidth: over the `nccl` backend it would cost a lazily created
    point-to-point communicator per neighbour pair and a device synchronisation per hop, on the critical chain of an
    operation whose whole budget is a third of a millisecond per rank.  NCCL/RCCL keeps the two collectives that move
    data (one all_gather of the meta rows, one gather of the bit strings).  Collective: every rank calls this once."""
    return dist.new_group(backend="gloo")


def encode_sharded(engine, dist=None, rank=0, world=1, side=None):
    """Run one sharded encode.  Returns the stream length on rank 0 (0 elsewhere).  engine.times (if the engine has
    the attribute) receives this rank's milliseconds: tables + split, waiting for the chain, encode, gather.
    `side`: the process group of the chain hand-off (side_group(dist)); None = the default group, device tensors."""
    import torch

    # A rank that fails before the collectives must still take part in them, or the others wait for ever:
    # the failure travels down the chain as start -1 and as a status word in the all-gathered meta row, and every
    # rank raises together.
    err, part, nbits, own = None, None, 0, []
    t = {"ms_plan": 0.0, "ms_wait": 0.0, "ms_encode": 0.0, "ms_gather": 0.0}
    dev = getattr(engine, "device", "cpu")
    nxt, start = -1, 0
    t0 = time.perf_counter()
    try:
        engine.tables()
    except Exception as e:  # noqa: BLE001 -- whatever it was, the other ranks have to hear about it
        err = e
    t1 = time.perf_counter()
    hop_dev = "cpu" if side is not None else dev
    if rank > 0:  # (always: the rank before always sends)
        box = torch.zeros(1, dtype=torch.int64, device=hop_dev)
        dist.recv(box, src=rank - 1, group=side)
        start = int(box.item())
        if start < 0 and err is None:
            err = ShardError(f"rank {rank}: a rank before this one failed")
    t2 = time.perf_counter()
    if err is None:
        try:
            blocks, b0, b1, nxt = own_blocks(engine, rank, world, start)
        except Exception as e:  # noqa: BLE001
            err, nxt = e, -1
    t3 = time.perf_counter()
    t["ms_plan"] = ((t1 - t0) + (t3 - t2)) * 1e3
    t["ms_wait"] = (t2 - t1) * 1e3
    if world > 1 and rank < world - 1:
        dist.send(torch.tensor([nxt], dtype=torch.int64, device=hop_dev), dst=rank + 1, group=side)
    if err is None:
        t3 = time.perf_counter()
        for attempt in (0, 1):
            try:
                part, nbits = engine.encode_range(b0, b1)
                own = engine.crcs(b0, b1)
                break
            except Exception as e:  # noqa: BLE001
                # a slab that turned out too small (cannot happen with worst_case_slab's bound; a caller may pass its own): once more with twice the room
                if attempt == 0 and is_cap_error(e) and hasattr(engine, "grow"):
                    engine.grow()
                    t["retries"] = 1
                    continue
                err = e
                break
        t["ms_encode"] = (time.perf_counter() - t3) * 1e3
    if hasattr(engine, "times"):
        engine.times = t
    if world == 1:
        if err is not None:
            raise err
        return engine.assemble([(part, nbits)], own)
    device = part.device if part is not None else getattr(engine, "device", "cpu")
    # (a process group that moves host memory only -- gloo, e.g. two ranks sharing one GPU in a test -- gets host copies
    # of the device tensors; over RCCL the tensors go as they are)
    host_coll = str(getattr(device, "type", device)) == "cuda" and dist.get_backend() == "gloo"
    cdev = "cpu" if host_coll else device
    # one small all-gather carries every rank's bit count (or -1 = failed), block count and block CRCs
    bounds = offsets(engine.n, world)
    width = 2 + max(bounds[r + 1] - bounds[r] for r in range(world)) // max(1, engine.min_block) + 2
    row = [-1, 0] if err is not None else [nbits, len(own)] + list(own)
    if len(row) > width:
        err, row = ShardError(f"rank {rank}: {len(own)} blocks do not fit the meta row of {width} words"), [-1, 0]
    meta = torch.zeros(width, dtype=torch.int64)
    meta[:len(row)] = torch.tensor(row, dtype=torch.int64)
    meta = meta.to(cdev)
    allmeta = torch.zeros(world * width, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(allmeta, meta)
    rows = allmeta.view(world, width).tolist()  # one read-back
    nb = [int(r[0]) for r in rows]
    failed = [k for k in range(world) if nb[k] < 0]
    if failed:
        raise ShardError(f"sharded encode failed on rank(s) {failed}" + (f"; this rank: {err!r}" if err is not None else ""))
    # gather only as many bytes as the longest bit string needs (whole 32-bit words), not the slab capacity
    used = (max(nb) + 31) // 32 * 4
    if part.numel() < used:  # (another rank's bit string is longer than this rank's slab: pad, the tail is not read)
        bigger = torch.zeros(used, dtype=torch.uint8, device=part.device)
        bigger[:part.numel()] = part
        part = bigger
    send = part[:used].to(cdev)
    slabs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    tg = time.perf_counter()
    dist.gather(send, slabs, dst=0)  # encoded blocks -> rank 0 (RCCL over xGMI on a GPU node)
    if host_coll and rank == 0:
        slabs = [x.to(device) for x in slabs]
    if rank != 0:
        t["ms_gather"] = (time.perf_counter() - tg) * 1e3
        return 0
    crcs = []
    for r in rows:
        crcs += [int(c) for c in r[2:2 + int(r[1])]]
    segs = [(slabs[k], nb[k]) for k in range(world)]
    out_len = engine.assemble(segs, crcs)
    t["ms_gather"] = (time.perf_counter() - tg) * 1e3  # (rank 0: gather + assembly)
    return out_len


def is_cap_error(e):
    """the engine's "output buffer too small" (libbzhip.so: BZH_E_CAP)"""
    return getattr(e, "status", None) == -4 or getattr(e, "is_cap", False)


def min_lookahead(level):
    """Bytes a rank must hold beyond its own range for its last cut to be settled whatever the input: a block inside
    one enormous run consumes 255 bytes per 5 of its M output bytes (lib/rle.rs:210-234), plus one more chunk."""
    return 255 * ((100000 * level - 1) // 5 + 2)


def worst_case_slab(n, world, level=9):
    """Slab size for a rank's bit string -- a bound, not an estimate (round 6; the same arithmetic as csrc/multi.hip): RLE1
    expands a range by at most 5/4 (lib/rle.rs:210-234), MTF + RLE2 leave at most one symbol per RLE1 byte + EOB
    (lib/mtf.rs:36), no code is longer than 17 bits (lib/huffman.rs:293-296), a selector costs at most 6 bits per 50 symbols,
    and a block's header, symbol map and coding tables fit 4,400 bytes; a rank's last block may start in its range and end up
    to M RLE1 bytes behind it.  2.2 bytes per RLE1 byte covers 17 + 6/50 bits.  The retry with a doubled slab (BZH_E_CAP ->
    engine.grow) stays behind it."""
    b = offsets(n, world)
    rng = max(b[r + 1] - b[r] for r in range(world))
    M = 100000 * level - 1
    blocks = rng // (M * 4 // 5) + 2
    rle = rng + rng // 4 + blocks * 8
    return ((rle + M) * 22 // 10 + blocks * 4400 + 65536 + 3) & ~3


class DeviceEngine:
    """libbzhip.so on this rank's GPU.  d_in holds input[lo, lo + resident) of the n-byte input (see resident_range;
    default: all of it)."""

    def __init__(self, ctx, d_in, n, d_out, seg_cap, resident=None, lo=0):
        import torch

        self.ctx, self.d_in, self.n, self.d_out = ctx, d_in, n, d_out
        self.lo = lo
        self.resident = n if resident is None else resident
        if self.lo + self.resident > n or self.resident < 0:
            raise ValueError("resident range outside the input")
        self.device = d_in.device
        self.cap = seg_cap
        self.times = {}
        self.part = torch.zeros(seg_cap, dtype=torch.uint8, device=d_in.device)
        # RLE1 expands by at most 5/4, so a block of M = 100000*level - 1 output bytes eats at least 0.8 M input
        # bytes (only the stream's last block may be shorter)
        self.min_block = (100000 * ctx.level - 1) * 4 // 5

    def grow(self):
        """twice the slab (after BZH_E_CAP)"""
        import torch

        self.cap *= 2
        self.part = torch.zeros(self.cap, dtype=torch.uint8, device=self.d_in.device)

    def check_lookahead(self, rank, world):
        """Refuses, up front and by name, a resident range whose look-ahead cannot settle every possible last cut
        (own_blocks would raise later, on the input that needs it)."""
        hi = offsets(self.n, world)[rank + 1]
        have = self.lo + self.resident - hi
        need = min_lookahead(self.ctx.level)
        if self.lo + self.resident < self.n and have < need:
            raise ShardError(f"rank {rank} holds {have} bytes beyond its range; level {self.ctx.level} needs {need} "
                             f"(sharded.min_lookahead) or the rest of the input")

    def tables(self):
        self.ctx.plan_tables_device(self.d_in.data_ptr(), self.resident)

    def split(self, start, stop=None):
        if start < self.lo or start > self.lo + self.resident:
            raise ShardError(f"the split starts at {start}, outside the resident bytes [{self.lo}, {self.lo + self.resident})")
        self.ctx.plan_split_device(start - self.lo, None if stop is None else max(0, stop - self.lo), crc=False)
        blocks = self.ctx.plan_blocks_np()
        blocks["in_off"] += self.lo
        return blocks, self.ctx.plan_open_np()

    def encode_range(self, b0, b1):
        nbits = self.ctx.encode_range_device(b0, b1, self.part.data_ptr(), self.cap)
        return self.part, nbits

    def crcs(self, b0, b1):
        return self.ctx.plan_blocks_np()["crc"][b0:b1].tolist()  # encode_range_device computed them

    def assemble(self, segments, crcs):
        segs = [(t.data_ptr(), nb) for t, nb in segments]
        return self.ctx.assemble_device(segs, crcs, self.d_out.data_ptr(), self.d_out.numel())
