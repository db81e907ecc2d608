"""Block-sharded encode across ranks (one process per GPU), SURVEY.md section 8(e).

bzip2 blocks are independent once the sequential split is known, so the path shards with no
data-path collective.  Ownership goes by START OFFSET: the input is divided at fixed byte offsets
B_0 = 0 < B_1 < ... < B_world = n and rank r encodes the blocks that start in [B_r, B_{r+1}).  A block's
cut depends only on the bytes before it (plus a little look-ahead), so rank r splits just the prefix
of the input up to B_{r+1} plus a margin -- the splitter marks the cuts that could still move if the
input went on ("open", the streaming rule), everything before the first open block is exact -- instead
of every rank splitting everything.  The split is the one sequential step (about 3 us per block on one
wavefront), so the low ranks, whose prefixes are short, get slightly longer ranges (`offsets`).
Each rank encodes its blocks into a bit string that starts at bit 0; one small all-gather (bit
counts, block CRCs) and one gather of the bit strings bring everything to rank 0, which funnel-shifts
them into the stream and folds the stream CRC in block order (reference lib/lib.rs:101-126 carries
only `raw`, `stream_crc`, `consumed` and the bit cursor between iterations).

The `engine` is the per-rank compute: on a GPU box it is DeviceEngine (libbzhip.so through the
C ABI); the CPU tests pass their own engine so that partitioning, gather order and assembly are
exercised under gloo without a GPU.  An engine provides
    n                              -> input length in bytes
    plan(prefix)                   -> ([(in_off, in_len, rle_len, crc)], [open]) for input[0:prefix]; the crc
                                      field may be 0 (block CRCs are only needed from the rank that encodes)
    encode_range(b0, b1)           -> (buffer tensor uint8 [cap], nbits) for blocks [b0, b1) of the last plan
    crcs(b0, b1)                   -> [crc] of those blocks, valid after encode_range(b0, b1)
    assemble(segments, crcs)       -> stream length; segments = [(tensor, nbits)] in rank order,
                                      crcs = all block CRCs in block order
    cap                            -> fixed gather slab size in bytes (same on every rank)
    min_block                      -> least number of input bytes a block consumes (bounds blocks per range)
"""

# cost of splitting one more input byte relative to encoding it (round 2, one-GPU simulation of 8 ranks on 800 MB:
# rank 7 plans its 800 MB prefix in 4.5 ms = 5.7 us/MB while encoding costs 136 us/MB): rank r+1's range is this
# much shorter than rank r's, which evens out split + encode over the ranks
PLAN_COST = 0.04
MARGIN = 4 << 20  # bytes planned beyond the end of the own range; grown when the last own cut is still open
# rank 0 also receives the gather and assembles the stream (~4 % of a step): its range is shortened by that much
ROOT_DISCOUNT = 0.96


def block_range(nblocks, rank, world):
    """Contiguous share of rank by block COUNT: blocks [b0, b1) (used where a full plan is at hand)."""
    return rank * nblocks // world, (rank + 1) * nblocks // world


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


def own_blocks(engine, rank, world):
    """Plan as far as needed and return (blocks, b0, b1): this rank's blocks are blocks[b0:b1] of that plan.
    `blocks` is whatever the engine's plan() returns: a list of (in_off, in_len, rle_len, crc) tuples or a structured
    numpy array with those fields (DeviceEngine: a high rank's prefix plan has ~1000 blocks per step)."""
    import numpy as np

    n = engine.n
    bounds = offsets(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    margin = MARGIN
    while True:
        prefix = n if rank == world - 1 else min(n, hi + margin)
        blocks, is_open = engine.plan(prefix)
        offs = blocks["in_off"] if isinstance(blocks, np.ndarray) else np.fromiter((b[0] for b in blocks), dtype=np.int64,
                                                                                   count=len(blocks))
        b0 = int(np.searchsorted(offs, lo, side="left"))  # first block that starts at or after lo
        b1 = int(np.searchsorted(offs, hi, side="left"))
        # exact if the plan saw the whole input, or if a block starting at/after `hi` exists whose predecessor's
        # cut is final (a cut is final once its block is not open; the cuts before a final cut are final too)
        if prefix == n or (b1 < len(blocks) and (b1 == 0 or not bool(is_open[b1 - 1]))):
            return blocks, b0, b1
        margin *= 4


class ShardError(RuntimeError):
    """Raised on EVERY rank when any rank's plan / encode failed (nobody is left waiting in a collective)."""


def encode_sharded(engine, dist=None, rank=0, world=1):
    """Run one sharded encode.  Returns the stream length on rank 0 (0 elsewhere)."""
    import torch

    # A rank that fails before the collectives must still take part in them, or the others wait for ever:
    # the failure travels as a status word in the all-gathered meta row and every rank raises together.
    err, part, nbits, own = None, None, 0, []
    try:
        blocks, b0, b1 = own_blocks(engine, rank, world)
        part, nbits = engine.encode_range(b0, b1)
        own = engine.crcs(b0, b1)
    except Exception as e:  # noqa: BLE001 -- whatever it was, the other ranks have to hear about it
        err = e
    if world == 1:
        if err is not None:
            raise err
        return engine.assemble([(part, nbits)], own)
    device = part.device if part is not None else getattr(engine, "device", "cpu")
    # one small all-gather carries every rank's bit count (or -1 = failed), block count and block CRCs
    bounds = offsets(engine.n, world)
    width = 2 + max(bounds[r + 1] - bounds[r] for r in range(world)) // max(1, engine.min_block) + 2
    row = [-1, 0] if err is not None else [nbits, len(own)] + list(own)
    if len(row) > width:
        err, row = ShardError(f"rank {rank}: {len(own)} blocks do not fit the meta row of {width} words"), [-1, 0]
    meta = torch.zeros(width, dtype=torch.int64)
    meta[:len(row)] = torch.tensor(row, dtype=torch.int64)
    meta = meta.to(device)
    allmeta = torch.zeros(world * width, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allmeta, meta)
    rows = allmeta.view(world, width).tolist()  # one read-back
    nb = [int(r[0]) for r in rows]
    failed = [k for k in range(world) if nb[k] < 0]
    if failed:
        raise ShardError(f"sharded encode failed on rank(s) {failed}" + (f"; this rank: {err!r}" if err is not None else ""))
    # gather only as many bytes as the longest bit string needs (whole 32-bit words), not the slab capacity
    used = min(engine.cap, (max(nb) + 31) // 32 * 4)
    send = part[:used]
    slabs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, slabs, dst=0)  # encoded blocks -> rank 0 (RCCL over xGMI on a GPU node)
    if rank != 0:
        return 0
    crcs = []
    for r in rows:
        crcs += [int(c) for c in r[2:2 + int(r[1])]]
    segs = [(slabs[k], nb[k]) for k in range(world)]
    return engine.assemble(segs, crcs)


def resident_bytes(n, rank, world, margin=64 << 20):
    """Input bytes rank `rank` has to hold: the prefix up to the end of its range plus the look-ahead the split
    may ask for (own_blocks grows its margin 4 -> 16 -> 64 MiB); the last rank holds everything."""
    if rank == world - 1:
        return n
    return min(n, offsets(n, world)[rank + 1] + margin)


def worst_case_slab(n, world, level=9):
    """Bytes that hold the bit string of any rank's range whatever the data: RLE1 expands by at most 5/4, the
    Huffman stage by less than 17/8 bits per symbol is never reached (<= 1.02 n + tables in practice, 5/4 n is
    safe), plus per-block headers and tables (< 4 KiB each)."""
    b = offsets(n, world)
    rng = max(b[r + 1] - b[r] for r in range(world))
    blocks = rng // ((100000 * level - 1) * 4 // 5) + 2
    return (rng + rng // 4 + blocks * 4096 + 65536 + 3) & ~3


class DeviceEngine:
    """libbzhip.so on this rank's GPU.  d_in holds the first `resident` bytes of the n-byte input (the rank's
    prefix, see resident_bytes; default: all of it)."""

    def __init__(self, ctx, d_in, n, d_out, seg_cap, resident=None):
        import torch

        self.ctx, self.d_in, self.n, self.d_out = ctx, d_in, n, d_out
        self.resident = n if resident is None else resident
        self.device = d_in.device
        self.cap = seg_cap
        self.part = torch.zeros(seg_cap, dtype=torch.uint8, device=d_in.device)
        # RLE1 expands by at most 5/4, so a block of M = 100000*level - 1 output bytes eats at least 0.8 M input
        # bytes (only the stream's last block may be shorter)
        self.min_block = (100000 * ctx.level - 1) * 4 // 5

    def plan(self, prefix):
        if prefix > self.resident:
            raise ShardError(f"the split needs {prefix} input bytes but only {self.resident} are resident on this rank")
        self.ctx.plan_device_only(self.d_in.data_ptr(), prefix, crc=False)
        return self.ctx.plan_blocks_np(), self.ctx.plan_open_np()

    def encode_range(self, b0, b1):
        nbits = self.ctx.encode_range_device(b0, b1, self.part.data_ptr(), self.cap)
        return self.part, nbits

    def crcs(self, b0, b1):
        return self.ctx.plan_blocks_np()["crc"][b0:b1].tolist()  # encode_range_device computed them

    def assemble(self, segments, crcs):
        segs = [(t.data_ptr(), nb) for t, nb in segments]
        return self.ctx.assemble_device(segs, crcs, self.d_out.data_ptr(), self.d_out.numel())
