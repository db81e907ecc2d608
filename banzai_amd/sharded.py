"""Block-sharded encode across ranks (one process per GPU), SURVEY.md section 8(e).

bzip2 blocks are independent once the sequential split is known, so the path shards with no
data-path collective: every rank runs the split on the whole input (replicated, cheap), encodes a
contiguous range of blocks into a bit string that starts at bit 0, and one gather brings the bit
strings to rank 0, which funnel-shifts them into the stream and folds the stream CRC in block
order (reference lib/lib.rs:101-126 carries only `raw`, `stream_crc`, `consumed` and the bit
cursor between iterations).

The `engine` is the per-rank compute: on a GPU box it is DeviceEngine (libbzhip.so through the
C ABI); the CPU tests pass their own engine so that partitioning, gather order and assembly are
exercised under gloo without a GPU.  An engine provides
    plan()                         -> [(in_off, in_len, rle_len, crc)]  (identical cuts on every rank; the crc
                                      field may be 0: block CRCs are only needed from the rank that encodes)
    encode_range(b0, b1)           -> (buffer tensor uint8 [cap], nbits)
    crcs(b0, b1)                   -> [crc] of blocks [b0, b1), valid after encode_range(b0, b1)
    assemble(segments, crcs)       -> stream length; segments = [(tensor, nbits)] in rank order,
                                      crcs = all block CRCs in block order
    cap                            -> fixed gather slab size in bytes (same on every rank)
"""


def block_range(nblocks, rank, world):
    """Contiguous share of rank: blocks [b0, b1)."""
    return rank * nblocks // world, (rank + 1) * nblocks // world


def encode_sharded(engine, dist=None, rank=0, world=1):
    """Run one sharded encode.  Returns the stream length on rank 0 (0 elsewhere)."""
    import torch

    blocks = engine.plan()
    nblk = len(blocks)
    b0, b1 = block_range(nblk, rank, world)
    part, nbits = engine.encode_range(b0, b1)
    own = engine.crcs(b0, b1)
    if world == 1:
        return engine.assemble([(part, nbits)], own)
    # one small all-gather carries every rank's bit count and block CRCs (ranges differ by at most one block)
    width = 1 + (nblk + world - 1) // world
    meta = torch.zeros(width, dtype=torch.int64, device=part.device)
    meta[0] = nbits
    if own:
        meta[1:1 + len(own)] = torch.tensor(own, dtype=torch.int64, device=part.device)
    allmeta = torch.zeros(world * width, dtype=torch.int64, device=part.device)
    dist.all_gather_into_tensor(allmeta, meta)
    rows = allmeta.view(world, width).tolist()  # one read-back
    nb = [int(r[0]) for r in rows]
    # gather only as many bytes as the longest bit string needs (whole 32-bit words), not the slab capacity
    used = min(engine.cap, (max(nb) + 31) // 32 * 4)
    send = part[:used]
    slabs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, slabs, dst=0)  # encoded blocks -> rank 0 (RCCL over xGMI on a GPU node)
    if rank != 0:
        return 0
    crcs = []
    for k in range(world):
        k0, k1 = block_range(nblk, k, world)
        crcs += [int(c) for c in rows[k][1:1 + (k1 - k0)]]
    segs = [(slabs[k], nb[k]) for k in range(world)]
    return engine.assemble(segs, crcs)


class DeviceEngine:
    """libbzhip.so on this rank's GPU; input already resident in HBM (tensor d_in, n bytes)."""

    def __init__(self, ctx, d_in, n, d_out, seg_cap):
        import torch

        self.ctx, self.d_in, self.n, self.d_out = ctx, d_in, n, d_out
        self.cap = seg_cap
        self.part = torch.zeros(seg_cap, dtype=torch.uint8, device=d_in.device)

    def plan(self):
        return self.ctx.plan_device(self.d_in.data_ptr(), self.n, crc=False)

    def crcs(self, b0, b1):
        return [b[3] for b in self.ctx.plan_blocks()[b0:b1]]  # encode_range_device computed them

    def encode_range(self, b0, b1):
        nbits = self.ctx.encode_range_device(b0, b1, self.part.data_ptr(), self.cap)
        return self.part, nbits

    def assemble(self, segments, crcs):
        segs = [(t.data_ptr(), nb) for t, nb in segments]
        return self.ctx.assemble_device(segs, crcs, self.d_out.data_ptr(), self.d_out.numel())
