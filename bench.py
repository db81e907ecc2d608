#!/usr/bin/env python3
"""bench.py -- encode MB/s (input) at level 9 on enwik8(-synthetic), 1/2/4/8 MI355X.

A "step" = one pass of the whole hot path (RLE1 split -> BWT -> MTF/RLE2 -> Huffman -> bit pack ->
stream assembly) over the workload, inputs resident in HBM when the timed region starts, the
finished .bz2 stream resident in HBM on rank 0 when it ends.

N = 1: 100,000,000 bytes (BASELINE.json configs[2]) through bzh_encode_device.
N > 1: weak scaling, N x 100,000,000 bytes as ONE stream: every rank holds the whole input
(all-gathered once, untimed), runs the cheap sequential block split itself (replicated, no
exchange), encodes its contiguous share of the blocks, and the encoded bit strings are gathered
to rank 0 over RCCL (torch.distributed backend "nccl") and funnel-shifted into the stream there.

Output: ONE JSON line on rank 0 (see the driver contract), with `roofline` for the dominant
kernel (radix_scatter) and `cpu_baseline` (the oracle = single-thread C restatement of banzai's
path, timed on a bounded sample of the same workload; also the bit-exactness check).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEGMENT = 100_000_000          # bytes per GPU (enwik8 size)
LEVEL = 9
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SORT_BYTES_PER_ELEM = 16.0     # one radix pass moves an 8-byte (key, suffix) pair: read 8 + write 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=SEGMENT, help="bytes per GPU")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="bytes of the workload timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from banzai_amd import _native as nv
    from banzai_amd import corpus

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    seg_bytes = args.bytes
    total = seg_bytes * world
    # ---- workload (untimed): each rank generates its own segment, ranks exchange once ----
    seg, wname = corpus.workload(seg_bytes, segment=rank)
    d_seg = torch.from_numpy(seg).to(dev)
    if world > 1:
        d_all = torch.empty(total + 16, dtype=torch.uint8, device=dev)
        parts = [d_all[k * seg_bytes:(k + 1) * seg_bytes] for k in range(world)]
        dist.all_gather(parts, d_seg)
        d_in = d_all
    else:
        d_in = torch.empty(total + 16, dtype=torch.uint8, device=dev)
        d_in[:total] = d_seg
    del d_seg
    out_cap = (total // 3 + total // 8 + (1 << 20)) & ~3
    d_out = torch.zeros(out_cap, dtype=torch.uint8, device=dev)

    ctx = nv.Context(local_rank, LEVEL, 128)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    from banzai_amd import sharded
    engine = None
    if world > 1:
        seg_cap = (seg_bytes // 3 + seg_bytes // 8 + (1 << 20)) & ~3
        engine = sharded.DeviceEngine(ctx, d_in, total, d_out, seg_cap)

    def step():
        """One pass of the hot path; returns the stream length on rank 0."""
        if world == 1:
            return ctx.encode_device(d_in.data_ptr(), total, d_out.data_ptr(), out_cap)
        return sharded.encode_sharded(engine, dist, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out_len = step()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    ctx.set_profiling(True)
    sort_ms = sort_launches = sort_elems = 0.0
    stage = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_len = step()
        st = ctx.stats()
        sort_ms += st["ms_bwt_sort"]
        sort_launches += st["bwt_sort_launches"]
        sort_elems += st["bwt_sort_elems"]
        nblocks = st["blocks"]
        for k in ("ms_plan", "ms_rle1", "ms_bwt", "ms_mtf", "ms_huff", "ms_pack"):
            stage[k] = stage.get(k, 0.0) + st[k]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt * 1e3 / args.steps
    value = total * args.steps / dt / 1e6

    result = None
    if rank == 0:
        stream_bytes = d_out[:out_len].cpu().numpy().tobytes()
        # ---- correctness (untimed): libbz2 round trip of the full stream when small enough, and
        # bit-exactness against the CPU oracle on the bounded sample that is also the cpu_baseline ----
        checks = {}
        if total <= 200_000_000:
            import bz2
            ref_in = d_in[:total].cpu().numpy().tobytes()
            checks["libbz2_roundtrip"] = bool(bz2.decompress(stream_bytes) == ref_in)
        if world > 1:
            # the sharded stream must equal what one GPU produces for the whole input (untimed)
            d_mono = torch.zeros(out_cap, dtype=torch.uint8, device=dev)
            ctx.set_profiling(False)
            mlen = ctx.encode_device(d_in.data_ptr(), total, d_mono.data_ptr(), out_cap)
            checks["sharded_equals_single_gpu"] = bool(mlen == out_len and torch.equal(d_mono[:mlen], d_out[:out_len]))
            del d_mono
        cpu = None
        if not args.no_cpu:
            from oracle import pyoracle as po
            sample_n = min(args.cpu_sample, seg_bytes)
            sample = d_in[:sample_n].cpu().numpy()
            t1 = time.perf_counter()
            oracle_stream = po.encode(sample.tobytes(), LEVEL)
            cpu_dt = time.perf_counter() - t1
            ctx.set_profiling(False)
            d_s_out = torch.zeros((sample_n // 2 + (1 << 20)) & ~3, dtype=torch.uint8, device=dev)
            slen = ctx.encode_device(d_in.data_ptr(), sample_n, d_s_out.data_ptr(), d_s_out.numel())
            checks["bit_exact_vs_oracle_sample"] = bool(d_s_out[:slen].cpu().numpy().tobytes() == oracle_stream)
            if total <= 200_000_000:  # the in-repo strict decoder (oracle/bz2_decode.c) on the whole GPU stream
                try:
                    checks["inrepo_decoder_roundtrip"] = bool(po.decode(stream_bytes, cap=total + 64) == ref_in)
                except po.DecodeError:
                    checks["inrepo_decoder_roundtrip"] = False
            cpu = {"value": round(sample_n / cpu_dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                   "sample": f"first {sample_n} bytes of the workload, level {LEVEL}, oracle/banzai_oracle.c -O2, "
                             f"1 thread of {os.cpu_count()} host cores"}
        # roofline of the dominant kernel (radix_scatter): algorithmic bytes / HIP-event time, per launch
        achieved = (SORT_BYTES_PER_ELEM * sort_elems / (sort_ms * 1e-3) / 1e9) if sort_ms > 0 else None
        # HBM bytes per launch from the committed PMC passes of this same command (FETCH_SIZE / WRITE_SIZE in
        # separate rocprofv3 runs, corrected as MI355X_MICROARCH.md prescribes); null if not collected
        traffic = None
        try:
            pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_traffic.json"))
            if pmc and world == 1 and seg_bytes == SEGMENT:
                with open(os.path.join(ROOT, "profiles", pmc[-1])) as f:
                    traffic = round(json.load(f)["radix_scatter_all"]["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        result = {
            "metric": "encode MB/s (input) at level 9, enwik8, 1/2/4/8 MI355X; bit-exact vs CPU",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic" if wname != "enwik8" else "enwik8",
            "config": {"workload": f"level {LEVEL} {wname}, {seg_bytes} bytes per GPU, {total} bytes in one stream, "
                                   "full RLE1->BWT->MTF->Huffman pipeline",
                       "blocks_on_rank0": int(nblocks),
                       "parallelism": f"block-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": "radix_scatter", "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
                         "launches": int(sort_launches), "avg_launch_us": round(sort_ms * 1e3 / max(1, sort_launches), 2),
                         "alg_bytes_per_launch": round(SORT_BYTES_PER_ELEM * sort_elems / max(1, sort_launches))},
            "cpu_baseline": cpu,
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage.items()},
            "compressed_bytes": out_len, "checks": checks,
        }
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None and not all(result["checks"].values()):
        raise SystemExit("bench: correctness check failed")


if __name__ == "__main__":
    main()
