/*
 * bzhip.h -- C ABI of libbzhip.so: an MI355X (gfx950) bzip2 block encoder that emits the
 * same bits as jgbyrne/banzai v0.3.1.
 *
 * The reference has no FFI; its boundary is the crate's public functions and the private
 * per-stage functions.  Each entry point below names the reference
 * interface it replaces.  Conventions: plain pointers and sizes, caller-owned buffers, int
 * status (0 = ok, negative = bzh_status), nothing unwinds across the ABI.  A context is bound
 * to ONE GPU (one process per GPU) and is single-threaded; distinct contexts may run
 * concurrently.  Pointers named d_* are DEVICE pointers (HBM); all others are host pointers.
 * There is no CPU fallback: every entry point that computes fails with BZH_E_HIP when no
 * gfx950 device is usable.
 */
#ifndef BZHIP_H
#define BZHIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(BZH_BUILD)
#define BZH_API __attribute__((visibility("default")))
#else
#define BZH_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bzh_ctx bzh_ctx;

typedef enum {
    BZH_OK = 0,
    BZH_E_ARG = -1,   /* bad argument (reference: assert!/panic!, lib/lib.rs:89) */
    BZH_E_NOMEM = -2, /* host or device allocation failed */
    BZH_E_HIP = -3,   /* HIP runtime error / no usable device */
    BZH_E_CAP = -4,   /* output buffer too small; *out_len holds the size needed where stated */
    BZH_E_STATE = -5  /* call sequence error (e.g. encode_range without a plan) */
} bzh_status;

/* One bzip2 block of a plan: which raw bytes it consumes and what RLE1 makes of them.
 * Mirrors `Rle { output, chk, raw, consumed }` (lib/rle.rs:94-99) minus the byte vectors. */
typedef struct {
    uint64_t in_off;  /* first raw byte of the block                                   */
    uint64_t in_len;  /* raw bytes consumed (Rle::consumed)                             */
    uint32_t rle_len; /* RLE1 output bytes, <= 100000*level-1 (lib/rle.rs:121)          */
    uint32_t crc;     /* CRC-32/BZIP2 of the raw bytes (Rle::chk, lib/crc32.rs:31-48)   */
} bzh_block;

/* Per-stage device timings (milliseconds, HIP events on the context's stream) and counters of
 * the last bzh_encode* / bzh_encode_range_device call; filled when profiling is enabled. */
typedef struct {
    double ms_plan, ms_rle1, ms_bwt, ms_mtf, ms_huff, ms_pack, ms_total;
    double ms_bwt_sort;           /* the global radix passes only (radix_scatter: the 8 passes of blocks that keep them and
                                   * the big-list passes of the rounds; text batches launch none since round 5 -- the
                                   * dominant kernel class is chunk_finish, see bzh_get_kernel_stats)               */
    uint64_t bwt_sort_launches;   /* radix scatter launches issued in ms_bwt_sort (incl. ones that find no work) */
    uint64_t bwt_sort_elems;      /* elements moved by those launches                           */
    uint64_t raw_bytes, rle_bytes, mtf_syms, out_bits;
    uint64_t bwt_rounds;          /* prefix-doubling rounds run (max over blocks)               */
    uint64_t bwt_active_sum;      /* sum over rounds and blocks of unresolved suffixes (A)      */
    uint32_t blocks, pad;
} bzh_stats;

/* ---- lifecycle -------------------------------------------------------------------------- */

/* Create a context on HIP device `device` for block size `level` (1..9; lib/lib.rs:89).
 * max_batch = bzip2 blocks processed per kernel batch (0 = default). */
BZH_API int bzh_create(bzh_ctx **ctx, int device, int level, int max_batch);
/* 1 if a device whose hipDeviceProp_t::gcnArchName is `gcn_arch_name` can run this library (gfx950 only;
 * e.g. "gfx950:sramecc+:xnack-"), else 0.  bzh_create applies it to the chosen device and fails with
 * BZH_E_HIP otherwise. */
BZH_API int bzh_arch_supported(const char *gcn_arch_name);
BZH_API void bzh_destroy(bzh_ctx *ctx);
BZH_API const char *bzh_strerror(int status);
BZH_API const char *bzh_last_error(const bzh_ctx *ctx); /* detail of the last failure on this ctx   */
BZH_API int bzh_set_stream(bzh_ctx *ctx, void *hip_stream); /* hipStream_t to launch on (default 0) */
BZH_API int bzh_set_profiling(bzh_ctx *ctx, int enabled);
/* Huffman stage behaviour.  BZH_MODE_REFERENCE (default): bit-identical to banzai 0.3.1, including its table
 * count rule (2 or 3 tables from the ALPHABET size, lib/huffman.rs:319-326) and the refinement loop that zeroes
 * the tables (lib/huffman.rs:402-409), after which every segment uses table 0.  BZH_MODE_FIXED (SURVEY.md 8f row
 * f4, opt-in, NOT bit-identical to the reference): 2..6 tables chosen from the number of symbols as libbz2 does,
 * four real refinement iterations, per-segment selectors.  Every stream is a valid bzip2 stream either way. */
enum { BZH_MODE_REFERENCE = 0, BZH_MODE_FIXED = 1 };
BZH_API int bzh_set_mode(bzh_ctx *ctx, int mode);
/* 1 (default): batches run one after the other on the context's stream.  2: two half-batch lanes on
 * internal streams and host threads.  Measured on the 100 MB headline (one batch split in two): 2 lanes are 3-5 % SLOWER
 * than 1 (round 5: 8.73-8.93 against 8.34-8.41 ms; the per-batch latency chains -- huff_build, the late doubling rounds -- are paid
 * twice and overlap less than they cost); kept for experiments only. */
BZH_API int bzh_set_lanes(bzh_ctx *ctx, int lanes);
BZH_API int bzh_get_stats(const bzh_ctx *ctx, bzh_stats *out);
/* Test hook: inject a fault into the next suffix sort of this context (kind 1: one tile of the first block never
 * publishes its look-back status; 0: none).  The call that runs into it returns an error status -- the waits of
 * the look-backs are bounded -- and the context stays usable.  Kind 2: the same fault, treated as the one a GPU
 * shared with other processes produces (a look-back of a small batch gives up): the sort runs again with every
 * block pinned to one XCD and the call succeeds.  No counterpart in the reference. */
BZH_API int bzh_debug_fault(bzh_ctx *ctx, int kind);

/* Per-kernel-class figures of the last whole-path call made with profiling on (bench.py's `roofline.kernels`):
 * HIP-event time of the class's launches on the context's stream, launches issued, and the ALGORITHMIC bytes they
 * moved (per-element figures in DESIGN.md section 4 x the element counts of the plan and the round summaries).
 * *count receives the number of classes; BZH_E_CAP if max is smaller.  The reference has no counterpart. */
typedef struct bzh_kstat {
    char name[48];
    double ms;
    uint64_t launches;
    uint64_t alg_bytes;
} bzh_kstat;
BZH_API int bzh_get_kernel_stats(const bzh_ctx *ctx, bzh_kstat *out, size_t max, size_t *count);

/* ---- whole path: replaces banzai::encode(reader, writer, level), lib/lib.rs:84-132 -------- */

/* Host buffers in and out (H2D + encode + D2H).  Produces the complete .bz2 stream for the
 * slice in[0..n) -- identical to encode() fed by a reader that yields the slice -- and
 * returns the bytes consumed (== n) in *consumed.  BZH_E_CAP if cap is too small. */
BZH_API int bzh_encode(bzh_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len,
               size_t *consumed);

/* Same, input and output resident in HBM (the timed path of bench.py). */
BZH_API int bzh_encode_device(bzh_ctx *ctx, const void *d_in, size_t n, void *d_out, size_t cap, size_t *out_len,
                      size_t *consumed);

/* ---- streaming: encode() fed by a reader that yields arbitrary chunks (lib/rle.rs:30-92) ------- */

/* Starts a stream on the context.  Then call bzh_stream_feed any number of times; the bytes it
 * returns, concatenated, are exactly the stream bzh_encode produces for the concatenated input,
 * whatever the chunking (the reference's incremental InputStream + margin_call, without its T16
 * truncation bug).  Memory stays bounded: input is buffered only until the blocks it completes
 * are final (at most about one block's worth of raw bytes is carried between feeds). */
BZH_API int bzh_stream_begin(bzh_ctx *ctx);

/* Feeds n more input bytes; eof != 0 marks the end of the input (n may be 0) and completes the
 * stream.  Writes the stream bytes that became final to out; *out_len receives their count.
 * The bytes are copied to the GPU at once (`in` may be reused on return).  Once a full batch is
 * pending, a pass (split + encode + copy back) starts on an internal thread and the call returns;
 * its bytes are handed out by the next call that starts a pass, or by the eof call -- so the
 * caller's reading and copying of the next chunks overlaps with the GPU.
 * cap must be >= bzh_stream_bound(ctx, n) evaluated right before the call; BZH_E_CAP is returned
 * before anything is consumed (the call can be repeated with a larger buffer). */
BZH_API int bzh_stream_feed(bzh_ctx *ctx, const uint8_t *in, size_t n, int eof, uint8_t *out, size_t cap,
                            size_t *out_len);

/* Upper bound of the bytes the next bzh_stream_feed(…, n, …) call can return (depends on what is
 * pending and in flight, so ask before every call). */
BZH_API size_t bzh_stream_bound(const bzh_ctx *ctx, size_t n);

/* Pending input that triggers a GPU pass (default: one full batch, max_batch blocks of raw input, at most
 * 128 MiB; smaller = lower latency and less buffering, but more and smaller launches). */
BZH_API int bzh_stream_set_chunk(bzh_ctx *ctx, size_t bytes);

/* Input bytes encoded so far (after the eof feed: the total, encode()'s return value). */
BZH_API size_t bzh_stream_consumed(const bzh_ctx *ctx);

/* ---- block-sharded path (one rank per GPU) -------------------------- */

/* Split d_in[0..n) into blocks: the sequential part of the loop at lib/lib.rs:101-126, i.e.
 * every rle_one() cut (lib/rle.rs:102-253) and block CRC, without encoding anything.
 * The plan stays in the context and references d_in (caller keeps it alive). */
BZH_API int bzh_plan_device(bzh_ctx *ctx, const void *d_in, size_t n, size_t *nblocks);

/* The same in two steps, for a rank that continues another rank's split (banzai_amd/sharded.py): the loop of
 * lib/lib.rs:101-126 carries only `raw`, the stream CRC, `consumed` and the bit cursor from one block to the
 * next, so a rank needs nothing from its predecessor but the offset its first block starts at.
 * bzh_plan_tables_device builds the run tables of d_in[0..n) (no wait; they do not depend on where blocks start);
 * bzh_plan_split_device then cuts blocks from offset `start` of that buffer (a block start) until one starts at or
 * after `stop` (listed last: its offset is what the next rank needs; SIZE_MAX = to the end of the buffer), with or
 * without block CRCs.  Offsets in the resulting plan are relative to d_in. */
BZH_API int bzh_plan_tables_device(bzh_ctx *ctx, const void *d_in, size_t n);
BZH_API int bzh_plan_split_device(bzh_ctx *ctx, size_t start, size_t stop, int with_crc, size_t *nblocks);
BZH_API int bzh_plan_blocks(const bzh_ctx *ctx, bzh_block *out, size_t max_blocks);

/* The same split without the block CRCs (their `crc` fields read 0): for the sharded path, where every
 * rank splits the whole input but only encodes its own blocks.  bzh_encode_range_device computes the
 * CRCs of the range it encodes (they go into the block headers); bzh_plan_crc_range computes them on
 * request; afterwards bzh_plan_blocks reports them.  The rank that assembles the stream needs all of
 * them (bzh_assemble_device's block_crcs), so the ranks exchange them with their bit strings. */
BZH_API int bzh_plan_device_nocrc(bzh_ctx *ctx, const void *d_in, size_t n, size_t *nblocks);
BZH_API int bzh_plan_crc_range(bzh_ctx *ctx, size_t b0, size_t b1);

/* Per block of the plan: 1 if its cut is NOT final unless the planned input is the whole input -- the cut
 * lies in the input's last run, or the block reaches the end of the input (the streaming rule of
 * bzh_stream_feed).  A plan over a PREFIX of a longer input therefore yields the true blocks of the whole
 * input up to the first open one: ranks of the sharded path plan only as far as their own block range. */
BZH_API int bzh_plan_open(const bzh_ctx *ctx, uint8_t *out, size_t max_blocks);

/* Encode blocks [b0, b1) of the plan: per block the header (lib/lib.rs:24-36), symbol map
 * (:39-64) and Huffman payload (lib/huffman.rs:313-575), bit-concatenated from bit 0 of d_out
 * (MSB first, lib/out.rs), zero padded to a 4-byte multiple.  No stream header/footer. */
BZH_API int bzh_encode_range_device(bzh_ctx *ctx, size_t b0, size_t b1, void *d_out, size_t cap, uint64_t *nbits);

/* Stream assembly on one GPU: "BZh"+level (lib/lib.rs:18-22), the nseg bit strings
 * d_segs[k][0..seg_bits[k]) concatenated in order (funnel shift), footer + stream CRC folded
 * over block_crcs in block order (lib/lib.rs:66-70, :108), zero pad to a byte (lib/out.rs:22-28). */
BZH_API int bzh_assemble_device(bzh_ctx *ctx, const void *const *d_segs, const uint64_t *seg_bits, size_t nseg,
                        const uint32_t *block_crcs, size_t nblocks, void *d_out, size_t cap, size_t *out_len);

/* ---- several GPUs behind one handle: banzai::encode for a caller that holds a node (lib/lib.rs:84-88, the loop at
 * :101-126 sharded by start offset as above) -- one host thread and one context per listed device INSIDE the library,
 * the chain hand-off a host variable, the encoded bit strings copied to devices[0] (hipMemcpyPeer: xGMI) and
 * funnel-shifted into the stream there.  A device may be listed more than once (one context per entry: the whole
 * flow runs on a box with one GPU that way).  The handle is single-threaded like a context.  The launcher flow (one
 * process per GPU over torch.distributed, banzai_amd/sharded.py) remains for multi-process jobs. */
typedef struct bzh_multi bzh_multi;
BZH_API int bzh_create_multi(bzh_multi **out, const int *devices, int ndev, int level);
BZH_API void bzh_destroy_multi(bzh_multi *m);
BZH_API const char *bzh_multi_last_error(const bzh_multi *m);
BZH_API int bzh_multi_device_count(const bzh_multi *m);
/* Host buffers in and out: the complete .bz2 stream of in[0..n), bit-identical to bzh_encode's on one device. */
BZH_API int bzh_multi_encode(bzh_multi *m, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len,
                             size_t *consumed);
/* The same in three steps (bench.py --single-process times the middle one: input resident in HBM when it starts,
 * the stream resident on devices[0] when it ends): every worker's byte range (+ look-ahead) to its device; tables,
 * chained split, encode, strings to devices[0], assembly; the stream back to host memory. */
BZH_API int bzh_multi_load(bzh_multi *m, const uint8_t *in, size_t n);
BZH_API int bzh_multi_run(bzh_multi *m, size_t *out_len);
BZH_API int bzh_multi_fetch(bzh_multi *m, uint8_t *out, size_t cap);
BZH_API const void *bzh_multi_output_device(const bzh_multi *m); /* the assembled stream on devices[0] (device pointer) */
/* Wall clocks of the last call per worker, 5 doubles each (ms): load, wait for the chain, tables + split, encode, copy. */
BZH_API int bzh_multi_times(const bzh_multi *m, double *out, size_t max_workers);
/* Test hook: the next runs give every worker a slab of `bytes` instead of the heuristic's (0 = the heuristic): a slab that is
 * too small is answered once with one of twice the size, a second overflow fails the call with BZH_E_CAP. */
BZH_API int bzh_multi_debug_slab(bzh_multi *m, size_t bytes);

/* ---- stage seams (host pointers; computed on the GPU; used by the parity tests) ------------ */

/* rle_one() applied repeatedly (lib/rle.rs:102-253): block table for in[0..n) and, if rle_out
 * is non-NULL, the RLE1 bytes of every block back to back (rle_cap bytes available). */
BZH_API int bzh_rle1_split(bzh_ctx *ctx, const uint8_t *in, size_t n, bzh_block *blocks, size_t max_blocks,
                   size_t *nblocks, uint8_t *rle_out, size_t rle_cap);

/* crc32::checksum (lib/crc32.rs:31-48). */
BZH_API int bzh_crc32(bzh_ctx *ctx, const uint8_t *in, size_t n, uint32_t *crc);

/* bwt::bwt (lib/bwt.rs:526-756) on nblk independent blocks: block k is
 * in[offs[k] .. offs[k]+lens[k]); bwt_out uses the same offsets; ptr[k]; has_byte[k*256..]. */
BZH_API int bzh_bwt_batch(bzh_ctx *ctx, const uint8_t *in, const uint64_t *offs, const uint32_t *lens, size_t nblk,
                  uint8_t *bwt_out, uint32_t *ptr, uint8_t *has_byte);
BZH_API int bzh_bwt(bzh_ctx *ctx, const uint8_t *in, size_t n, uint8_t *bwt_out, uint32_t *ptr, uint8_t *has_byte);

/* Verification tooling.  The reference has no decoder (README.md:9); its fuzz
 * target round-trips through libbz2 (fuzz/fuzz_targets/round_trip.rs:8-22).  bzh_unbwt_batch is the inverse of
 * bzh_bwt_batch computed on the GPU (one radix pass for the LF mapping, then pointer doubling): block k is
 * bwt[offs[k] .. offs[k]+lens[k]) with origin pointer ptr[k]; out receives the original bytes at the same offsets. */
BZH_API int bzh_unbwt_batch(bzh_ctx *ctx, const uint8_t *bwt, const uint64_t *offs, const uint32_t *lens,
                    const uint32_t *ptr, size_t nblk, uint8_t *out);

/* The same check without leaving the device, at any size: for blocks [b0, b1) of the current plan
 * (bzh_plan_device*), RLE1 bytes -> BWT -> inverse BWT, compared with the RLE1 bytes on the GPU.
 * *mismatches = number of differing bytes (0 when the transform is correct). */
BZH_API int bzh_bwt_roundtrip_device(bzh_ctx *ctx, size_t b0, size_t b1, uint64_t *mismatches);

/* mtf::mtf_and_rle (lib/mtf.rs:14-121): syms must hold n+1 entries, freqs 258. */
BZH_API int bzh_mtf(bzh_ctx *ctx, const uint8_t *bwt, size_t n, const uint8_t *has_byte, uint16_t *syms, size_t *m,
            uint32_t *freqs, uint32_t *num_syms);

/* huffman::encode (lib/huffman.rs:313-575) of one block as a standalone bit string from bit 0;
 * code_lengths (optional) receives num_tables x 258 final lengths, *num_tables the count. */
BZH_API int bzh_huffman(bzh_ctx *ctx, const uint16_t *syms, size_t m, uint32_t num_syms, const uint32_t *freqs,
                uint8_t *bits_out, size_t cap, uint64_t *nbits, uint8_t *code_lengths, uint32_t *num_tables);

#ifdef __cplusplus
}
#endif
#endif /* BZHIP_H */
