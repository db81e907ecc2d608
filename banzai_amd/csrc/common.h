// common.h -- shared declarations of libbzhip.so (MI355X / gfx950 only).
// Context, workspace arena, error plumbing and the wavefront-64 scan primitives every stage uses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bzhip.h"

#define BZH_WAVE 64

// ---- error plumbing -----------------------------------------------------------------------
struct bzh_ctx;
void bzh_set_error(bzh_ctx *ctx, const char *fmt, ...);

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            bzh_set_error((ctx), "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return BZH_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

// Waits for a stream.  The suffix-sort rounds no longer wait for the host (bwt.hip), so a batch is left with a
// handful of waits (block table, bit totals, end of the pack): a short poll catches the ones that are about to
// complete, then the thread blocks in hipStreamSynchronize instead of spinning on a core.
static inline hipError_t bzh_stream_wait(hipStream_t st)
{
    for (unsigned it = 0; it < 256u; it++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
    return hipStreamSynchronize(st);
}

#define BZH_TRY(expr)                                                                              \
    do {                                                                                           \
        int s_ = (expr);                                                                           \
        if (s_ != BZH_OK) return s_;                                                               \
    } while (0)

// ---- geometry -----------------------------------------------------------------------------
// Every per-block device array uses one stride S (bytes/elements per bzip2 block), a multiple
// of the sort tile so tiles never straddle blocks.
constexpr int SORT_THREADS = 512;
constexpr int DB_STRIDE = 1280; // digit-base entries per block: up to 5 digits x 256 values
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS; // 8192 elements per workgroup (512 x 16: halves the look-back /
                                                     // scan overhead per element against 256 x 16, -7 % on the bench)
constexpr uint32_t RANK_RESOLVED = 0x80000000u;      // suffix is alone in its group
constexpr uint32_t RANK_EMITTED = 31u << 26;         // (with RANK_RESOLVED, less = 0) its byte of the last column has been written (chunk_finish)
constexpr int SUMMARY_WORDS = 24; // round summary: see round_begin (bwt.hip)
constexpr uint32_t GID_MAX = 4096;  // large groups of a block a round can number densely (12 key bits)
constexpr int RS_ROWS = 22;       // per-block rows of the suffix sort's round state (layout_batch, api.hip)
constexpr uint32_t MS_BG_ROW = 65552, MS_LEVELS = 5, MS_SEG_SLOTS = 112, MS_SEG_ROW = 264, MS_UNIT_CAP = 4096, MS_ITEM_CAP = 224,
                   MS_CNT_WORDS = 48, MS_MIN_N = 131072; // (levels whose blocks stay below MS_MIN_N bytes keep the 8-pass path: no tables for them)
constexpr int MAX_ROUNDS = 30; // depth 8 doubles every round and ends at 2^20; < 31 keeps the rank words' round tags unique

struct BlockDesc { // device-side description of one planned block (mirrors bzh_block + restart info)
    uint64_t in_off;
    uint64_t in_len;
    uint32_t rle_len;
    uint32_t crc;
};

// Device arrays of one batch of B blocks.  Passed to kernels by value.
struct Batch {
    uint32_t B;       // blocks in this batch
    uint32_t S;       // stride (elements) between blocks in per-block arrays
    uint32_t TPB;     // sort tiles per block stride (S / SORT_TILE)
    uint32_t M;       // max RLE1 bytes per block (100000*level-1)
    uint8_t *rle;     // [B][S]  RLE1 output = BWT input
    uint32_t *n;      // [B]     RLE1 length per block
    uint8_t *bwt;     // [B][S]
    uint32_t *ptr;    // [B]
    uint8_t *hasbyte; // [B][256]
    // suffix sorting
    uint32_t *rank; // [B][S]
    uint32_t *sa;   // [B][S]
    uint32_t *headp; // [B][S] group rank by SA position (SWEEP rounds read it instead of gathering)
    uint2 *binned;   // [B][S] (rank word, suffix) pairs of the initial sort, binned by 4096-suffix window (rank_apply); memory of
                     //        its own since round 5: a block on the 8 passes leaves its SA order in sa / headp at the same time
    uint2 *listA;   // [B][S] sort elements (ping-pong of the radix passes; the big-group list between rounds)
    uint2 *listB;   // [B][S]
    uint2 *listC;   // [B][S] small-group (TAIL) list of the block
    uint2 *listD;   // [B][S] TAIL records of the round in flight
    uint32_t *hist; // [B][TPB][512]: 2 KiB per sort tile -- look-back status words (256 x u64) or refine digit rows
    uint32_t *dbase; // [B][DB_STRIDE] digit bases of the look-back passes
    uint32_t *dtot;  // [B][DB_STRIDE] digit totals of an ACTIVE round (5 digits x 256)
    uint8_t *flg;   // [B][S]
    int4 *tagg;     // [B][TPB] tile carries: last group start / last boundary before the tile, first boundary after it
    // Round state of the suffix sort, all [B] unless noted.  The rounds are driven from the device: round_begin
    // turns the counters of the round before into this round's work lists, the host only sizes the launches
    // from a summary it reads one round late.
    uint32_t *st_mode;  // 0: whole block on the radix path, SA-order enumeration (SWEEP); 1: groups routed by size (SPLIT)
    uint32_t *st_h;     // depth of the block's next round
    uint32_t *st_nbig;  // records in the big-group list (SPLIT) / unresolved suffixes (SWEEP)
    uint32_t *st_ntail; // records in the small-group list
    uint32_t *st_tdst;  // which of listC (0) / listD (1) receives the block's small-group records this round: the survivors of
                        // tail_round (which reads the other one) and what refine appends; a block whose small groups sit a
                        // round out keeps its list where it is (round_begin)
    uint32_t *c_big, *c_small, *c_tail, *c_prog; // produced by a round: list lengths, "some group was refined"
    uint32_t *c_nolist; // produced by a round: refine did not write the block's lists (SWEEP mode, mostly large groups)
    uint32_t *c_groups; // groups of the block after the initial sort (refine_one<init>; round_begin picks the first mode)
    uint32_t *scratch;  // a row nobody reads
    uint32_t *chain;    // [B][4] near-periodic blocks: flags, period, leading tails (period_probe, bwt.hip)
    uint32_t *pshrink;  // [B][4] blocks sorted as eight of their periods: flags, period, the block's real length, periods kept (period_detect / period_expand)
    uint32_t *gateS, *gateA, *gateR, *gateT;     // this round: sorted-list length per path (0 = not on that path)
    uint32_t *actS, *actA, *actR, *actT, *actQ;  // this round: ids of the blocks on each path (Q: TAIL at depth x4)
    uint32_t *nlist;    // [8] lengths of those lists (S, A, R, T, Q)
    uint32_t *summary;  // [SUMMARY_WORDS] what the host reads, one round late
    unsigned long long *stat_A; // [1] sum over rounds of the unresolved suffixes entering them
    uint32_t *errflag; // [1]
    // Numbers for the large groups of a round (bwt.hip): whoever writes a large group to a big list (chunk_finish,
    // refine_one, refine) draws a number for it -- one atomic add per GROUP -- and leaves number -> rank and rank -> number;
    // the big lists are then sorted on [number : 12][key2 : 20] in FOUR 8-bit passes instead of on [rank : 20][key2 : 20]
    // in five.  (The order of the groups among each other does not matter: only that a group's records meet.)
    uint16_t *gidof;   // [B][S]  number of the large group whose rank this is (written for the ranks of large groups only)
    uint32_t *grank;   // [2][B][GID_MAX] rank of every numbered group; [round & 1]: a round's refine_one reads one half
                       //         while it fills the other for the next round
    uint32_t *gcount;  // [B]     numbers drawn for the lists being written (round_begin clears it)
    uint32_t *gwide;   // [2]     [round & 1] != 0: some block ran out of numbers for that round: its lists are sorted on ranks
    // bucket-first initial sort (bwt_msd.h): 2-byte buckets, oversized buckets split level by level, every bucket
    // that fits a tile finished inside one workgroup
    uint32_t *ms_bgcur;  // [B][65536] bigram counts, then claim cursors of the partition
    uint32_t *ms_pool;   // bucket starts: [B][MS_BG_ROW] (2-byte buckets), then [MS_LEVELS][B][MS_SEG_SLOTS][MS_SEG_ROW]
    uint32_t *ms_segcur; // [MS_LEVELS][B][MS_SEG_SLOTS][256] digit counts of an oversized bucket, then claim cursors
    uint4 *ms_units;     // [B * MS_UNIT_CAP] work list of the finishing kernel
    uint4 *ms_segs;      // [MS_LEVELS + 1][B * MS_SEG_SLOTS] oversized buckets per level
    uint32_t *ms_items;  // [MS_LEVELS + 1][B * MS_ITEM_CAP] (oversized bucket, tile) pairs per level
    uint32_t *ms_cnt;    // [MS_CNT_WORDS + (MS_LEVELS + 7) * B] counters; behind the first MS_CNT_WORDS per block: units, slot counters
                         // of the levels, unit tickets, "holds a group that spans several units", tickets and tile counts of
                         // mid_sort, records | runs << 20 of the big list being written (two rows, by round parity); then
                         // [3][B][MS_UNIT_CAP] x 2 words: the runs of those lists (two halves) and the tiles mid_plan packs them
                         // into (bwt.hip: msc_* accessors)
    uint32_t *ms_np;     // [B] 1: the block takes the bucket-first path (its first doubling round has depth 7)
    uint32_t *ms_old, *ms_new; // [B] ids of the blocks on the 8-pass path / on the bucket-first path
    uint32_t *ms_bincur; // [B][256] rank binning: pairs already claimed in each 4096-suffix window
    // MTF / RLE2
    uint8_t *mtfpos;   // [B][S]   MTF position of every BWT byte
    uint8_t *tilelist; // [B][MT][256] recency list at each MTF tile entry
    uint32_t *tinfo;   // [B][MT][4] per-tile zero-run bookkeeping
    uint16_t *syms;    // [B][S+64]
    uint32_t *m;       // [B]   symbol count incl. EOB
    uint32_t *freqs;   // [B][258]
    uint32_t *nsyms;   // [B]
    // Huffman
    uint32_t *tfreq;   // [B][3][258]
    uint8_t *lens;     // [B][3][258]
    uint8_t *lens2;    // [2][B][3][258] huff_build: what each half of a block's attempts found (huff_header picks)
    uint32_t *lfit;    // [2][B][3] the scaling exponent that half found to fit (0xFFFFFFFF: none)
    uint32_t *ntab;    // [B]
    uint32_t *codes;   // [B][258]  (len << 24 | word) for table 0
    uint8_t *hdr;      // [B][HDR_BYTES] per-block header bits (block header .. coding tables)
    uint32_t *hdrbits; // [B][4] bits of part A, selector count, bits of part B, payload bits
    uint64_t *bits;    // [B]   total bits of the block
    uint64_t *bitoff;  // [B+1] exclusive scan of bits
    uint32_t *packgate; // [1] pack_gate: 1 = the batch's bits fit the output (the pack kernels of a gated call write nothing otherwise)
    uint32_t *symbits; // [B][PT] per pack tile bit counts
    BlockDesc *desc;   // [B]
    const BlockDesc *pdesc; // [B] where the block CRCs are read from: the plan's descriptors of this batch (rle1_emit; the CRCs may
                            //     arrive there on a side stream while the batch is already being sorted) or `desc` itself (stage seams)
    // "fixed" Huffman mode only (bzh_set_mode) -- the default path never touches these
    uint32_t *fx_tfreq;  // [B][6][258]
    uint8_t *fx_lens;    // [B][6][258]
    uint32_t *fx_codes;  // [B][6][258] (len << 24 | word)
    uint8_t *fx_sel;     // [B][FX_SELMAX] table of every 50-symbol segment
    uint8_t *fx_selbits; // [B][FX_SELBYTES] selectors, MTF + unary coded, as a bit string
    uint8_t *fx_hdr;     // [B][FX_HDR_BYTES] block header .. selector count, then the delta-coded tables
};

constexpr uint32_t MTF_TILE = 2048;  // BWT bytes walked by one wavefront (twice that in batches of 64 blocks and more: mtf_run)
constexpr uint32_t HDR_BYTES = 4160; // 64 B block header/symbol map/counts + up to 3 delta-coded tables (< 25.6 kbit)
constexpr uint32_t PACK_TILE = 4096; // MTF symbols packed by one workgroup
constexpr uint32_t FX_TABLES = 6;         // lib/huffman.rs:319-326 allows 2..6 tables
constexpr uint32_t FX_HDR_A = 64;         // bytes reserved for the part before the selectors
constexpr uint32_t FX_HDR_BYTES = 64 + 6 * 1152; // + up to 6 delta-coded tables (<= 5 + 258 * 35 bits each)

// Kernel classes of the per-kernel roofline table (bzh_get_kernel_stats).  With profiling on, the launches of a
// class are bracketed by HIP events on the context's stream (KSpan); `bytes` = ALGORITHMIC bytes the launches move
// (per-element figures in DESIGN.md, element counts from the plan / the round summaries).
enum KClass : int {
    K_PLAN = 0, K_CRC, K_RLE1_EMIT, K_BYTE_COUNT, K_RADIX_INIT, K_RADIX_GID, K_REFINE_INIT, K_RANK_APPLY,
    K_ROUND_BEGIN, K_SWEEP, K_ACTIVE_GEN, K_RADIX_ROUNDS, K_TAIL_ROUND, K_REFINE_ROUNDS, K_BWT_EMIT, K_MTF_LAST,
    K_MTF_WALK, K_RLE2, K_HUFF, K_PACK, K_MSD_PLAN, K_MSD_SCATTER, K_MSD_LEVELS, K_MSD_FINISH, K_MID_SORT, K_COUNT
};
static const char *const KCLASS_NAME[K_COUNT] = {
    "plan (granules, carries, split)", "crc_tiles", "rle1_emit", "byte_count", "radix_scatter (initial sort)",
    "radix_scatter<GID> (re-key pass)", "refine_one<init> (+ rank binning)", "rank_apply", "round_begin",
    "SWEEP path (3 passes + 3-kernel refine)", "active_gen", "radix_scatter (big-list rounds)", "tail_round",
    "refine_one (rounds)", "bwt_emit", "mtf_tile_last + mtf_prefix", "mtf_walk", "rle2 (tiles, block, emit)",
    "huffman (segments, build, header)", "pack_symbols", "bigram_hist + bigram_plan", "bigram_scatter (2-byte buckets)",
    "seg_count/plan/scatter (oversized buckets)", "chunk_finish (bucket sort + ranks in LDS)",
    "mid_sort (round 0: large groups in LDS)"};

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
};

struct bzh_ctx {
    bzh_ctx *parent = nullptr;        // lanes: the context that owns the plan and the arena
    std::vector<bzh_ctx *> lanes;     // two half-batch workers (own stream, half of the arena each)
    int nlanes = 1;                   // 1: batches run one after the other on this context; 2: on the lanes
    int device = 0;
    int level = 9;
    uint32_t M = 0;
    uint32_t S = 0;
    uint32_t max_batch = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // second stream of the suffix sort (big-list path beside the small groups)
    hipStream_t side2_stream = nullptr;  // third stream: the global passes of the blocks mid_sort does not take, beside it (rounds >= 1)
    hipEvent_t side_ev[3] = {nullptr, nullptr, nullptr};
    // the plan's work beside the main stream (rle1.hip): the block CRCs -- nothing needs them before the block headers are
    // written -- and the table prefetch of the split run on the second stream between these events
    hipEvent_t plan_ev[2] = {nullptr, nullptr};
    bool crc_pending = false;           // the CRCs of the current plan are on their way (rle1_plan_crc_join collects them)
    uint8_t *crc_host = nullptr;        // their landing place: PINNED (a copy to pageable memory holds the host until it is done)
    size_t crc_host_cap = 0, crc_host_len = 0;
    int profiling = 0;
    int mode = 0;                     // BZH_MODE_REFERENCE / BZH_MODE_FIXED (bzh_set_mode)
    char err[512] = {0};      // last failure (guarded by err_mu: the streaming worker writes it too)
    char err_out[512] = {0};  // copy handed out by bzh_last_error
    std::mutex err_mu;
    // arena
    uint8_t *arena = nullptr;
    size_t arena_size = 0;
    uint32_t arena_blocks = 0;        // blocks per batch the arena is laid out for (ensure_arena, api.hip)
    Batch bt{};
    // plan
    const uint8_t *plan_in = nullptr; // device
    size_t plan_n = 0;
    uint32_t wgflag = 0;                // WG_SPREAD for the launches of the current group when few blocks are active
    std::vector<bzh_block> plan_blocks;
    std::vector<uint8_t> plan_open;     // per block: 1 = cut not final unless the input ends here
    std::vector<uint8_t> plan_host;     // host copy of the plan's device records (scratch of rle1_plan)
    std::vector<uint8_t> plan_crc_ok;   // per block: CRC computed (bzh_plan_device_nocrc leaves them to the encoder)
    void *plan_ws = nullptr;            // device scratch of the plan (run tables)
    size_t plan_ws_size = 0;
    // staging
    uint8_t *d_stage_in = nullptr;
    size_t stage_in_size = 0;
    uint8_t *d_stage_out = nullptr;
    size_t stage_out_size = 0;
    uint32_t *h_pinned = nullptr; // small pinned readback area
    void *d_crctab = nullptr;     // GF(2) tables of the block CRC (rle1.hip)
    // streaming encode (bzh_stream_*)
    struct Stream {
        bool active = false, header_done = false;
        // Two device buffers.  d_buf[fill] receives the fed bytes from offset `head` on (copy stream);
        // when a pass starts, the unconsumed tail of the previous pass is placed right before `head`,
        // so the pass sees one contiguous range.  The previous pass's buffer is free again by then.
        uint8_t *d_buf[2] = {nullptr, nullptr};
        size_t cap[2] = {0, 0};
        int fill = 0;
        size_t head = 0;                // offset of the first fed byte in d_buf[fill]
        size_t pending = 0;             // fed bytes waiting in d_buf[fill]
        uint64_t bitpos = 0;            // stream bits handed out or in `carry_word`
        uint32_t carry_word = 0;        // the bitpos % 32 bits not yet handed out (big-endian word)
        uint32_t stream_crc = 0;
        size_t consumed = 0;
        size_t min_feed = (size_t)32 << 20; // pending bytes that trigger a GPU pass
        hipStream_t copy_stream = nullptr;  // H2D of fed bytes, concurrent with the pass in flight
        // the pass in flight (worker thread): plan + encode + D2H of its final blocks
        std::thread worker;
        bool inflight = false;
        struct Pass {
            int buf = 0;                // which d_buf
            int obuf = 0;               // which d_out
            size_t off = 0, total = 0;  // input range of the pass
            bool eof = false;
            uint32_t phase = 0, seed = 0; // bit phase / carried bits at the start of the pass
            // results
            int rc = 0;
            size_t used = 0;            // input bytes consumed by the final blocks
            uint64_t nbits = 0;
            size_t out_bytes = 0;       // whole words copied to h_out
            uint32_t lastw = 0;         // the partial word after them (big-endian value)
            std::vector<uint32_t> crcs; // CRCs of the final blocks, in order
        } pass;
        uint8_t *h_out = nullptr;       // pinned: the partial last word of a pass's output
        size_t h_out_cap = 0;
        // A pass leaves its bits on the device (two buffers, alternating): the next pass is started first, then the
        // finished one's words go straight to the caller's buffer while the GPU is already at work again.
        uint8_t *d_out[2] = {nullptr, nullptr};
        size_t d_out_cap[2] = {0, 0};
        int osel = 0;
    } strm;
    bzh_stats stats{};
    uint32_t debug_fault = 0;         // bzh_debug_fault: fault to inject into the next suffix sort
    bool no_spread = false;           // look-back kernels keep every block on one XCD (set for good after a look-back gave up: bwt_run)
    uint32_t bwt_epoch = 0;           // calls of bwt_run so far (tags the round summaries in pinned memory)
    std::vector<hipEvent_t> evpool;
    size_t evnext = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> sort_spans;
    struct KRec { int cls; hipEvent_t a, b; };
    std::vector<KRec> kspans;          // profiling: event pairs per kernel class of the call in flight
    uint64_t k_cur_ntotal = 0;         // RLE1 bytes of the batch in flight (byte counts of the stages around the sort)
    double k_ms[K_COUNT] = {0};        // collected by kstats_collect
    uint64_t k_bytes[K_COUNT] = {0}, k_launch[K_COUNT] = {0};
};

hipEvent_t bzh_event(bzh_ctx *ctx);
// Brackets the launches issued during its lifetime (one class) with events when profiling is on.
struct KSpan {
    bzh_ctx *c;
    int cls;
    hipEvent_t a = nullptr;
    KSpan(bzh_ctx *ctx, int k, uint64_t bytes, uint64_t launches = 1) : c(ctx), cls(k)
    {
        if (!c->profiling) return;
        a = bzh_event(c);
        hipEventRecord(a, c->stream);
        c->k_bytes[k] += bytes;
        c->k_launch[k] += launches;
    }
    ~KSpan()
    {
        if (!a) return;
        hipEvent_t b = bzh_event(c);
        hipEventRecord(b, c->stream);
        c->kspans.push_back({cls, a, b});
    }
};

// ---- wavefront-64 primitives ----------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ int wave_incl_max(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v = max(v, t);
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Wavefront reductions without address registers: five ds_swizzle steps (the xor pattern is an immediate) and two lane
// reads.  The __shfl_xor forms above cost six lane-dependent addresses, which a kernel that loops over work items keeps alive
// across the whole loop -- in a kernel at its register limit that is six registers in scratch memory.  Result in every lane.
#define BZH_SWZ(v, x) __builtin_amdgcn_ds_swizzle((int)(v), ((x) << 10) | 0x1F)
__device__ __forceinline__ uint32_t wave_all_add(uint32_t v)
{
    v += (uint32_t)BZH_SWZ(v, 1);
    v += (uint32_t)BZH_SWZ(v, 2);
    v += (uint32_t)BZH_SWZ(v, 4);
    v += (uint32_t)BZH_SWZ(v, 8);
    v += (uint32_t)BZH_SWZ(v, 16);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}
__device__ __forceinline__ uint32_t wave_all_max(uint32_t v)
{
    v = max(v, (uint32_t)BZH_SWZ(v, 1));
    v = max(v, (uint32_t)BZH_SWZ(v, 2));
    v = max(v, (uint32_t)BZH_SWZ(v, 4));
    v = max(v, (uint32_t)BZH_SWZ(v, 8));
    v = max(v, (uint32_t)BZH_SWZ(v, 16));
    return max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 32));
}
__device__ __forceinline__ uint32_t wave_all_or(uint32_t v)
{
    v |= (uint32_t)BZH_SWZ(v, 1);
    v |= (uint32_t)BZH_SWZ(v, 2);
    v |= (uint32_t)BZH_SWZ(v, 4);
    v |= (uint32_t)BZH_SWZ(v, 8);
    v |= (uint32_t)BZH_SWZ(v, 16);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}
__device__ __forceinline__ uint32_t wave_all_and(uint32_t v)
{
    v &= (uint32_t)BZH_SWZ(v, 1);
    v &= (uint32_t)BZH_SWZ(v, 2);
    v &= (uint32_t)BZH_SWZ(v, 4);
    v &= (uint32_t)BZH_SWZ(v, 8);
    v &= (uint32_t)BZH_SWZ(v, 16);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) & (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}

// Inclusive OR-scan over the 64 lanes by data-parallel primitives (no lane addresses, no compares): four shifts inside the
// rows of 16 (a lane without a source reads 0), then the last lane of a row to the rows behind it.  OR is idempotent, so
// the plain doubling needs no bank masks.
#define BZH_DPP_OR(v, ctrl, rowmask) ((v) | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rowmask), 0xF, true))
__device__ __forceinline__ uint32_t wave_incl_or(uint32_t v)
{
    v = BZH_DPP_OR(v, 0x111, 0xF); // row_shr:1
    v = BZH_DPP_OR(v, 0x112, 0xF); // row_shr:2
    v = BZH_DPP_OR(v, 0x114, 0xF); // row_shr:4
    v = BZH_DPP_OR(v, 0x118, 0xF); // row_shr:8
    v = BZH_DPP_OR(v, 0x142, 0xA); // row_bcast:15 -> rows 1 and 3
    v = BZH_DPP_OR(v, 0x143, 0xC); // row_bcast:31 -> rows 2 and 3
    return v;
}
// the value of the lane below (0 into lane 0)
__device__ __forceinline__ uint32_t wave_from_below(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); // wave_shr:1
}

// Workgroup exclusive add-scan of one value per thread.  `lds` needs (threads/64)+1 words.
// Returns the exclusive prefix; *total receives the workgroup sum.
// (`tid`: the thread's index as the caller holds it -- a kernel that loops over work items and has made its index opaque
// per item passes that one, so that nothing here is hoisted out of its loop and kept alive across it)
__device__ __forceinline__ uint32_t block_excl_add_at(uint32_t v, uint32_t *lds, uint32_t *total, uint32_t tid)
{
    const int lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc = wave_incl_add(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < nw ? lds[lane] : 0;
        uint32_t wi = wave_incl_add(w, lane);
        if (lane < nw) lds[lane] = wi - w;
        if (lane == nw - 1) lds[nw] = wi;
    }
    __syncthreads();
    uint32_t res = inc - v + lds[wave];
    *total = lds[nw];
    __syncthreads();
    return res;
}
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t *lds, uint32_t *total)
{
    return block_excl_add_at(v, lds, total, threadIdx.x);
}

// Workgroup inclusive max-scan of one int per thread. `lds` needs (threads/64) ints.
__device__ __forceinline__ int block_incl_max(int v, int *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = wave_incl_max(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int carry = INT32_MIN;
    for (int w = 0; w < wave; w++) carry = max(carry, lds[w]);
    (void)nw;
    int res = max(inc, carry);
    __syncthreads();
    return res;
}

// Two inclusive max-scans at once (same barriers).  `lds` needs 2*(threads/64) ints.
__device__ __forceinline__ void block_incl_max2(int &a, int &b, int *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const int ia = wave_incl_max(a, lane), ib = wave_incl_max(b, lane);
    if (lane == 63) {
        lds[wave] = ia;
        lds[nw + wave] = ib;
    }
    __syncthreads();
    int ca = INT32_MIN, cb = INT32_MIN;
    for (int w = 0; w < wave; w++) {
        ca = max(ca, lds[w]);
        cb = max(cb, lds[nw + w]);
    }
    a = max(ia, ca);
    b = max(ib, cb);
    __syncthreads();
}

// Workgroup EXCLUSIVE suffix-min: result = min of v over all threads with a higher index (INT32_MAX for the
// last thread).  `lds` needs (threads/64) ints.
__device__ __forceinline__ int block_excl_min_rev(int v, int *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v; // inclusive suffix-min inside the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_down(inc, d, 64);
        if (lane + d < 64) inc = min(inc, t);
    }
    if (lane == 0) lds[wave] = inc;
    __syncthreads();
    int carry = INT32_MAX;
    for (int w = wave + 1; w < nw; w++) carry = min(carry, lds[w]);
    int ex = __shfl_down(inc, 1, 64);
    if (lane == 63) ex = INT32_MAX;
    const int res = min(ex, carry);
    __syncthreads();
    return res;
}

// ---- stage entry points (host side, defined in the stage files) ---------------------------------
int bwt_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal, bool is_retry = false); // bwt.hip
int unbwt_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax);               // bwt.hip: inverse transform, bt.bwt/ptr -> bt.mtfpos
int unbwt_compare(bzh_ctx *ctx, uint32_t B, uint32_t nmax, unsigned long long *d_acc); // bwt.hip: bt.rle vs bt.mtfpos
int mtf_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal = 0); // mtf.hip (ntotal: statistics only)
int huff_prepare(bzh_ctx *ctx, uint32_t B, uint32_t mmax);            // huffman.hip: tables, header bits, bit totals
int huff_pack(bzh_ctx *ctx, uint32_t B, uint32_t mmax, uint8_t *d_out, uint64_t bit_base, bool gated = false); // huffman.hip
// (one batch, no host in between: zeroes the output words the batch's bits will occupy -- the first one may carry bits owed to
// it --, checks the capacity on the device and opens or shuts the gate of the pack kernels; hostrec[0] = bits, [1] = fits)
int huff_pack_gate(bzh_ctx *ctx, uint32_t B, uint8_t *d_out, uint64_t bit_base, uint64_t cap_words, uint32_t seed, bool has_seed,
                   uint64_t *hostrec, uint32_t tail_bits = 0);
int huff_frame_stream(bzh_ctx *ctx, uint32_t B, uint8_t *d_out); // huffman.hip: stream header + footer of a one-batch stream, on the device
int rle1_plan(bzh_ctx *ctx, const uint8_t *d_in, size_t n, bool with_crc = true, bool crc_async = false); // rle1.hip: tables + split from 0
int rle1_plan_tables(bzh_ctx *ctx, const uint8_t *d_in, size_t n);                 // rle1.hip
int rle1_plan_split(bzh_ctx *ctx, size_t start, bool with_crc, size_t stop, bool crc_async = false); // rle1.hip
int rle1_plan_crc_join(bzh_ctx *ctx);                                             // rle1.hip: CRCs queued on the side stream -> plan_blocks
hipStream_t bzh_side_stream(bzh_ctx *ctx);                                        // api.hip: the context's second stream (created once; null: none to be had)
int rle1_plan_crc(bzh_ctx *ctx, size_t b0, size_t b1);               // rle1.hip: CRCs of plan blocks [b0, b1)
int rle1_emit(bzh_ctx *ctx, size_t b0, uint32_t B);                   // rle1.hip: fill bt.rle / bt.n / bt.desc
int crc_device(bzh_ctx *ctx, const uint8_t *d_in, size_t n, uint32_t *crc_out); // rle1.hip

hipEvent_t bzh_event(bzh_ctx *ctx);
