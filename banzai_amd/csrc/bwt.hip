// bwt.hip -- batched wrap-around Burrows-Wheeler transform on gfx950.
//
// Replaces bwt::bwt (reference lib/bwt.rs:526-756: SA-IS on the doubled block).  Output
// contract: bwt[k] = S[(sa_k - 1) mod n] for the rotations of S in order, ties
// between identical rotations broken by DESCENDING start index (lib/bwt.rs:564-573 sorts S||S);
// ptr = k with sa_k = 0; has_byte[c] = c in S.
//
// Method: cyclic prefix doubling, all bzip2 blocks of a batch at once.  Every sort element is one
// 64-bit word
//      init pass :  [ 4 bytes of the cyclic prefix : 32 ][ 0 : 12 ][ suffix : 20 ]
//      rounds    :  [ 0:4 ][ group rank r : 20 ][ key2 = rank[i+h] : 20 ][ suffix i : 20 ]
// (n < 2^20 at every level).  rank = first SA position of the suffix's group, bit 31 = resolved.
// Initial LSD radix sort on the 8-byte prefix (8 passes x 8 bits, key half swapped after pass 4),
// then doubling rounds from depth h = 8.  A block is in one of two modes:
//   SWEEP mode (more than a third of its suffixes sit in large groups: periodic / run-heavy blocks):
//          all unresolved suffixes are ENUMERATED in SA order of suffix i+h (a coalesced sweep of SA +
//          rank gathers) and stably sorted by their group rank only -- 3 passes of 7 bits; stability
//          leaves every group in key2 order;
//   SPLIT mode (everything else, for good once entered): refine routes every group by its size.
//          Small groups (<= TAIL_G members) live in the block's TAIL list and are ranked locally in LDS
//          (tail_sort / tail_finish); large groups live in the big list, which is re-keyed once
//          (active_gen) and sorted on (r, key2) in 5 passes of 8 bits.  Groups that shrink move over.
// After the radix passes: boundary flags, scans of the group heads in both directions (a group's size
// is the distance between boundaries), rank/SA update, routing.  When h >= n the survivors are
// identical rotations (block = w^k): key2 becomes n-1-i, the reference's tie rule.  A block
// whose round refined no group at all is exactly periodic (equal ranks at depth h imply equal ranks at
// every depth), so its depth jumps straight to "h >= n".
//
// The rounds are driven from the DEVICE.  round_begin (one workgroup) turns the counters the previous
// round left into this round's per-block depths, modes and work lists; every kernel of a round takes
// its blocks from those lists and exits when there is nothing for it.  The host never waits for a round:
// it reads a 16-word summary ONE ROUND LATE -- enough to size the next launches (list lengths only
// shrink, up to the bounds used below) and to notice the end.
//
// Every radix pass is ONE kernel (radix_scatter): a tile publishes its digit counts and finds its
// first slots by decoupled look-back over the earlier tiles of its block; the digit totals a pass
// needs up front are a by-product of the step before it (byte_count / refine + sweep_bases /
// active_gen + active_bases), so nothing is ever read just to be counted.
//
// Launch geometry: workgroup ids are mapped so that all tiles of a block run on one XCD (wg_map),
// keeping the block's rank/SA arrays (3.6 MB each) inside one 4 MiB L2.
// Kernels (integer only, HBM/LDS bound, no MFMA):
//   byte_count     digit totals of the 8 initial passes (= byte counts of the cyclic block)
//   radix_scatter  stable single-pass scatter: wave match-any ranking, per-wave LDS cursors,
//                  look-back for the tile's global offsets, elements reordered in LDS so each
//                  digit's run leaves the CU as coalesced stores
//   flag_tiles / flag_carry / refine   boundary flags, group extents, rank + SA update, routing
//   sweep_bases / active_gen / active_bases   digit bases of the SWEEP / big-list passes
//   tail_sort / tail_finish   small groups
//   round_begin    per-round bookkeeping on the device
//   bwt_emit       last column, ptr, has_byte
#include <type_traits>
#include <vector>

#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned long long u64;

enum GenMode : int {
    GEN_BYTES5 = 0, // element e is suffix e keyed by bytes 3..7 of its rotation
    GEN_GID = 2,    // element e is src[e], re-keyed: bytes 0..2 of its rotation + dense index of its 5-byte group
    GEN_SWEEP = 1,  // doubling round, SA-order enumeration
    GEN_LIST = 3,   // element e is src[e]
    GEN_LCOL = 4    // inverse BWT: element e is position e of the last column, keyed by its byte
};

// Timing experiments that leave parts of a kernel out (wrong results on purpose) exist in builds made with
// -DBZH_EXPERIMENTS only; the product build has neither the branches nor the environment variables.
#ifdef BZH_EXPERIMENTS
#define BZH_DBG(x) (x)
#else
#define BZH_DBG(x) 0u
#endif

constexpr u64 SUF_MASK = 0xFFFFFull;
constexpr uint32_t H_DONE = 1u << 30; // depth that stands for "h >= n"

// rank word: [resolved:1][tag:5][less:6][rank:20].  tag = id of the round that wrote the word (1..31; 0: written
// by refine, which no reader of the same launch ever races with).  A reader of round `tag` sees the rank as it
// stood when that round began: `rank` if the word was written in this round, `rank + less` otherwise.
__device__ __forceinline__ uint32_t rank_word(uint32_t r, uint32_t less, uint32_t tag, bool resolved)
{
    return (resolved ? RANK_RESOLVED : 0u) | (tag << 26) | (less << 20) | r;
}
__device__ __forceinline__ uint32_t rank_at(uint32_t w, uint32_t tag)
{
    const uint32_t r = w & 0xFFFFFu, less = (w >> 20) & 63u;
    return ((w >> 26) & 31u) == tag ? r : r + less;
}
__device__ __forceinline__ uint32_t rank_final(uint32_t w) { return (w & 0xFFFFFu) + ((w >> 20) & 63u); }

// The blocks a launch works on: ids[0 .. *cnt) (both on the device), or all of 0 .. B-1 when ids is null.
struct Lst {
    const uint32_t *ids;
    const uint32_t *cnt;
    uint32_t B;
};

// ---- the large groups of a bucket-first block, round by round (bwt_msd.h: mid_plan / mid_sort) --------------------------
// Rows of per-block words behind the counters of the initial sort (Batch::ms_cnt) and, behind the rows, three tables of
// MS_UNIT_CAP (first record, records) pairs per block: the RUNS of the big list a round's refinement (round 0: chunk_finish)
// writes -- every writer claims its run's room with one atomic add on records | runs << 20, which also numbers the runs in
// list order --, two halves by the parity of the round that READS the list, and the TILES mid_plan packs a list's runs into.
constexpr uint32_t MSR_SPANS = MS_LEVELS + 2, MSR_MTICKET = MS_LEVELS + 3, MSR_MTILES = MS_LEVELS + 4, MSR_RUNQ = MS_LEVELS + 5; // (MSR_RUNQ: two rows)
__host__ __device__ __forceinline__ uint32_t *msc_row(uint32_t *cnt, uint32_t B, uint32_t row) { return cnt + MS_CNT_WORDS + (size_t)row * B; }
__host__ __device__ __forceinline__ uint2 *msc_runs(uint32_t *cnt, uint32_t B, uint32_t par)
{
    return reinterpret_cast<uint2 *>(cnt + ((MS_CNT_WORDS + (size_t)(MS_LEVELS + 7) * B + 1) & ~(size_t)1)) + (size_t)par * B * MS_UNIT_CAP;
}
__host__ __device__ __forceinline__ uint2 *msc_tiles(uint32_t *cnt, uint32_t B) { return msc_runs(cnt, B, 2); }

struct SortArgs {
    const uint8_t *blk;   // [B][S]
    const uint32_t *n;    // [B]
    const uint32_t *cnt;  // [B] elements enumerated this pass
    const uint32_t *rank; // [B][S]
    const uint32_t *sa;   // [B][S]
    const uint32_t *headp; // [B][S] group rank of the suffix at each SA position
    const uint32_t *hb;   // [B] depth of the block's round (GEN_SWEEP, active_gen)
    const u64 *src;       // [B][S]
    u64 *dst;             // [B][S]
    uint32_t S, TPB, h, shift;
    uint32_t T;           // tiles per block in this launch (| WG_SPREAD)
    Lst lst;
    // single-pass (look-back) scatter:
    u64 *look;             // [B][TPB][256] tile status words  [pass:32][state:2][count:30]
    const uint32_t *dbase; // [B][DB_STRIDE] first slot of every digit (exclusive scan of the pass's digit totals)
    uint32_t doff;         // which 128/256-entry group of dbase this pass uses
    uint32_t *err;         // bit 1: a look-back gave up (internal error, never a hang)
    uint32_t pass;         // id of this pass in the status words (stale words read as "not there yet")
    uint32_t tag;          // id of the round in the rank words (rank_at)
    u64 *gst;              // [B][TPB][2] tile status words of GEN_GID's look-back
    const uint32_t *chain; // [B][4] near-periodic blocks (period_probe): flags, period, tails that lead the order
    const uint32_t *clist; // [B][2S] those tails, then the other tails, ascending (the block's listD)
    uint32_t fault;        // test hook (bzh_debug_fault): 1 = tile 1 of block 0 never publishes its digit counts
    uint32_t patient;      // 1: the pinned retry after a look-back gave up -- waits are bounded by LOOK_TICKS_PATIENT
    // big-list rounds with numbered groups: *gwide == 0 -> the keys are 32 bits, four passes, and the passes run over the
    // buffers named here instead (active_gen wrote in place; a null src_n: this pass is not needed)
    const uint32_t *gwide; // the word of THIS round (bt.gwide + (round & 1))
    const u64 *src_n;
    u64 *dst_n;
    uint32_t has_n;        // 1: the alternates above are meaningful for this launch
    uint32_t only4;        // the host launches four passes only (no list of the round can run out of numbers): a raised *gwide is an error
    const uint16_t *gidof; // [B][S] (active_gen)
    // round 0 of a batch on the bucket-first initial sort: the blocks whose large groups mid_sort orders in LDS (bwt_msd.h) are
    // not this path's -- mid_np[b] != 0 (the block took that sort) and mid_spans[b] == 0 (no group spans several units); null:
    // every listed block is
    const uint32_t *mid_np, *mid_spans;
};
__device__ __forceinline__ bool ms_block_is_mid(const uint32_t *np, const uint32_t *spans, uint32_t b) { return np[b] != 0u && spans[b] == 0u; }

constexpr int NBMAX = 256;
constexpr u64 LIST_INVALID = ~0ull; // list slot without an unresolved suffix

// XCD-aware workgroup -> (bzip2 block, tile) map.  Workgroups are dealt round-robin over the 8
// XCDs (observed dispatch behaviour, used for speed only): ids congruent mod 8 share an XCD and
// its private 4 MiB L2.  All tiles of the k-th listed block get ids = k (mod 8).  Grid = 8*ceil(NB/8)*T.
// With very few blocks that would leave XCDs idle (a single block would run on 32 of the 256 CUs): launches over
// fewer than 6 blocks (few_blocks below) set WG_SPREAD in T and get the plain
// mapping, consecutive workgroup ids = consecutive tiles of one block, i.e. every block on all XCDs.
// Either way tile t-1 of a block has a lower workgroup id than tile t (the look-backs rely on it).
constexpr uint32_t WG_SPREAD = 0x80000000u;
__device__ __forceinline__ bool wg_map(uint32_t T, const Lst &l, uint32_t &b, uint32_t &tile)
{
    const uint32_t L = blockIdx.x;
    const uint32_t nb = l.ids ? *l.cnt : l.B;
    uint32_t k;
    if (T & WG_SPREAD) {
        T &= ~WG_SPREAD;
        k = L / T;
        tile = L - k * T;
    } else {
        const uint32_t slot = L >> 3;
        const uint32_t kk = slot / T;
        tile = slot - kk * T;
        k = kk * 8u + (L & 7u);
    }
    if (k >= nb) return false;
    b = l.ids ? l.ids[k] : k;
    return true;
}

// Which launches spread: those over fewer than 6 blocks.  Pinned, a block's rank array, lists and output stay in ONE
// XCD's L2 (the random 4-byte gathers and stores of the rounds, the one-byte scatter of bwt_emit: 177 -> 78 us for 16
// blocks), which outweighs idle XCDs from 6 blocks on -- whole encodes of 6 / 8 / 16 / 24 / 31 text blocks 2.21 / 2.42 /
// 3.09 / 3.73 / 4.38 ms spread against 2.14 / 2.27 / 2.87 / 3.38 / 3.94 ms pinned, 5 blocks the same, 4 blocks 1.73
// against 1.82 ms.  (BZH_SPREAD_MAX moves the limit: A/B timing.)
static inline bool few_blocks(uint32_t NB)
{
    static const uint32_t lim = getenv("BZH_SPREAD_MAX") ? (uint32_t)atoi(getenv("BZH_SPREAD_MAX")) : 6u;
    return NB < lim;
}

static inline uint32_t xcd_grid(uint32_t tiles, uint32_t NB)
{
    return (tiles & WG_SPREAD) ? (tiles & ~WG_SPREAD) * NB : 8u * ((NB + 7u) / 8u) * tiles;
}

// The host sizes every launch of a round from a summary that is one round old (bounds, see bwt_run).  A launch that
// turned out too small would silently skip blocks or tiles; this makes it loud instead (bit 2 of the error word):
// workgroup 0 compares the list length with the blocks the grid covers, every block's first tile compares the tiles
// it needs with the tiles launched.
__device__ __forceinline__ void launch_check(uint32_t T, const Lst &l, uint32_t tile, uint32_t need_tiles, uint32_t *err)
{
    if (threadIdx.x != 0) return;
    const uint32_t tl = T & ~WG_SPREAD;
    if (blockIdx.x == 0 && (l.ids ? *l.cnt : l.B) > gridDim.x / tl) atomicOr(err, 4u);
    if (tile == 0 && need_tiles > tl) atomicOr(err, 4u);
}

// Where the rank of suffix i lives inside the block's rank array.  Periodic blocks visit suffixes at a
// power-of-two stride (a 1024-byte tile repeated: members of a group sit 1024 suffixes = 4 KiB apart), which
// would keep hitting the same few cache sets; folding bits 10..14 into bits 5..9 spreads such walks over 32
// times as many lines.  Bits 0..4 are untouched: 32 consecutive suffixes still share one 128-byte line.
// A bijection on every aligned 1024-block, so slots stay below the array stride.
__device__ __forceinline__ uint32_t rslot(uint32_t i) { return i ^ (((i >> 10) & 31u) << 5); }

// 4 bytes of the cyclic text starting at position i (big-endian), i < n.
__device__ __forceinline__ uint32_t text4(const uint8_t *s, uint32_t i, uint32_t n)
{
    if (i + 3 < n) { // one (possibly unaligned) dword load; gfx950 global loads need no alignment
        uint32_t w;
        __builtin_memcpy(&w, s + i, 4);
        return __builtin_bswap32(w);
    }
    // cyclic wrap (n may be smaller than 4)
    const uint32_t i1 = (i + 1) % n, i2 = (i + 2) % n, i3 = (i + 3) % n;
    return ((uint32_t)s[i] << 24) | ((uint32_t)s[i1] << 16) | ((uint32_t)s[i2] << 8) | (uint32_t)s[i3];
}

// (suffix of x + 4) mod n
__device__ __forceinline__ uint32_t wrap_add(u64 x, uint32_t n)
{
    uint32_t i = (uint32_t)(x & SUF_MASK) + 4u;
    if (i >= n) i = n > 4 ? i - n : i % n;
    return i;
}

template <int MODE>
__device__ __forceinline__ bool gen_elem(const SortArgs &a, uint32_t b, uint32_t e, uint32_t n, uint32_t h, u64 &v)
{
    const size_t base = (size_t)b * a.S;
    if (MODE == GEN_BYTES5) {
        // key = bytes 3..7 of the rotation, the most significant first, in bits 20..59
        const uint32_t a4 = (e + 3u) & ~3u;
        if (a4 + 8u <= n) { // one aligned 8-byte window holds all five bytes (the lanes of a wavefront share 17 dwords)
            uint2 w;
            __builtin_memcpy(&w, a.blk + base + a4, 8);
            const u64 x = (((u64)w.y << 32) | w.x) >> (8u * ((e + 3u) & 3u)); // byte 3 of the rotation lowest
            v = ((u64)(__builtin_bswap64(x << 24) & 0xFFFFFFFFFFull) << 20) | e;
            return true;
        }
        uint32_t i = e + 3u;
        if (i >= n) i = n > 3 ? i - n : i % n;
        uint32_t i7 = i + 4u;
        if (i7 >= n) i7 = n > 4 ? i7 - n : i7 % n;
        const uint32_t w4 = text4(a.blk + base, i, n);
        v = ((u64)w4 << 28) | ((u64)a.blk[base + i7] << 20) | e;
        return true;
    } else if (MODE == GEN_SWEEP) {
        uint32_t i, k2;
        if (h < n) { // suffix j = sa[e] is the e-th smallest; i = j - h has it as its second half
            const uint32_t j = a.sa[base + e];
            k2 = a.headp[base + e]; // = rank[j] without the gather (loaded next to sa[e], not behind the rank gather)
            i = j >= h ? j - h : j + n - h;
            const uint32_t r = a.rank[base + rslot(i)]; // (a block in SWEEP mode only ever holds refine's plain words)
            if (r & RANK_RESOLVED) return false;
            v = ((u64)rank_at(r, a.tag) << 40) | ((u64)k2 << 20) | i;
            return true;
        }
        i = n - 1 - e; // identical rotations: larger index first; e doubles as a distinct key2
        const uint32_t cf = a.chain[b * 4];
        if (cf & 1u) { // near-periodic block: every group is a chain ordered by (effective) index, see period_probe
            const uint32_t p = a.chain[b * 4 + 1], c2 = a.chain[b * 4 + 2];
            const uint32_t q = (cf & 2u) ? e : n - 1 - e; // ascending / descending
            const uint32_t *cl = a.clist + (size_t)b * a.S * 2;
            const uint32_t inner = n - p + 1; // indices 0 .. n-p are no tails
            i = q < c2 ? cl[q] : (q - c2 < inner ? q - c2 : cl[p + (q - c2 - inner)]);
        }
        const uint32_t r = a.rank[base + rslot(i)];
        if (r & RANK_RESOLVED) return false;
        v = ((u64)rank_at(r, a.tag) << 40) | ((u64)e << 20) | i;
        return true;
    } else if (MODE == GEN_LCOL) {
        v = ((u64)a.blk[base + e] << 32) | e;
        return true;
    } else {
        v = a.src[base + e];
        return true;
    }
}

// Digit totals of the initial sort.  Every one of its 8 passes keys on one byte of the CYCLIC
// rotation, so each pass's digit histogram is the block's byte histogram: BYTE_SEGS workgroups per
// block count a segment each (per-wave private counters) and add it to dtot[b][0..255];
// active_bases(…, 1) then leaves the exclusive scan in dbase.
constexpr int BYTE_SEGS = 8;
__global__ void __launch_bounds__(1024) byte_count(const uint8_t *blk, const uint32_t *nn, uint32_t *dtot, uint32_t S)
{
    const uint32_t b = blockIdx.y, n = nn[b];
    const uint8_t *s = blk + (size_t)b * S;
    const uint32_t per = ((n + BYTE_SEGS - 1) / BYTE_SEGS + 15u) & ~15u; // keeps the 16-byte loads aligned
    const uint32_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    if (lo >= hi) return;
    __shared__ uint32_t h[16][256];
    for (int k = threadIdx.x; k < 16 * 256; k += 1024) (&h[0][0])[k] = 0;
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
    // 16 bytes a step; equal neighbours (RLE1 leaves runs of up to four, text has its doubled letters) share one add
    for (uint32_t i = lo + threadIdx.x * 16; i < hi; i += 16384) {
        if (i + 16 <= hi) {
            const uint4 q = *reinterpret_cast<const uint4 *>(s + i); // (lo and the block base are 16-byte aligned)
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t cur = w[0] & 255u, cnt = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t c = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
                if (c != cur) {
                    atomicAdd(&mine[cur], cnt);
                    cur = c;
                    cnt = 0;
                }
                cnt++;
            }
            atomicAdd(&mine[cur], cnt);
        } else {
            for (uint32_t j = i; j < hi; j++) atomicAdd(&mine[s[j]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) c += h[w][threadIdx.x];
        if (c) atomicAdd(&dtot[(size_t)b * DB_STRIDE + threadIdx.x], c);
    }
}

// Big-list round, step 1: re-key the block's big list ONCE (one gather of rank[i+h] per listed suffix;
// the suffix's own group rank is in the record) into dst, same slot, and count all five 8-bit digits of
// the new keys (bits 20..59) into the block's totals, from which active_bases makes the bases of the
// five look-back passes that follow (which also clears the totals again).
__global__ void __launch_bounds__(SORT_THREADS) active_gen(SortArgs a, uint32_t *dtot)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.lst, b, tile)) return;
    if (a.mid_np && ms_block_is_mid(a.mid_np, a.mid_spans, b)) return; // (mid_sort's)
    const uint32_t cnt = a.cnt[b], n = a.n[b], h = a.hb[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    launch_check(a.T, a.lst, tile, ntile, a.err);
    if (tile >= ntile) return;
    __shared__ uint32_t hh[5 * 256];
    for (int k = threadIdx.x; k < 5 * 256; k += SORT_THREADS) hh[k] = 0;
    __syncthreads();
    const size_t base = (size_t)b * a.S;
    // Numbered groups: the record leaves with the group's NUMBER in place of its rank -- 12 bits, so the key is 32 bits and
    // four passes sort it -- and in place: the four passes then end in the buffer five would end in.
    const bool narrow = *a.gwide == 0u;
    if (!narrow && a.only4 && threadIdx.x == 0) atomicOr(a.err, 4u); // (cannot happen: the host's bound on the lists rules it out)
    const u64 *src = a.src + base;
    u64 *dst = (narrow ? const_cast<u64 *>(a.src) : a.dst) + base;
    const uint32_t *rank = a.rank + base;
    const uint16_t *gidof = a.gidof + base;
    const int lane = threadIdx.x & 63;
#pragma unroll 4
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t e = tile * SORT_TILE + k * SORT_THREADS + threadIdx.x;
        const bool ok = e < cnt;
        u64 v = 0;
        if (ok) {
            const u64 x = src[e];
            const uint32_t i = (uint32_t)(x & SUF_MASK);
            uint32_t r = (uint32_t)(x >> 40) & 0xFFFFFu;
            uint32_t k2;
            if (h < n) {
                uint32_t i2 = i + h;
                if (i2 >= n) i2 -= n;
                k2 = rank_at(rank[rslot(i2)], a.tag);
            } else {
                k2 = n - 1 - i;
            }
            if (narrow) r = gidof[r]; // (a group's members read the same word)
            v = ((u64)r << 40) | ((u64)k2 << 20) | i;
            dst[e] = v;
            atomicAdd(&hh[(uint32_t)(v >> 20) & 255u], 1u);
            atomicAdd(&hh[256 + ((uint32_t)(v >> 28) & 255u)], 1u);
            atomicAdd(&hh[512 + ((uint32_t)(v >> 36) & 255u)], 1u);
        }
        // the list is ordered by rank, so a wavefront's 64 suffixes nearly always share the two upper
        // rank digits: one LDS add per distinct value instead of 64 adds onto the same counter
        const uint32_t up = (uint32_t)(v >> 44) & 0xFFFFu;
        u64 todo = __ballot(ok);
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const uint32_t u0 = (uint32_t)__builtin_amdgcn_readlane((int)up, first);
            const u64 same = __ballot(ok && up == u0) & todo;
            if (lane == first) {
                const uint32_t c = (uint32_t)__popcll(same);
                atomicAdd(&hh[768 + (u0 & 255u)], c);
                atomicAdd(&hh[1024 + (u0 >> 8)], c);
            }
            todo &= ~same;
        }
    }
    __syncthreads();
    uint32_t *tot = dtot + (size_t)b * DB_STRIDE;
    for (int k = threadIdx.x; k < 5 * 256; k += SORT_THREADS)
        if (hh[k]) atomicAdd(&tot[k], hh[k]);
}

// ---- numbers for the large groups -----------------------------------------------------------------------------------------
// Called once per large group, by the thread that writes the group's first record: draws the group's number for the round
// the list is written for (`par` = that round & 1) and records number <-> rank; a block that runs out of numbers makes
// that round sort on ranks.
struct GidOut {
    uint16_t *gidof; // [B][S]
    uint32_t *grank; // [2][B][GID_MAX]
    uint32_t *gcount; // [B]
    uint32_t *gwide;  // [2]
    uint32_t S, B, par;
};
// (the same for a workgroup's worth of groups whose ranks stand in a list: numbers base .. base + count - 1, one of them a thread)
__device__ __forceinline__ void gid_assign(const GidOut &g, uint32_t b, uint32_t k, uint32_t rank)
{
    if (k < GID_MAX) {
        g.grank[((size_t)g.par * g.B + b) * GID_MAX + k] = rank;
        g.gidof[(size_t)b * g.S + rank] = (uint16_t)k;
    } else {
        atomicOr(&g.gwide[g.par], 1u);
    }
}
__device__ __forceinline__ void gid_draw(const GidOut &g, uint32_t b, uint32_t rank)
{
    if (!g.gcount) return;
    const uint32_t k = atomicAdd(&g.gcount[b], 1u);
    if (k < GID_MAX) {
        g.grank[((size_t)g.par * g.B + b) * GID_MAX + k] = rank;
        g.gidof[(size_t)b * g.S + rank] = (uint16_t)k;
    } else {
        atomicOr(&g.gwide[g.par], 1u);
    }
}

// One workgroup per listed block: exclusive scan inside each of the ndig digit groups of dtot.
__global__ void __launch_bounds__(256) active_bases(uint32_t *dtot, uint32_t *dbase, Lst lst, int ndig)
{
    const uint32_t k = blockIdx.x;
    if (k >= (lst.ids ? *lst.cnt : lst.B)) return;
    const uint32_t b = lst.ids ? lst.ids[k] : k;
    __shared__ uint32_t ls[8];
#pragma unroll 1
    for (int p = 0; p < ndig; p++) {
        const uint32_t v = dtot[(size_t)b * DB_STRIDE + p * 256 + threadIdx.x];
        dtot[(size_t)b * DB_STRIDE + p * 256 + threadIdx.x] = 0; // consumed: clean for the next round's counting
        uint32_t tot;
        const uint32_t ex = block_excl_add(v, ls, &tot);
        dbase[(size_t)b * DB_STRIDE + p * 256 + threadIdx.x] = ex;
    }
}

// A look-back waits for a predecessor that is resident or about to be: no legitimate wait is longer than one kernel
// (about a millisecond).  The wait is bounded by TIME, not by a spin count (2^26 sleeps were 16 seconds): the constant
// 100 MHz clock (s_memrealtime) is read every 64th idle spin; LOOK_TICKS = 20 ms, then the caller raises bit 1 of the
// error word and bwt_run starts over with every block pinned to one XCD.
constexpr unsigned long long LOOK_TICKS = 2000000ull, LOOK_TICKS_FAULT = 100000ull; // (injected faults: 1 ms)
// The clock keeps running while the queue is preempted (another process time-slicing the GPU, a profiler serialising
// kernels), so a waiter and its also-preempted predecessor can both expire with no logic error.  In the pinned retry a
// tile only ever waits for a workgroup that is resident already -- no deadlock is possible there -- so the retry waits
// seconds, not milliseconds: a correct encode does not fail because of scheduling (SortArgs / RefineArgs::patient).
constexpr unsigned long long LOOK_TICKS_PATIENT = 3000000000ull; // 30 s
struct LookWait {
    unsigned long long t0 = 0;
    uint32_t spins = 0;
    __device__ __forceinline__ bool expired(unsigned long long budget = LOOK_TICKS)
    {
        if ((++spins & 63u) != 0u) return false;
        const unsigned long long now = wall_clock64();
        if (t0 == 0) {
            t0 = now | 1ull;
            return false;
        }
        return now - t0 > budget;
    }
};

constexpr uint32_t LOOK_LOCAL = 1u, LOOK_GLOBAL = 2u;
__device__ __forceinline__ u64 look_word(uint32_t pass, uint32_t state, uint32_t count)
{
    return ((u64)pass << 32) | ((u64)state << 30) | count;
}
// two counts in one status word: [pass:20][state:2][a:21][b:21]
__device__ __forceinline__ u64 look2_word(uint32_t pass, uint32_t state, uint32_t ca, uint32_t cb)
{
    return ((u64)(pass & 0xFFFFFu) << 44) | ((u64)state << 42) | ((u64)ca << 21) | cb;
}

// Exclusive prefix of a tile's count over the earlier tiles of its block: decoupled look-back, run by ONE WHOLE
// WAVEFRONT (lane j inspects tile t - j, so 64 predecessors cost one round trip: with every tile of a block in
// flight at once a one-tile-at-a-time walk is a chain of hundreds of dependent loads).  Publishes the tile's own
// status words on the way.  Status word: [pass:32][state:2][count:30].  Returns the prefix in every lane.
__device__ __forceinline__ uint32_t lookback_wave(u64 *st, uint32_t tile, uint32_t pass, uint32_t own, uint32_t *err,
                                                   bool published, unsigned long long budget)
{
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    if (tile > 0) {
        if (lane == 0 && !published) __hip_atomic_store(st + tile, look_word(pass, LOOK_LOCAL, own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int t = (int)tile - 1; // nearest predecessor not yet summed
        LookWait lw;
        for (;;) {
            const int idx = t - lane;
            u64 w = look_word(pass, LOOK_GLOBAL, 0u); // before tile 0: a known prefix of nothing
            if (idx >= 0) w = __hip_atomic_load(st + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t state = (uint32_t)(w >> 30) & 3u;
            const bool ready = (uint32_t)(w >> 32) == pass && state != 0u;
            const u64 nr = __ballot(!ready);
            const int first_nr = nr ? __ffsll((long long)nr) - 1 : 64;
            const u64 usable = first_nr == 64 ? ~0ull : ((1ull << first_nr) - 1ull);
            const u64 gl = __ballot(ready && state == LOOK_GLOBAL) & usable;
            const int upto = gl ? __ffsll((long long)gl) - 1 : first_nr - 1; // last lane whose count is taken
            acc += wave_reduce_add(lane <= upto ? (uint32_t)w & 0x3FFFFFFFu : 0u);
            if (gl) break;
            t -= first_nr;
            if (first_nr == 0) {
                if (lw.expired(budget)) { // 20 ms: only a logic error or a shared GPU gets here
                    if (lane == 0) atomicOr(err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    if (lane == 0) __hip_atomic_store(st + tile, look_word(pass, LOOK_GLOBAL, acc + own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return acc;
}

// MODE == GEN_GID (first pass over the upper 3 bytes of the 8-byte prefix): the list arrives ordered by bytes
// 3..7; an element leaves with [bytes 0..2 : 24][dense index of its 5-byte group : 20][suffix : 20], so the
// refinement after the last pass compares (bytes 0..2, index) pairs and never goes back to the text.  The index
// is a prefix count of key changes: inside the tile by ballots, over the earlier tiles by a look-back (gst).
// Single pass: no histogram / scan launches before this kernel.  The tile publishes its digit
// counts, then each digit's thread looks back over the earlier tiles of the block (decoupled
// look-back: a predecessor offers either its own counts or, once known, its inclusive prefix) for
// the tile's first slot; digit bases come from a.dbase.  A status word is one 64-bit atomic, so no
// fences are needed; tiles only ever wait for LOWER workgroup ids, which the dispatcher starts
// first; waits are bounded (a.err) so that a logic error cannot hang the device.
template <int BITS, int MODE>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter(SortArgs a)
{
    constexpr int NB = 1 << BITS;
    constexpr int NW = SORT_THREADS / 64;
    if (MODE == GEN_LIST && a.has_n && !a.src_n && *a.gwide == 0u) return; // the fifth pass of a round whose keys have 32 bits
    uint32_t b, tile;
    if (!wg_map(a.T, a.lst, b, tile)) return;
    if (MODE == GEN_LIST && a.mid_np && ms_block_is_mid(a.mid_np, a.mid_spans, b)) return; // (round 0: mid_sort's blocks)
    const uint32_t cnt = a.cnt[b], n = a.n[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    launch_check(a.T, a.lst, tile, ntile, a.err);
    if (tile >= ntile) return;
    const uint32_t h = MODE == GEN_SWEEP ? a.hb[b] : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == GEN_LIST && a.has_n && *a.gwide == 0u) { // 32-bit keys this round: the other chain of buffers
        a.src = a.src_n;
        a.dst = a.dst_n;
    }

    __shared__ uint32_t cur[NW][NB];  // per-wave counters, then cursors (tile-local positions)
    __shared__ uint32_t binstart[NB]; // tile-local start of each digit's run
    __shared__ uint32_t goff[NB];     // global offset of this tile's run of each digit
    __shared__ u64 stage[SORT_TILE];  // tile in digit order
    __shared__ uint32_t ls[NW + 2];
    for (int k = threadIdx.x; k < NW * NB; k += SORT_THREADS) (&cur[0][0])[k] = 0;

    // wave w owns the contiguous run [w*ITEMS*64, (w+1)*ITEMS*64) of the tile, 64 elements a step
    u64 v[SORT_ITEMS];
    uint32_t actmask = 0;
    uint32_t gid_total = 0;
    if (MODE == GEN_GID) {
        const u64 *src = a.src + (size_t)b * a.S;
        const uint32_t e0w = tile * SORT_TILE + wave * (SORT_ITEMS * 64);
        // the element before my wavefront's run (lane 0 of the first step compares with it)
        u64 carry = (e0w > 0 && e0w <= cnt) ? src[e0w - 1] : 0ull;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t e = e0w + k * 64 + lane;
            v[k] = e < cnt ? src[e] : 0ull;
            if (e < cnt) actmask |= 1u << k;
        }
        uint32_t incp[SORT_ITEMS / 2]; // key changes up to and including the element, inside the run: 16 bits each
#pragma unroll
        for (int k = 0; k < SORT_ITEMS / 2; k++) incp[k] = 0;
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t klo = (uint32_t)(v[k] >> 20), khi = (uint32_t)(v[k] >> 52) & 0xFFu; // the 40 key bits
            uint32_t plo = (uint32_t)__shfl_up((int)klo, 1, 64), phi = (uint32_t)__shfl_up((int)khi, 1, 64);
            if (lane == 0) {
                plo = (uint32_t)(carry >> 20);
                phi = (uint32_t)(carry >> 52) & 0xFFu;
            }
            const bool f = ((actmask >> k) & 1u) && (e0w + k * 64 + lane == 0 || klo != plo || khi != phi);
            carry = ((u64)(uint32_t)__shfl((int)khi, 63, 64) << 52) | ((u64)(uint32_t)__shfl((int)klo, 63, 64) << 20);
            const u64 m = __ballot(f);
            const uint32_t inc = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) + (f ? 1u : 0u);
            incp[k >> 1] |= inc << (16 * (k & 1));
            run += (uint32_t)__popcll(m);
        }
        if (lane == 0) ls[wave] = run;
        __syncthreads();
        uint32_t wpre = 0, gtot = 0; // key changes in the earlier wavefronts of the tile / in the whole tile
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = ls[w];
            if (w < wave) wpre += c;
            gtot += c;
        }
        // The tile's count goes out right away; the count of the earlier tiles is only needed when the elements
        // leave (the index does not take part in this pass's digit), so the look-back runs beside the digits'
        // look-back further down and finds its predecessors long published.
        if (threadIdx.x == 0 && tile > 0)
            __hip_atomic_store(a.gst + (size_t)b * a.TPB * 2 + tile, look_word(a.pass, LOOK_LOCAL, gtot), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        gid_total = gtot;
        const uint8_t *txt = a.blk + (size_t)b * a.S;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            if ((actmask >> k) & 1u) {
                const uint32_t i = (uint32_t)(v[k] & SUF_MASK);
                const uint32_t gid = wpre + ((incp[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
                v[k] = ((u64)(text4(txt, i, n) >> 8) << 40) | ((u64)gid << 20) | i;
            }
        }
    } else if (MODE == GEN_BYTES5 && tile * SORT_TILE + wave * (SORT_ITEMS * 64) + SORT_ITEMS * 64 + 16 <= n) {
        // The first pass has no order to keep (any order of equal keys does), so a lane takes 16 CONSECUTIVE suffixes
        // and reads their 20 text bytes with two aligned 16-byte loads instead of one load per suffix.
        const uint32_t e0 = tile * SORT_TILE + wave * (SORT_ITEMS * 64) + lane * SORT_ITEMS;
        const uint4 *tp = reinterpret_cast<const uint4 *>(a.blk + (size_t)b * a.S + e0);
        const uint4 q0 = tp[0], q1 = tp[1];
        const uint32_t d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        actmask = 0xFFFFu;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const int o = k + 3, wi = o >> 2, sh = (o & 3) * 8; // bytes o .. o+4 of the window = bytes 3..7 of rotation e0+k
            const u64 lo = ((u64)d[wi + 1] << 32) | d[wi];
            const uint32_t b3456 = (uint32_t)(lo >> sh);                                            // byte 3 lowest
            const uint32_t b7 = (uint32_t)((((u64)d[wi + 2] << 32) | d[wi + 1]) >> sh) & 255u;
            v[k] = ((u64)__builtin_bswap32(b3456) << 28) | ((u64)b7 << 20) | (e0 + k);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t e = tile * SORT_TILE + wave * (SORT_ITEMS * 64) + k * 64 + lane;
            v[k] = 0;
            if (e < cnt && gen_elem<MODE>(a, b, e, n, h, v[k])) actmask |= 1u << k;
        }
    }
    __syncthreads();
    // Rank inside the wavefront, once: a step's 64 elements are grouped by digit with BITS ballots
    // (match-any); every lane reads its digit's running count, the lowest lane of each group then
    // adds the group size (LDS operations of one wavefront execute in order, so the next step sees
    // it); wr = count before the step + lanes of the group below this one = the element's stable
    // rank among the wavefront's elements with that digit.
    uint32_t wr[SORT_ITEMS / 2]; // 16 bits each
#pragma unroll
    for (int k = 0; k < SORT_ITEMS / 2; k++) wr[k] = 0;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const bool act = (actmask >> k) & 1u;
        const uint32_t d = (uint32_t)(v[k] >> a.shift) & (NB - 1);
        // match-any: keep the lanes whose digit agrees with mine in every bit.  Inactive lanes are
        // outside the initial mask, so their (arbitrary) digit bits need no masking in the ballots.
        const u64 m0 = __ballot(act);
        uint32_t mlo = (uint32_t)m0, mhi = (uint32_t)(m0 >> 32);
#pragma unroll
        for (int bit = 0; bit < BITS; bit++) {
            const int om = ((int)(d << (31 - bit))) >> 31; // all ones if my digit has the bit
            const u64 bm = __builtin_amdgcn_ballot_w64(om != 0);
            mlo &= ~((uint32_t)bm ^ (uint32_t)om);
            mhi &= ~((uint32_t)(bm >> 32) ^ (uint32_t)om);
        }
        if (act) {
            const uint32_t before = cur[wave][d];
            const uint32_t off = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            if (off == 0) cur[wave][d] = before + (uint32_t)(__popc(mlo) + __popc(mhi));
            wr[k >> 1] |= (before + off) << (16 * (k & 1));
        }
        // the next step's reads must follow this store in program order (another lane wrote the
        // count I read next); the DS unit keeps one wavefront's operations in order, so a compiler
        // barrier is all that is needed -- `volatile` would turn these into flat sc0 sc1 accesses
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    // digit totals -> tile-local exclusive starts; cursors = start + counts of earlier waves
    uint32_t mytot = 0;
    if (threadIdx.x < NB) {
#pragma unroll
        for (int w = 0; w < NW; w++) mytot += cur[w][threadIdx.x];
    }
    uint32_t tile_total;
    const uint32_t ex = block_excl_add(mytot, ls, &tile_total);
    const bool mute = a.fault != 0u && b == 0u && tile == 1u; // (test hook: this tile's successors must give up, not hang)
    if (threadIdx.x < NB) {
        const uint32_t bin = threadIdx.x;
        binstart[bin] = ex;
        if (!mute)
            __hip_atomic_store(a.look + ((size_t)b * a.TPB + tile) * NBMAX + bin, look_word(a.pass, LOOK_LOCAL, mytot),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t g = ex;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t t = cur[w][bin];
            cur[w][bin] = g;
            g += t;
        }
    }
    __syncthreads();
    // placement: tile-local slot = start of (digit, wave) + rank inside the wave
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        if ((actmask >> k) & 1u) {
            const uint32_t d = (uint32_t)(v[k] >> a.shift) & (NB - 1);
            stage[cur[wave][d] + ((wr[k >> 1] >> (16 * (k & 1))) & 0xFFFFu)] = v[k];
        }
    }
    if (MODE == GEN_GID && wave == NW - 1) { // (the digits keep the first NB threads busy)
        const uint32_t pre = lookback_wave(a.gst + (size_t)b * a.TPB * 2, tile, a.pass, gid_total, a.err, true, a.patient ? LOOK_TICKS_PATIENT : LOOK_TICKS);
        if (lane == 0) ls[NW + 1] = pre;
    }
    if (threadIdx.x < NB) { // look back for the counts of digit `bin` in tiles 0 .. tile-1
        const uint32_t bin = threadIdx.x;
        u64 *col = a.look + (size_t)b * a.TPB * NBMAX + bin;
        uint32_t acc = 0;
        LookWait lw;
        int t = (int)tile - 1;
        while (t >= 0) {
            const u64 w = __hip_atomic_load(col + (size_t)t * NBMAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t state = (uint32_t)(w >> 30) & 3u;
            if ((uint32_t)(w >> 32) != a.pass || state == 0) { // predecessor has not published yet
                if (lw.expired(a.fault ? LOOK_TICKS_FAULT : a.patient ? LOOK_TICKS_PATIENT : LOOK_TICKS)) { // only a logic error or a shared GPU gets here
                    atomicOr(a.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            acc += (uint32_t)w & 0x3FFFFFFFu;
            if (state == LOOK_GLOBAL) break;
            t--;
        }
        if (!mute)
            __hip_atomic_store(col + (size_t)tile * NBMAX, look_word(a.pass, LOOK_GLOBAL, acc + mytot), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        goff[bin] = a.dbase[(size_t)b * DB_STRIDE + a.doff + bin] + acc;
    }
    __syncthreads();
    u64 *dst = a.dst + (size_t)b * a.S;
    // (the staged elements count the key changes from the tile's start, 1-based; the index is 0-based over the block)
    const u64 gadd = MODE == GEN_GID ? ((u64)ls[NW + 1] << 20) - (1ull << 20) : 0ull;
    for (uint32_t e = threadIdx.x; e < tile_total; e += SORT_THREADS) {
        const u64 x = stage[e];
        const uint32_t d = (uint32_t)(x >> a.shift) & (NB - 1);
        dst[goff[d] + (e - binstart[d])] = x + gadd;
    }
}

// ---- group refinement ---------------------------------------------------------------------------
// small groups (<= TAIL_G members) are ranked locally (tail_round)
constexpr int TAIL_G = 64;

struct RefineArgs {
    const uint32_t *n;   // [B]
    const uint32_t *cnt; // [B] length of the sorted list (n for the init pass)
    const u64 *list;     // [B][S] sorted elements
    u64 *big;            // [B][S] the block's OTHER list buffer: receives the suffixes of large groups, compacted, in order
    u64 *tail0, *tail1;  // [B][S] the two small-group list buffers; the block's records of this round go to the one
    const uint32_t *tdst; // [B]   st_tdst names (behind `tbase`)
    const uint32_t *tbase; // [B] records already in the small-group list (SPLIT-mode blocks; survivors of this round)
    uint32_t *mode;        // [B] st_mode (flag_carry of the init pass decides the first mode)
    u64 *cstat;          // tile status words of the compaction's look-back (word 192 of the tile's hist row)
    uint32_t cpass;      // pass id in those words
    u64 *carry;          // refine_one: [B][TPB][2] tile status words of the carry look-back (slot 1 of a pair)
    uint32_t bpass;      // refine_one<init>: pass id of the rank binning's status words (rows of cstat)
    uint32_t *err;       // bit 1: a look-back gave up
    uint32_t patient;    // 1: the pinned retry (LOOK_TICKS_PATIENT)
    const uint8_t *blk;  // [B][S] the text (init pass: low half of the 8-byte prefix is compared from it)
    uint8_t *bwt;        // [B][S] the last column (init pass: the bytes of the rotations it resolves)
    uint32_t *rank;      // [B][S]
    uint32_t *sa;        // [B][S]
    uint32_t *headp;     // [B][S]
    uint8_t *flg;        // [B][S]
    int4 *tagg;          // [B][TPB]
    uint32_t *dig;       // [B][TPB][512]: per tile, counts of the three 7-bit digits of the new rank over the suffixes
                         // left unresolved (bases of the next SWEEP round's look-back passes; SWEEP-mode blocks only)
    uint32_t *c_big, *c_small, *c_prog; // [B] results: list lengths, "a group was refined"
    uint32_t *c_nolist;  // [B] result: the lists were not written (see refine)
    const uint32_t *nbig_in; // [B] suffixes in large groups when the round began
    uint32_t S, TPB;
    int init;
    uint32_t T;
    Lst lst;
    uint32_t dig_stride;   // words between two blocks' digit rows (sweep_bases; 0: TPB * 512, the rows of `hist`)
    GidOut gout;           // numbers for the large groups this kernel writes (for the NEXT round's sort)
    const uint32_t *grank; // [B][GID_MAX] refine_one: the rank of every numbered group of THIS round
    const uint32_t *gwide; // [1] refine_one: 0 = the sorted list carries group numbers, not ranks (this round's word)
    // refine_one, blocks whose big lists mid_sort orders (bwt_msd.h; mid_np null: none): the list is refined tile by tile of
    // mid_plan's table -- whole groups, so no carries and no look-back --, the records carry ranks, and the large groups
    // leave as ONE RUN per tile, claimed on the block's run counter of the next round (the next mid_plan packs those runs)
    const uint32_t *mid_np, *mid_spans;
    const uint2 *mid_tiles;      // [B][MS_UNIT_CAP]
    const uint32_t *mid_ntiles;  // [B]
    uint32_t *mid_runq;          // [B] records | runs << 20 of the list being written
    uint2 *mid_runs;             // [B][MS_UNIT_CAP]
};

// LDS staging of one tile: coalesced global loads, then each thread owns 16 consecutive elements.
// Element e lives at slot e + (e >> 4): the +1 per 16 keeps the blocked ds_read_b64 conflict free.
constexpr int STAGE_SLOTS = SORT_TILE + SORT_TILE / 16;
__device__ __forceinline__ uint32_t slot_of(uint32_t e) { return e + (e >> 4); }

__device__ __forceinline__ void stage_tile(const u64 *list, uint32_t tile0, uint32_t cnt, u64 *lds)
{
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t e = k * SORT_THREADS + threadIdx.x;
        const uint32_t q = tile0 + e;
        lds[slot_of(e)] = q < cnt ? list[q] : 0ull;
    }
}

__device__ __forceinline__ void elem_flags(const RefineArgs &a, uint32_t q, u64 cur, u64 prev, bool &gs, bool &bd)
{
    if (a.init) { // [bytes 0..2][index of the 5-byte group][suffix]
        gs = (q == 0);
        bd = gs || (cur >> 20) != (prev >> 20);
    } else {
        gs = (q == 0) || (cur >> 40) != (prev >> 40);
        bd = gs || (cur >> 20) != (prev >> 20);
    }
}

// flag bit0: first element of its (old) group; bit1: first element of its refined group.
// Tile aggregates: last group start, last boundary, FIRST boundary of the tile.
__global__ void __launch_bounds__(SORT_THREADS) flag_tiles(RefineArgs a)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.lst, b, tile)) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    launch_check(a.T, a.lst, tile, ntile, a.err);
    if (tile >= ntile) return;
    const size_t base = (size_t)b * a.S;
    const u64 *list = a.list + base;
    const uint32_t tile0 = tile * SORT_TILE;
    __shared__ u64 lds[STAGE_SLOTS];
    stage_tile(list, tile0, cnt, lds);
    __syncthreads();
    const uint32_t e0 = threadIdx.x * SORT_ITEMS, q0 = tile0 + e0;
    uint32_t packed[SORT_ITEMS / 4] = {0, 0, 0, 0};
    int lastgs = -1, lastbd = -1, firstbd = INT32_MAX, nbd = 0;
    if (q0 < cnt) {
        u64 prev = e0 ? lds[slot_of(e0 - 1)] : (q0 ? list[q0 - 1] : 0ull);
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (q < cnt) {
                const u64 cur = lds[slot_of(e0 + k)];
                bool gs, bd;
                elem_flags(a, q, cur, prev, gs, bd);
                if (gs) lastgs = (int)q;
                if (bd) {
                    lastbd = (int)q;
                    if (firstbd == INT32_MAX) firstbd = (int)q;
                    nbd++;
                }
                packed[k >> 2] |= ((gs ? 1u : 0u) | (bd ? 2u : 0u)) << ((k & 3) * 8);
                prev = cur;
            }
        }
        *reinterpret_cast<uint4 *>(a.flg + base + q0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
    __shared__ int red[4][SORT_THREADS / 64];
    int g = lastgs, d = lastbd, f = firstbd, c = nbd;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        g = max(g, __shfl_xor(g, s, 64));
        d = max(d, __shfl_xor(d, s, 64));
        f = min(f, __shfl_xor(f, s, 64));
        c += __shfl_xor(c, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = g;
        red[1][threadIdx.x >> 6] = d;
        red[2][threadIdx.x >> 6] = f;
        red[3][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SORT_THREADS / 64; w++) {
            g = max(g, red[0][w]);
            d = max(d, red[1][w]);
            f = min(f, red[2][w]);
            c += red[3][w];
        }
        a.tagg[(size_t)b * a.TPB + tile] = make_int4(g, d, f, c); // w: groups that start in the tile
    }
}

// One workgroup per listed block: exclusive max-scan of (last group start, last boundary) from the left and
// exclusive min-scan of the first boundary from the right over the tile aggregates (carries into each tile).
// ntile <= TPB <= 1024 (checked at context creation), so one sweep of 1024 threads covers it.
__global__ void __launch_bounds__(1024) flag_carry(RefineArgs a)
{
    const uint32_t kb = blockIdx.x;
    if (kb >= (a.lst.ids ? *a.lst.cnt : a.lst.B)) return;
    const uint32_t b = a.lst.ids ? a.lst.ids[kb] : kb;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (ntile == 0) return;
    int4 *t = a.tagg + (size_t)b * a.TPB;
    __shared__ int l01[32];
    __shared__ int inc0[1024], inc1[1024];
    const uint32_t e = threadIdx.x;
    const int4 v = e < ntile ? t[e] : make_int4(-1, -1, INT32_MAX, 0);
    int s0 = v.x, s1 = v.y;
    block_incl_max2(s0, s1, l01);
    inc0[e] = s0;
    inc1[e] = s1;
    const int nx = block_excl_min_rev(v.z, l01); // also a barrier: inc0/inc1 are visible after it
    if (e < ntile)
        t[e] = make_int4(e == 0 ? -1 : inc0[e - 1], e == 0 ? -1 : inc1[e - 1], nx == INT32_MAX ? (int)cnt : nx, 0);
    if (a.init) {
        // First mode of the block, before its first refine: with fewer than one group per 8 suffixes after the
        // 8-byte sort the block is run-heavy / periodic and starts in SWEEP mode (refine then keeps provisional
        // SA order and group heads at every position); text-like blocks start in SPLIT mode and never pay for that.
        __shared__ uint32_t lsum[1024 / 64 + 2];
        uint32_t groups;
        (void)block_excl_add(e < ntile ? (uint32_t)v.w : 0u, lsum, &groups);
        if (e == 0) a.mode[b] = ((uint64_t)groups * 8u < cnt) ? 0u : 1u;
    }
}

// Element classes after refinement
constexpr uint32_t CLS_SINGLE = 0u, CLS_SMALL = 1u, CLS_BIG = 2u;

__global__ void __launch_bounds__(SORT_THREADS) refine(RefineArgs a)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.lst, b, tile)) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    launch_check(a.T, a.lst, tile, ntile, a.err);
    if (tile >= ntile) return;
    // a block that may enumerate SA positions next round needs SA / group heads at EVERY position and the
    // digit counts of its unresolved ranks; a SPLIT-mode block only needs the final SA entries
    const bool sweep = a.mode[b] == 0u;
    // A SWEEP-mode block that entered the round with most of its suffixes in large groups stays in SWEEP mode
    // (which reads no list), so its lists are not written; round_begin keeps such a block from changing mode.
    const bool nolist = sweep && !a.init && (uint64_t)a.nbig_in[b] * 2u >= a.n[b];
    const size_t base = (size_t)b * a.S;
    const uint32_t tile0 = tile * SORT_TILE;
    __shared__ u64 lds[STAGE_SLOTS];
    const uint32_t e0 = threadIdx.x * SORT_ITEMS, q0 = tile0 + e0;

    uint32_t packed[4] = {0, 0, 0, 0};
    if (q0 < cnt) {
        const uint4 f = *reinterpret_cast<const uint4 *>(a.flg + base + q0);
        packed[0] = f.x;
        packed[1] = f.y;
        packed[2] = f.z;
        packed[3] = f.w;
    }
    // group extents: last group start / boundary at or before every element (scan from the left), first
    // boundary behind it (scan from the right); a group's size is the distance between its boundaries
    int tg = -1, td = -1, fb = INT32_MAX;
    bool progress = false;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
        if (q0 + k < cnt) {
            if (f & 1u) tg = (int)(q0 + k);
            if (f & 2u) {
                td = (int)(q0 + k);
                if (fb == INT32_MAX) fb = (int)(q0 + k);
                if (!(f & 1u)) progress = true; // a boundary inside an old group: that group was refined
            }
        }
    }
    __shared__ int l01[2 * SORT_THREADS / 64];
    __shared__ int ex0[SORT_THREADS], ex1[SORT_THREADS];
    block_incl_max2(tg, td, l01);
    ex0[threadIdx.x] = tg;
    ex1[threadIdx.x] = td;
    int nxt = block_excl_min_rev(fb, l01); // barrier inside: ex0 / ex1 visible
    const int4 tc = a.tagg[(size_t)b * a.TPB + tile];
    int cg = tc.x, cd = tc.y;
    if (threadIdx.x > 0) {
        cg = max(cg, ex0[threadIdx.x - 1]);
        cd = max(cd, ex1[threadIdx.x - 1]);
    }
    if (nxt == INT32_MAX) nxt = tc.z; // first boundary behind this tile (the list end counts as one)
    // backward: end of every element's group; forward: its start -> class, 2 bits per element
    uint32_t gend[SORT_ITEMS];
    {
        int run = nxt;
#pragma unroll
        for (int k = SORT_ITEMS - 1; k >= 0; k--) {
            gend[k] = (uint32_t)run;
            const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
            if (q0 + k < cnt && (f & 2u)) run = (int)(q0 + k);
        }
    }
    uint32_t cls = 0, nS = 0, nB = 0;
    {
        int run = cd;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            if (q0 + k < cnt) {
                const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
                if (f & 2u) run = (int)(q0 + k);
                const uint32_t size = gend[k] - (uint32_t)run;
                const uint32_t c = size == 1u ? CLS_SINGLE : (size <= (uint32_t)TAIL_G ? CLS_SMALL : CLS_BIG);
                cls |= c << (2 * k);
                nS += c == CLS_SMALL;
                nB += c == CLS_BIG;
            }
        }
    }
    // compaction of the unresolved records, one stream per class: slot = (class members in earlier tiles:
    // look-back over the tile counts) + (in earlier threads of the tile) + (before the element in the thread)
    __shared__ uint32_t lsu[SORT_THREADS / 64 + 2];
    __shared__ uint32_t cpreS, cpreB;
    uint32_t totS = 0, totB = 0;
    const uint32_t offS = block_excl_add(nS, lsu, &totS);
    const uint32_t offB = block_excl_add(nB, lsu, &totB);
    u64 *cst = a.cstat + (size_t)b * a.TPB * NBMAX + 192;
    if (threadIdx.x == 0)
        __hip_atomic_store(cst + (size_t)tile * NBMAX, look2_word(a.cpass, tile ? LOOK_LOCAL : LOOK_GLOBAL, totS, totB),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stage_tile(a.list + base, tile0, cnt, lds);
    if (threadIdx.x == 0) {
        uint32_t accS = 0, accB = 0;
        LookWait lw;
        if (tile > 0) {
            int t = (int)tile - 1;
            while (t >= 0) {
                const u64 w = __hip_atomic_load(cst + (size_t)t * NBMAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t state = (uint32_t)(w >> 42) & 3u;
                if ((uint32_t)(w >> 44) != (a.cpass & 0xFFFFFu) || state == 0) {
                    if (lw.expired(a.patient ? LOOK_TICKS_PATIENT : LOOK_TICKS)) {
                        atomicOr(a.err, 2u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                accS += (uint32_t)(w >> 21) & 0x1FFFFFu;
                accB += (uint32_t)w & 0x1FFFFFu;
                if (state == LOOK_GLOBAL) break;
                t--;
            }
            __hip_atomic_store(cst + (size_t)tile * NBMAX, look2_word(a.cpass, LOOK_GLOBAL, accS + totS, accB + totB),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cpreS = accS;
        cpreB = accB;
        if (tile == ntile - 1) { // the block's list lengths after this round
            a.c_small[b] = accS + totS;
            a.c_big[b] = accB + totB;
            if (nolist) a.c_nolist[b] = 1u;
        }
    }
    __syncthreads(); // cpre*, and stage_tile's stores before the blocked reads below
    const uint32_t tbase = (a.init || sweep) ? 0u : a.tbase[b];
    uint32_t *rank = a.rank + base;
    uint32_t *sa = a.sa + base;
    uint32_t *headp = a.headp + base;
    u64 *big = a.big + base;
    u64 *tail = (a.tdst[b] ? a.tail1 : a.tail0) + base;
    __shared__ uint32_t dh[384];
    if (sweep) {
        for (int k = threadIdx.x; k < 384; k += SORT_THREADS) dh[k] = 0;
        __syncthreads();
    }
    uint32_t ph = 0, pc = 0, ph7 = 0, pc7 = 0; // open runs of the digit counting
    u64 outv[SORT_ITEMS]; // class : SA position : group rank : suffix (2 + 3 x 20 bits), all ones = none
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) outv[k] = ~0ull;
    if (q0 < cnt) {
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (q < cnt) {
                const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
                if (f & 1u) cg = (int)q;
                if (f & 2u) cd = (int)q;
                const uint32_t c = (cls >> (2 * k)) & 3u;
                const u64 cur = lds[slot_of(e0 + k)];
                const uint32_t i = (uint32_t)(cur & SUF_MASK);
                // SA position of the group's first list entry, minus that entry's list index
                const uint32_t gbase = a.init ? 0u : ((uint32_t)(cur >> 40) - (uint32_t)cg);
                const uint32_t pos = gbase + q;
                const uint32_t head = gbase + (uint32_t)cd;
                const bool single = c == CLS_SINGLE;
                if (c == CLS_BIG && (f & 2u) && !nolist) gid_draw(a.gout, b, head);
                rank[rslot(i)] = single ? (head | RANK_RESOLVED) : head;
                outv[k] = ((u64)c << 62) | ((u64)pos << 40) | ((u64)head << 20) | i;
                if (sweep && !single) {
                    // heads rise with q, so a thread's 16 entries share their upper digits (and, inside
                    // a group, the whole head): count runs in registers, touch LDS once per run
                    if (head != ph) {
                        if (pc) atomicAdd(&dh[ph & 127u], pc);
                        ph = head;
                        pc = 0;
                    }
                    pc++;
                    if ((head >> 7) != ph7) {
                        if (pc7) {
                            atomicAdd(&dh[128 + (ph7 & 127u)], pc7);
                            atomicAdd(&dh[256 + (ph7 >> 7)], pc7);
                        }
                        ph7 = head >> 7;
                        pc7 = 0;
                    }
                    pc7++;
                }
            }
        }
    }
    if (pc) atomicAdd(&dh[ph & 127u], pc);
    if (pc7) {
        atomicAdd(&dh[128 + (ph7 & 127u)], pc7);
        atomicAdd(&dh[256 + (ph7 >> 7)], pc7);
    }
    __syncthreads(); // every thread is done with the staged tile
    if (sweep) {
        // provisional SA order and group heads at every position, through LDS so that consecutive lanes store
        // consecutive positions
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) lds[slot_of(e0 + k)] = outv[k];
        __syncthreads();
        uint32_t *row = a.dig + ((size_t)b * a.TPB + tile) * 512;
        for (int k = threadIdx.x; k < 384; k += SORT_THREADS) row[k] = dh[k];
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t e = k * SORT_THREADS + threadIdx.x;
            const u64 x = lds[slot_of(e)];
            if (x != ~0ull) {
                const uint32_t pos = (uint32_t)(x >> 40) & 0xFFFFFu;
                sa[pos] = (uint32_t)(x & SUF_MASK);
                headp[pos] = (uint32_t)(x >> 20) & 0xFFFFFu;
            }
        }
        __syncthreads();
    }
    // the ranked records of the unresolved suffixes leave through LDS as well: the tile's small-group records at
    // [0, totS), its large-group records behind them, then coalesced copies to the two lists
    if (__ballot(progress) && (threadIdx.x & 63) == 0) a.c_prog[b] = 1u; // same value from everyone
    if (nolist) return;
    {
        uint32_t wS = offS, wB = totS + offB;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            if (outv[k] != ~0ull) {
                const uint32_t c = (uint32_t)(outv[k] >> 62);
                const u64 rec = (((outv[k] >> 20) & 0xFFFFFull) << 40) | (outv[k] & SUF_MASK);
                if (c == CLS_SMALL) lds[wS++] = rec;
                if (c == CLS_BIG) lds[wB++] = rec;
            }
        }
    }
    __syncthreads();
    {
        u64 *ts = tail + tbase + cpreS;
        u64 *bs = big + cpreB;
        for (uint32_t e = threadIdx.x; e < totS; e += SORT_THREADS) ts[e] = lds[e];
        for (uint32_t e = threadIdx.x; e < totB; e += SORT_THREADS) bs[e] = lds[totS + e];
    }
}

// ---- refinement in ONE kernel (the initial sort, and the big lists of SPLIT-mode blocks) ---------------------
// What flag_tiles + flag_carry + refine do in three launches and two reads of the list, for blocks that need
// neither provisional SA entries nor digit counts (i.e. everything but SWEEP mode):
//  * the extent of a group to the LEFT (its head position = its new rank) is carried from tile to tile by a
//    look-back over per-tile aggregates (last group start, last boundary) that every tile publishes right after
//    its own flags -- no tile waits for another tile's look-back;
//  * to the RIGHT only "does the group end within TAIL_G elements" matters (single / small / large), so a tile
//    reads TAIL_G elements of its successor instead of waiting for it;
//  * neither list needs an order among tiles (the big list is re-sorted from scratch, the groups of the small
//    list are independent), so a tile claims its room with one atomic add per list.  A small group that
//    straddles a tile end is written, whole, by the tile it starts in (its members must stay adjacent); the
//    next tile ranks its part of that group and leaves the records alone.
constexpr int POS_FAR = INT32_MAX; // "no boundary within reach"

// Carries of a tile: last group start / last boundary in any earlier tile of the block (-1: none).  Run by one
// whole wavefront; lane j inspects tile t - j.  Status word: look2_word(pass, state, start + 1, boundary + 1).
__device__ __forceinline__ void carry_lookback(u64 *st, uint32_t tile, uint32_t pass, int own_gs, int own_bd, bool need_gs,
                                               int &cg, int &cd, uint32_t *err, unsigned long long budget) // st: one word per tile, 2 words apart
{
    const int lane = threadIdx.x & 63;
    const uint32_t egs = (uint32_t)(own_gs + 1), ebd = (uint32_t)(own_bd + 1);
    uint32_t fg = 0, fb = 0;
    if (tile > 0) {
        if (lane == 0) __hip_atomic_store(st + (size_t)tile * 2, look2_word(pass, LOOK_LOCAL, egs, ebd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool hg = !need_gs, hb = false;
        int t = (int)tile - 1;
        LookWait lw;
        for (;;) {
            const int idx = t - lane;
            u64 w = look2_word(pass, LOOK_GLOBAL, 0u, 0u); // before tile 0: nothing
            if (idx >= 0) w = __hip_atomic_load(st + (size_t)idx * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t state = (uint32_t)(w >> 42) & 3u;
            const bool ready = (uint32_t)(w >> 44) == (pass & 0xFFFFFu) && state != 0u;
            const uint32_t vg = (uint32_t)(w >> 21) & 0x1FFFFFu, vb = (uint32_t)w & 0x1FFFFFu;
            const u64 nr = __ballot(!ready);
            const int first_nr = nr ? __ffsll((long long)nr) - 1 : 64;
            const u64 usable = first_nr == 64 ? ~0ull : ((1ull << first_nr) - 1ull);
            const u64 mg = __ballot(ready && state == LOOK_GLOBAL) & usable; // inclusive over everything before
            if (!hb) {
                const u64 m = (__ballot(ready && vb != 0u) & usable) | mg;
                if (m) {
                    fb = (uint32_t)__shfl((int)vb, __ffsll((long long)m) - 1, 64);
                    hb = true;
                }
            }
            if (!hg) {
                const u64 m = (__ballot(ready && vg != 0u) & usable) | mg;
                if (m) {
                    fg = (uint32_t)__shfl((int)vg, __ffsll((long long)m) - 1, 64);
                    hg = true;
                }
            }
            if (hb && hg) break;
            t -= first_nr;
            if (first_nr == 0) {
                if (lw.expired(budget)) {
                    if (lane == 0) atomicOr(err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    if (lane == 0)
        __hip_atomic_store(st + (size_t)tile * 2, look2_word(pass, LOOK_GLOBAL, max(egs, fg), max(ebd, fb)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    cg = (int)fg - 1;
    cd = (int)fb - 1;
}

template <bool INIT>
__global__ void __launch_bounds__(SORT_THREADS) refine_one(RefineArgs a, uint32_t *c_groups, u64 *recs)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.lst, b, tile)) return;
    uint32_t cnt = a.cnt[b];
    const size_t base = (size_t)b * a.S;
    const u64 *list = a.list + base;
    // A block of mid_sort's: my tile is a tile of mid_plan's table, refined as a list of its own (it holds whole groups: what
    // follows sees tile 0 of a list of `cnt` records that begins at the tile's first record)
    const bool midb = !INIT && a.mid_np && ms_block_is_mid(a.mid_np, a.mid_spans, b);
    if (midb) {
        const uint32_t nt = a.mid_ntiles[b];
        launch_check(a.T, a.lst, tile, nt, a.err);
        if (tile >= nt) return;
        const uint2 t = a.mid_tiles[(size_t)b * MS_UNIT_CAP + tile];
        list += t.x;
        cnt = t.y;
        tile = 0;
    } else {
        const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
        launch_check(a.T, a.lst, tile, ntile, a.err);
        if (tile >= ntile) return;
    }
    const uint32_t tile0 = tile * SORT_TILE, tend = tile0 + SORT_TILE;
    constexpr int NW = SORT_THREADS / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ u64 lds[STAGE_SLOTS];
    __shared__ u64 halo[TAIL_G + 1]; // [0]: the element before the tile; [1 + t]: element tend + t
    __shared__ int l01[2 * NW];
    __shared__ int ex0[SORT_THREADS], ex1[SORT_THREADS];
    __shared__ uint32_t lsu[NW + 2];
    __shared__ int s_cg, s_cd, s_hend;
    __shared__ uint32_t s_offS, s_offB;
    __shared__ uint32_t s_ng, s_gbase, glist[SORT_TILE / TAIL_G]; // ranks of the large groups that start in the tile (they get numbers, below)
    __shared__ uint32_t bh[INIT ? 256 : 1], bcur[INIT ? 256 : 1], bgo[INIT ? 256 : 1]; // rank binning: counts, cursors, offsets
    __shared__ uint32_t dh[INIT ? 384 : 1]; // INIT: counts of the three 7-bit digits of the unresolved heads (a SWEEP start needs them)
    if (INIT && threadIdx.x < 256) bh[threadIdx.x] = 0;
    if (INIT && threadIdx.x < 384) dh[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_ng = 0;
    stage_tile(list, tile0, cnt, lds);
    if (threadIdx.x < (uint32_t)TAIL_G) halo[1 + threadIdx.x] = tend + threadIdx.x < cnt ? list[tend + threadIdx.x] : 0ull;
    if (threadIdx.x == (uint32_t)TAIL_G) halo[0] = tile0 ? list[tile0 - 1] : 0ull;
    __syncthreads();

    // flags of my 16 elements -- bit 0: first of its (old) group, bit 1: first of its refined group
    const uint32_t e0 = threadIdx.x * SORT_ITEMS, q0 = tile0 + e0;
    uint32_t packed = 0;
    int lastgs = -1, lastbd = -1, firstbd = POS_FAR;
    uint32_t nbd = 0;
    bool progress = false;
    if (q0 < cnt) {
        u64 prev = e0 ? lds[slot_of(e0 - 1)] : halo[0];
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (q < cnt) {
                const u64 cur = lds[slot_of(e0 + k)];
                // init: [bytes 0..2][index of the 5-byte group][suffix], one "old group"; rounds: [rank][key2][suffix]
                const bool gs = q == 0 || (!INIT && (cur >> 40) != (prev >> 40));
                const bool bd = gs || (cur >> 20) != (prev >> 20);
                if (gs) lastgs = (int)q;
                if (bd) {
                    lastbd = (int)q;
                    if (firstbd == POS_FAR) firstbd = (int)q;
                    nbd++;
                    if (!gs) progress = true; // a boundary inside an old group: that group was refined
                }
                packed |= ((gs ? 1u : 0u) | (bd ? 2u : 0u)) << (2 * k);
                prev = cur;
            }
        }
    }
    // first boundary behind the tile, as far as it matters: within TAIL_G elements (the list end counts)
    if (wave == 0) {
        bool bd = false;
        const uint32_t q = tend + lane;
        if (q == cnt) {
            bd = true;
        } else if (q < cnt) {
            const u64 cur = halo[1 + lane], prev = lane ? halo[lane] : lds[slot_of(SORT_TILE - 1)];
            bd = (cur >> 20) != (prev >> 20);
        }
        const u64 m = __ballot(bd);
        if (lane == 0) s_hend = cnt <= tend ? (int)cnt : (m ? (int)(tend + __ffsll((long long)m) - 1) : POS_FAR);
    }
    int tg = lastgs, td = lastbd;
    block_incl_max2(tg, td, l01);
    ex0[threadIdx.x] = tg;
    ex1[threadIdx.x] = td;
    int nxt = block_excl_min_rev(firstbd, l01); // barrier inside: ex0 / ex1 / s_hend visible
    if (wave == 0) {
        int cgi = -1, cdi = -1;
        if (!midb)
            carry_lookback(a.carry + (size_t)b * a.TPB * 2 + 1, tile, a.cpass, ex0[SORT_THREADS - 1], ex1[SORT_THREADS - 1], !INIT,
                           cgi, cdi, a.err, a.patient ? LOOK_TICKS_PATIENT : LOOK_TICKS);
        if (lane == 0) {
            s_cg = cgi;
            s_cd = cdi;
        }
    }
    if (INIT) { // groups of the block: round_begin picks the block's first mode from it
        const uint32_t c = wave_reduce_add(nbd);
        if (lane == 0 && c) atomicAdd(&c_groups[b], c);
    }
    __syncthreads();
    const int cg_in = s_cg, cd_in = s_cd, hend = s_hend;
    int cg = cg_in, cd = cd_in;
    if (threadIdx.x > 0) {
        cg = max(cg, ex0[threadIdx.x - 1]);
        cd = max(cd, ex1[threadIdx.x - 1]);
    }
    if (nxt == POS_FAR) nxt = hend;
    // backward: end of every element's group; forward: its start -> class, 2 bits per element
    uint32_t gend[SORT_ITEMS];
    {
        int run = nxt;
#pragma unroll
        for (int k = SORT_ITEMS - 1; k >= 0; k--) {
            gend[k] = (uint32_t)run;
            if (q0 + k < cnt && ((packed >> (2 * k)) & 2u)) run = (int)(q0 + k);
        }
    }
    uint32_t cls = 0, nS = 0, nB = 0, foreign = 0;
    {
        int run = cd;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            if (q0 + k < cnt) {
                if ((packed >> (2 * k)) & 2u) run = (int)(q0 + k);
                const uint32_t size = gend[k] - (uint32_t)run;
                const uint32_t c = size == 1u ? CLS_SINGLE : (size <= (uint32_t)TAIL_G ? CLS_SMALL : CLS_BIG);
                cls |= c << (2 * k);
                const bool fo = c == CLS_SMALL && run < (int)tile0; // the tile the group starts in writes its records
                foreign |= (fo ? 1u : 0u) << k;
                nS += c == CLS_SMALL && !fo;
                nB += c == CLS_BIG;
            }
        }
    }
    // the small group that runs over the tile end (if any): its members behind the tile are mine to write
    const int last_start = max(cd_in, ex1[SORT_THREADS - 1]), last_gs = max(cg_in, ex0[SORT_THREADS - 1]);
    const uint32_t nH = (cnt > tend && hend != POS_FAR && (uint32_t)hend > tend && (uint32_t)(hend - last_start) <= (uint32_t)TAIL_G)
                            ? (uint32_t)hend - tend
                            : 0u;
    uint32_t totS = 0, totB = 0;
    const uint32_t offS = block_excl_add(nS, lsu, &totS);
    const uint32_t offB = block_excl_add(nB, lsu, &totB);
    if (threadIdx.x == 0) {
        s_offS = (totS + nH) ? atomicAdd(&a.c_small[b], totS + nH) : 0u;
        if (midb && totB) { // one run of whole groups: its room and its number in list order from ONE atomic add
            const uint32_t q = atomicAdd(&a.mid_runq[b], (1u << 20) | totB);
            s_offB = q & 0xFFFFFu;
            a.mid_runs[(size_t)b * MS_UNIT_CAP + (q >> 20)] = make_uint2(q & 0xFFFFFu, totB);
            atomicAdd(&a.c_big[b], totB); // (the list's length where round_begin looks for it)
        } else {
            s_offB = totB ? atomicAdd(&a.c_big[b], totB) : 0u;
        }
    }
    uint32_t *rank = a.rank + base;
    // INIT: a block may turn out to start in SWEEP mode (round_begin decides from the number of groups this kernel counts):
    // it then needs the suffixes in SA order, the group head at every position and the digit counts of its unresolved
    // heads, so they are left for every block here -- 8 coalesced bytes per suffix and a few LDS adds per run of equal
    // heads -- instead of refining such blocks a second time with the three-kernel form (0.57 ms for the 28 blocks of a
    // near-periodic quarter of config 5).
    const bool narrow = !INIT && !midb && *a.gwide == 0u;
    const uint32_t *grank = a.grank + (size_t)b * GID_MAX; // (this round's half)
    // per element: [class:2 @62][foreign:1 @61][valid:1 @60][head:20 @40][suffix:20 @0]
    u64 outv[SORT_ITEMS];
    uint32_t isuf[INIT ? SORT_ITEMS : 1], ihead[INIT ? SORT_ITEMS : 1]; // INIT: SA order and heads of my 16 positions
    uint32_t ph = 0, pc = 0, ph7 = 0, pc7 = 0;                          // INIT: open runs of the digit counting
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) outv[k] = 0ull;
    if (q0 < cnt) {
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (INIT) isuf[k] = ihead[k] = 0u;
            if (q < cnt) {
                const uint32_t f = (packed >> (2 * k)) & 3u;
                if (f & 1u) cg = (int)q;
                if (f & 2u) cd = (int)q;
                const uint32_t c = (cls >> (2 * k)) & 3u;
                const u64 cur = lds[slot_of(e0 + k)];
                const uint32_t i = (uint32_t)(cur & SUF_MASK);
                // SA position of the group's first list entry, minus that entry's list index
                // (a list sorted on group numbers: the number's rank from gid_scan's table -- the same word for the whole group)
                const uint32_t oldr = (!INIT && narrow) ? grank[(uint32_t)(cur >> 40)] : (uint32_t)(cur >> 40);
                const uint32_t gbase = INIT ? 0u : (oldr - (uint32_t)cg);
                const uint32_t head = gbase + (uint32_t)cd;
                if (c == CLS_BIG && (f & 2u)) glist[atomicAdd(&s_ng, 1u)] = head; // (a large group has more than TAIL_G members: the list cannot overflow)
                // (the last column is emitted from the ranks; a rank that did not move and stays unresolved is in place)
                if (INIT) {
                    atomicAdd(&bh[i >> 12], 1u); // every suffix gets a rank: binned by 4096-suffix window, then applied
                    isuf[k] = i;
                    ihead[k] = head;
                    // (alone in its group: its byte of the last column leaves now, in sorted order -- as in chunk_finish, bwt_msd.h;
                    // the rank word below says so and bwt_emit skips it)
                    if (c == CLS_SINGLE) a.bwt[base + head] = a.blk[base + (i ? i - 1u : a.n[b] - 1u)];
                    if (c != CLS_SINGLE) { // heads rise with q: runs counted in registers, LDS touched once per run (as refine)
                        if (head != ph) {
                            if (pc) atomicAdd(&dh[ph & 127u], pc);
                            ph = head;
                            pc = 0;
                        }
                        pc++;
                        if ((head >> 7) != ph7) {
                            if (pc7) {
                                atomicAdd(&dh[128 + (ph7 & 127u)], pc7);
                                atomicAdd(&dh[256 + (ph7 >> 7)], pc7);
                            }
                            ph7 = head >> 7;
                            pc7 = 0;
                        }
                        pc7++;
                    }
                }
                else if (head != oldr) // (see tail_round: a SPLIT-mode block's "resolved" bits have no reader)
                    rank[rslot(i)] = c == CLS_SINGLE ? (head | RANK_RESOLVED) : head;
                outv[k] = ((u64)c << 62) | ((u64)((foreign >> k) & 1u) << 61) | (1ull << 60) | ((u64)head << 40) | i;
            }
        }
    }
    if (INIT) {
        if (pc) atomicAdd(&dh[ph & 127u], pc);
        if (pc7) {
            atomicAdd(&dh[128 + (ph7 & 127u)], pc7);
            atomicAdd(&dh[256 + (ph7 >> 7)], pc7);
        }
        if (q0 < cnt) { // my 16 consecutive SA positions (q0 is a multiple of 16; a ragged end is written entry by entry)
            uint32_t *sap = a.sa + base + q0, *hpp = a.headp + base + q0;
            if (q0 + SORT_ITEMS <= cnt) {
#pragma unroll
                for (int k = 0; k < SORT_ITEMS; k += 4) {
                    *reinterpret_cast<uint4 *>(sap + k) = make_uint4(isuf[k], isuf[k + 1], isuf[k + 2], isuf[k + 3]);
                    *reinterpret_cast<uint4 *>(hpp + k) = make_uint4(ihead[k], ihead[k + 1], ihead[k + 2], ihead[k + 3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < SORT_ITEMS; k++)
                    if (q0 + k < cnt) {
                        sap[k] = isuf[k];
                        hpp[k] = ihead[k];
                    }
            }
        }
    }
    if (__ballot(progress) && lane == 0) a.c_prog[b] = 1u; // same value from everyone
    __syncthreads(); // every thread is done with the staged tile; s_off* and the bin counts are there
    if (INIT) { // the tile's digit row (read by sweep_bases in round 0 if the block starts in SWEEP mode)
        uint32_t *row = a.dig + (size_t)b * a.dig_stride + (size_t)tile * 512;
        for (int k = threadIdx.x; k < 384; k += SORT_THREADS) row[k] = dh[k];
    }
    // numbers for the tile's large groups: ONE atomic add for all of them, requested now and used at the very end
    uint32_t pend_g = 0;
    const uint32_t ng = midb ? 0u : s_ng; // (mid_sort's lists carry ranks: no numbers)
    if (threadIdx.x == 0 && ng) pend_g = atomicAdd(&a.gout.gcount[b], ng);
    // records leave through LDS: the tile's small-group records at [0, totS + nH), its large-group records behind them
    {
        uint32_t wS = offS, wB = totS + nH + offB;
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t c = (uint32_t)(outv[k] >> 62);
            if (((outv[k] >> 60) & 3ull) == 1ull && c != CLS_SINGLE) { // valid, not foreign, unresolved
                const u64 rec = outv[k] & 0x0FFFFFFFFFFFFFFFull;
                if (c == CLS_SMALL) lds[wS++] = rec;
                else lds[wB++] = rec;
            }
        }
        if (threadIdx.x < nH) { // same (old and refined) group as the tile's last element
            const uint32_t oldr = (!INIT && narrow) ? grank[(uint32_t)(halo[1 + threadIdx.x] >> 40)] : (uint32_t)(halo[1 + threadIdx.x] >> 40);
            const uint32_t head = (INIT ? 0u : oldr - (uint32_t)last_gs) + (uint32_t)last_start;
            lds[totS + threadIdx.x] = ((u64)head << 40) | (halo[1 + threadIdx.x] & SUF_MASK);
        }
    }
    // rank binning, step 1 (threads 0..255 = bins): the tile's first slot of every bin, its count to the look-back
    uint32_t bmine = 0;
    if (INIT) {
        if (threadIdx.x < 256) bmine = bh[threadIdx.x];
        uint32_t btot;
        const uint32_t bex = block_excl_add(bmine, lsu, &btot);
        if (threadIdx.x < 256) {
            bcur[threadIdx.x] = bex;
            bh[threadIdx.x] = bex; // (from here on: the bin's first slot)
            __hip_atomic_store(a.cstat + ((size_t)b * a.TPB + tile) * NBMAX + threadIdx.x, look_word(a.bpass, LOOK_LOCAL, bmine),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) s_gbase = pend_g;
    __syncthreads();
    if (threadIdx.x < ng) gid_assign(a.gout, b, s_gbase + threadIdx.x, glist[threadIdx.x]);
    {
        const uint32_t tbase = INIT ? 0u : a.tbase[b];
        u64 *ts = (a.tdst[b] ? a.tail1 : a.tail0) + base + tbase + s_offS;
        u64 *bs = a.big + base + s_offB;
        const uint32_t nSm = totS + nH;
        for (uint32_t e = threadIdx.x; e < nSm; e += SORT_THREADS) ts[e] = lds[e];
        for (uint32_t e = threadIdx.x; e < totB; e += SORT_THREADS) bs[e] = lds[nSm + e];
    }
    if (!INIT) return;
    // rank binning, step 2: the (rank word, suffix) pairs of the tile in bin order in LDS; the bins' offsets among the
    // earlier tiles of the block by look-back (every bin of the block holds exactly its 4096 suffixes, so the bases
    // need no counting); stores.  rank_apply turns each window into whole lines of the rank array.
    __syncthreads(); // the list records have left LDS
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        if ((outv[k] >> 60) & 1ull) {
            const uint32_t i = (uint32_t)(outv[k] & SUF_MASK), head = (uint32_t)(outv[k] >> 40) & 0xFFFFFu;
            const uint32_t word = (uint32_t)(outv[k] >> 62) == CLS_SINGLE ? (head | RANK_RESOLVED | RANK_EMITTED) : head;
            lds[atomicAdd(&bcur[i >> 12], 1u)] = ((u64)word << 32) | i;
        }
    }
    if (threadIdx.x < 256) {
        const uint32_t bin = threadIdx.x;
        u64 *col = a.cstat + (size_t)b * a.TPB * NBMAX + bin;
        uint32_t acc = 0;
        LookWait lw;
        int t = (int)tile - 1;
        while (t >= 0) {
            const u64 w = __hip_atomic_load(col + (size_t)t * NBMAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t state = (uint32_t)(w >> 30) & 3u;
            if ((uint32_t)(w >> 32) != a.bpass || state == 0) { // predecessor has not published yet
                if (lw.expired(a.patient ? LOOK_TICKS_PATIENT : LOOK_TICKS)) {
                    atomicOr(a.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            acc += (uint32_t)w & 0x3FFFFFFFu;
            if (state == LOOK_GLOBAL) break;
            t--;
        }
        __hip_atomic_store(col + (size_t)tile * NBMAX, look_word(a.bpass, LOOK_GLOBAL, acc + bmine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        bgo[bin] = min(a.n[b], bin * 4096u) + acc;
    }
    __syncthreads();
    {
        u64 *dst = recs + base;
        const uint32_t ntl = min((uint32_t)SORT_TILE, cnt - tile0);
        for (uint32_t e = threadIdx.x; e < ntl; e += SORT_THREADS) {
            const u64 x = lds[e];
            const uint32_t d = ((uint32_t)x & (uint32_t)SUF_MASK) >> 12;
            dst[bgo[d] + (e - bh[d])] = x;
        }
    }
}

// ---- small groups: ranked locally, one kernel per round -------------------------------------------------
// The block's small-group list holds, in `len` slots, the unresolved suffixes of every group with at most
// TAIL_G members; a group's records are adjacent, the order of the groups is arbitrary (refine appends new
// small groups behind the survivors).  Each record carries the suffix and its group rank, so nothing has to
// be gathered but key2.
//
// tail_round does a whole round for these groups in ONE kernel: a workgroup owns the groups whose first member
// lies in its range of TR_T slots and sees TAIL_G slots either side, so every owned group is complete in its
// window; every slot walks outwards over the members of its group (keys in LDS) and counts the members that
// sort before it -- all-pairs work stays below ~6 comparisons per suffix of the block because larger groups
// never come here.  New ranks are stored straight away, which is only sound because a rank word keeps BOTH
// versions (rank_word / rank_at): another workgroup of the same launch that gathers the word as its key2 still
// reads the rank the suffix had when the round began (a mix of old and new ranks inside one group of key2
// values would mis-order its readers).  A group of <= TAIL_G members moves a member by less than TAIL_G
// places, so the new rank is the old one plus a 6-bit count.  Survivors are compacted into the block's OTHER
// list buffer (ballots inside the tile; a tile claims its room in the new list with one atomic add -- the order
// of the groups in a list does not matter, only that a group's members are adjacent), so the next round touches
// only what is left.
constexpr int TR_THREADS = 256, TR_PER = 8, TR_W = TR_THREADS * TR_PER, TR_T = TR_W - 2 * TAIL_G;
static_assert(TAIL_G <= 64, "the rank word keeps a 6-bit displacement");

struct TailArgs {
    const uint32_t *n;   // [B]
    const uint32_t *len; // [B] slot count of the block's list this round
    u64 *buf0, *buf1;    // [B][S] the two list buffers: a block's survivors go, compacted, to the one st_tdst names;
    const uint32_t *tdst; // [B]   the round finds the block's list in the other one (read only)
    uint32_t *rank;      // [B][S]
    uint32_t *c_tail;    // [B] survivors of the round
    uint32_t *c_prog;    // [B] "a group was refined"
    uint32_t tag;        // this round's id in the rank words
    uint32_t dbg;        // builds with -DBZH_EXPERIMENTS only (BZH_TAIL_DBG: 1 = no ranking loop, 2 = no rank stores, 4 = no key gather; wrong results)
    uint32_t *err;       // [1] precondition violations
    const uint32_t *hb;  // [B] depth h of each block
    uint32_t S, T;
    Lst lst;
    const uint32_t *nplain; // the host launched no plain form this round: *nplain (blocks listed for it) must be 0
};

// list record: [rank:20 @40][0:20][suffix:20 @0]
template <bool QUAD>
__global__ void __launch_bounds__(TR_THREADS) tail_round(TailArgs a)
{
    if (QUAD && a.nplain && blockIdx.x == 0 && threadIdx.x == 0 && *a.nplain != 0u) atomicOr(a.err, 4u); // (loud, not wrong bytes)
    uint32_t b, tile;
    if (!wg_map(a.T, a.lst, b, tile)) return;
    const uint32_t len = a.len[b];
    const uint32_t r0 = tile * TR_T;
    launch_check(a.T, a.lst, tile, (len + TR_T - 1) / TR_T, a.err);
    if (r0 >= len) return;
    const uint32_t r1 = min(len, r0 + (uint32_t)TR_T);
    const uint32_t s_lo = r0 >= (uint32_t)TAIL_G ? r0 - TAIL_G : 0u;
    const uint32_t s_hi = min(len, r1 + (uint32_t)TAIL_G);
    const uint32_t nwin = s_hi - s_lo; // <= TR_W
    const uint32_t n = a.n[b], h = a.hb[b], tag = a.tag;
    const size_t base = (size_t)b * a.S;
    const uint32_t td = a.tdst[b];
    const u64 *src = (td ? a.buf0 : a.buf1) + base;
    u64 *dst = (td ? a.buf1 : a.buf0) + base;
    uint32_t *rank = a.rank + base;
    constexpr int NWV = TR_THREADS / 64, ROWS = TR_W / 64;
    typedef typename std::conditional<QUAD, u64, uint32_t>::type key_t; // the 4h form packs three 20-bit ranks
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // K: key of every window slot; HM: per 64 slots, which of them start a group; OUT: the records in their new order
    __shared__ key_t K[TR_W];
    __shared__ u64 OUT[TR_W];
    __shared__ u64 HM[ROWS];
    __shared__ uint32_t wc[TR_PER * NWV], wsurv[NWV];
    __shared__ uint32_t tpre;

    // slot w = k * TR_THREADS + thread: coalesced loads, and the lanes of a wavefront hold 64 consecutive slots
    // (the ranking loop below runs as long as the largest group among them, not among 4 times as many)
    uint32_t ci[TR_PER], cr[TR_PER];
    {
        u64 x[TR_PER], xp[TR_PER];
#pragma unroll
        for (int k = 0; k < TR_PER; k++) {
            const uint32_t w = k * TR_THREADS + threadIdx.x;
            x[k] = w < nwin ? __builtin_nontemporal_load(src + s_lo + w) : LIST_INVALID; // (read once: the L2 is for the rank array)
            xp[k] = 0; // the record before my wavefront's 64 slots (lane 0 compares with it)
            if (lane == 0 && w > 0 && w <= nwin) xp[k] = src[s_lo + w - 1];
        }
#pragma unroll
        for (int k = 0; k < TR_PER; k++) {
            const uint32_t w = k * TR_THREADS + threadIdx.x;
            ci[k] = (uint32_t)(x[k] & SUF_MASK);
            cr[k] = w < nwin ? (uint32_t)(x[k] >> 40) & 0xFFFFFu : 0xFFFFFFFFu;
            key_t key = 0;
            if (w < nwin) {
                const uint32_t i = ci[k];
                if (h < n) {
                    uint32_t i2 = i + h;
                    if (i2 >= n) i2 -= n;
                    const uint32_t k2 = BZH_DBG(a.dbg & 4u) ? i2 * 2654435761u >> 12 : rank_at(rank[rslot(i2)], tag);
                    key = (key_t)k2;
                    if (QUAD) { // two more h-blocks of the (cyclic) rotation
                        uint32_t i3 = i2 + h;
                        if (i3 >= n) i3 -= n;
                        uint32_t i4 = i3 + h;
                        if (i4 >= n) i4 -= n;
                        const uint32_t k3 = rank_at(rank[rslot(i3)], tag), k4 = rank_at(rank[rslot(i4)], tag);
                        key = (key_t)(((u64)k2 << 40) | ((u64)k3 << 20) | k4);
                    }
                } else {
                    key = (key_t)(n - 1 - i); // identical rotations: larger index first
                    if (QUAD) key = (key_t)((u64)(n - 1 - i) << 40);
                }
            }
            K[w] = key;
            // group structure: slot w starts a group if its rank differs from slot w-1's; the end of the window counts
            uint32_t pr = (uint32_t)__shfl_up((int)cr[k], 1, 64);
            if (lane == 0) pr = (uint32_t)(xp[k] >> 40) & 0xFFFFFu;
            const bool head = w <= nwin && (w == 0 || w == nwin || pr != cr[k]);
            const u64 hm = __ballot(head);
            if (lane == 0) HM[k * NWV + wave] = hm;
        }
    }
    __syncthreads();
    // Ranking: every member counts the members of its group that sort before it.
    uint32_t res[TR_PER]; // [owned:1 @31][single:1 @30][less:6 @24][dest slot:12 @0]
    bool moved = false, bad = false;
    uint32_t nsurv = 0;
#pragma unroll
    for (int k = 0; k < TR_PER; k++) {
        const uint32_t w = k * TR_THREADS + threadIdx.x;
        res[k] = w; // slots without a record stay where they are
        if (w < nwin) {
            const uint32_t row = k * NWV + wave;
            const u64 own = HM[row];
            const u64 upto = (2ull << lane) - 1ull; // bits 0..lane (lane 63: all)
            const u64 below = own & upto, above = own & ~upto;
            uint32_t g, ge;
            if (below) {
                g = row * 64u + 63u - (uint32_t)__clzll((long long)below);
            } else { // the group began in the 64 slots before (row > 0: slot 0 starts a group)
                const u64 pm = HM[row - 1];
                g = pm ? (row - 1) * 64u + 63u - (uint32_t)__clzll((long long)pm) : (row - 1) * 64u;
            }
            if (above) {
                ge = row * 64u + (uint32_t)__ffsll((long long)above) - 1u;
            } else {
                const u64 nm = row + 1 < (uint32_t)ROWS ? HM[row + 1] : 0ull;
                ge = nm ? (row + 1) * 64u + (uint32_t)__ffsll((long long)nm) - 1u : min(nwin, (row + 2) * 64u);
            }
            const uint32_t fs = s_lo + g;
            const bool owned = fs >= r0 && fs < r1;
            if (ge - g > (uint32_t)TAIL_G) { // beyond the window guarantee (never for an owned group)
                bad |= owned;
                ge = g + TAIL_G;
            }
            const key_t my = K[w];
            uint32_t less = 0, eq = 0, eqb = 0;
            if (BZH_DBG(a.dbg & 1u)) ge = g;
#pragma unroll 4
            for (uint32_t f = g; f < ge; f++) { // bounds known up front: the LDS reads pipeline
                const key_t kf = K[f];
                less += kf < my;
                eq += kf == my;
                eqb += kf == my && f < w;
            }
            const bool single = eq == 1u;
            if (owned) {
                moved |= less != 0u;
                // The last column is emitted from the ranks, so nothing but the rank word is stored per suffix, and
                // only if the rank moved: nobody reads the "resolved" bit of a block in SPLIT mode (the SA-order
                // enumeration of SWEEP mode is its one reader), and a random 4-byte store is the most expensive
                // thing this kernel does (it leaves the XCD as a partial 64-byte write).
                if (less && !BZH_DBG(a.dbg & 2u)) rank[rslot(ci[k])] = rank_word(cr[k], less, tag, single);
                nsurv += single ? 0u : 1u;
            }
            res[k] = (owned ? 0x80000000u : 0u) | (single ? 0x40000000u : 0u) | (less << 24) | (g + less + eqb);
        }
        // the u-th smallest member takes the slot of the u-th member; resolved and foreign records leave as "none"
        u64 rec = LIST_INVALID;
        if (w < nwin && (res[k] >> 30) == 2u) // owned, not single
            rec = ((u64)(cr[k] + ((res[k] >> 24) & 63u)) << 40) | ci[k];
        OUT[res[k] & 0xFFFu] = rec;
    }
    nsurv = wave_reduce_add(nsurv);
    if (lane == 0) wsurv[wave] = nsurv;
    if (__ballot(moved) && lane == 0) a.c_prog[b] = 1u; // same value from everyone
    if (bad) atomicOr(a.err, 1u);
    __syncthreads();
    // The order of the GROUPS in a small-group list is arbitrary (only a group's members must be adjacent), so a tile
    // just claims room behind whatever is there: no tile ever waits for another.  c_tail ends up as the block's
    // survivor count (round_begin cleared it).  The atomic's round trip runs beside the ballots below.
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) tot += wsurv[w];
        tpre = tot ? atomicAdd(&a.c_tail[b], tot) : 0u;
    }
    u64 x[TR_PER];
    uint32_t lo[TR_PER]; // survivors of my wavefront's row k in lower lanes
#pragma unroll
    for (int k = 0; k < TR_PER; k++) {
        x[k] = OUT[k * TR_THREADS + threadIdx.x];
        const u64 m = __ballot(x[k] != LIST_INVALID);
        lo[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0) wc[k * NWV + wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    // exclusive scan of the row counts in slot order (every wavefront for itself: 32 values)
    const uint32_t c = lane < (uint32_t)(TR_PER * NWV) ? wc[lane] : 0u;
    const uint32_t ex = wave_incl_add(c, (int)lane) - c;
    const uint32_t pre = tpre;
#pragma unroll
    for (int k = 0; k < TR_PER; k++) {
        const uint32_t off = (uint32_t)__shfl((int)ex, k * NWV + (int)wave, 64);
        if (x[k] != LIST_INVALID) __builtin_nontemporal_store(x[k], dst + pre + off + lo[k]);
    }
}

// ---- last column ---------------------------------------------------------------------------------
// Every suffix is resolved, so rank[i] is the position of rotation i in the sorted order: the last column is
// a scatter of the text, bwt[rank[i]] = S[i-1] -- coalesced reads of the ranks and the text, one-byte stores
// into the block's 0.9 MB output (resident in the XCD's L2); ptr = rank[0]; has_byte straight from the text.
// The rotations the bucket-first initial sort resolved (over half of a text block's) have their bytes in place: chunk_finish
// wrote them in sorted order, next to each other; their rank words say so and they are skipped here.
__global__ void __launch_bounds__(256) bwt_emit(Batch bt, uint32_t T, uint32_t B)
{
    uint32_t b, tile;
    const Lst all{nullptr, nullptr, B};
    if (!wg_map(T, all, b, tile)) return;
    T &= ~WG_SPREAD;
    const uint32_t n = bt.n[b];
    const size_t base = (size_t)b * bt.S;
    const uint8_t *s = bt.rle + base;
    const uint32_t *rank = bt.rank + base;
    uint8_t *out = bt.bwt + base;
    __shared__ uint32_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i0 = (tile * 256 + threadIdx.x) * 4; i0 < n; i0 += T * 256 * 4) {
        // bytes S[i0-1 .. i0+2]: the predecessors of suffixes i0 .. i0+3
        uint32_t r[4];
        uint8_t c[4];
        const uint32_t m = min(4u, n - i0);
        if (m == 4) { // (4 | i0, and rslot keeps the low five bits: the four slots are adjacent)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 rv = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rank + rslot(i0))); // (read once: the L2 is for the output)
            r[0] = rv.x;
            r[1] = rv.y;
            r[2] = rv.z;
            r[3] = rv.w;
        } else {
            for (uint32_t k = 0; k < m; k++) r[k] = rank[rslot(i0 + k)];
        }
        c[0] = s[i0 ? i0 - 1 : n - 1];
        for (uint32_t k = 1; k < m; k++) c[k] = s[i0 + k - 1];
        for (uint32_t k = 0; k < m; k++) {
            // (a rotation chunk_finish resolved has its byte in place already: bwt_msd.h, RANK_EMITTED)
            if ((r[k] >> 20) != ((RANK_RESOLVED | RANK_EMITTED) >> 20)) out[rank_final(r[k])] = c[k];
            seen[c[k]] = 1; // every byte of S appears exactly once in the last column
        }
        if (i0 == 0) bt.ptr[b] = rank_final(r[0]);
    }
    __syncthreads();
    if (seen[threadIdx.x]) bt.hasbyte[(size_t)b * 256 + threadIdx.x] = 1;
}

// ---- near-periodic blocks --------------------------------------------------------------------------------------
// A block that is a word w repeated, cut off inside a repetition (S = w^k w', |w| = p, 0 < |w'| = r < p: a tile
// laid over and over, "abab...a", what RLE1 leaves of one enormous run), keeps prefix doubling busy for log2(n)
// rounds over nearly all its suffixes: rotations i and i + p agree for n - i - p characters.  But their order is
// known.  S[u] = w[u mod p] for every u < n, so two members x < y of a group with x = y (mod p) agree until y
// wraps around (offset n - y); from there x reads w from phase r and y from phase 0, which differ at the same
// offset d0 < p for EVERY such pair (w is primitive, r != 0): either every such x sorts before its y or every one
// after it.  A rotation a of the last p - 1 that has wrapped inside the group's depth (n - a <= h) can also sit in
// the group of the phase it reads after the wrap; it then behaves like index a - n (it is the one that keeps
// reading).  So if every group of the block is such a chain -- checked over adjacent members in SA order -- the
// block is finished in ONE round keyed on the (effective) index, ascending or descending.
// One workgroup per SWEEP-mode block, every round until it fires: candidate period = smallest positive member of
// suffix 0's group (at most the true period; a smaller candidate is no period and fails the check), verified
// against the text byte by byte; direction from w; chain check; the tails that lead the order (effective index
// below 0) and the others go, ascending, into two lists for GEN_SWEEP.  Exactly periodic blocks (r = 0) are left
// to the "nothing was refined" rule of round_begin.
struct ProbeArgs {
    const uint8_t *blk;
    const uint32_t *n;
    const uint32_t *rank, *sa, *headp;
    uint32_t *st_h;
    uint32_t *chain; // [B][4]
    uint32_t *clist; // [B][2S]
    uint8_t *tflag;  // [B][S] scratch: tail rotation leads the order
    uint32_t S;
    Lst lst;
};

__global__ void __launch_bounds__(1024) period_probe(ProbeArgs a)
{
    const uint32_t kb = blockIdx.x;
    if (kb >= (a.lst.ids ? *a.lst.cnt : a.lst.B)) return;
    const uint32_t b = a.lst.ids ? a.lst.ids[kb] : kb;
    const uint32_t n = a.n[b], h = a.st_h[b];
    const size_t base = (size_t)b * a.S;
    const uint8_t *s = a.blk + base;
    const uint32_t *sa = a.sa + base, *headp = a.headp + base;
    uint32_t *cl = a.clist + base * 2;
    uint8_t *tf = a.tflag + base;
    __shared__ uint32_t sh_min, sh_bad, sh_d0;
    __shared__ uint32_t ls[1024 / 64 + 2];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        a.chain[b * 4] = 0;
        sh_min = 0xFFFFFFFFu;
        sh_bad = 0;
        sh_d0 = 0xFFFFFFFFu;
    }
    __syncthreads();
    if (h >= H_DONE || n < 16) return;
    const uint32_t w0 = a.rank[base + rslot(0)];
    if (w0 & RANK_RESOLVED) return;
    const uint32_t head = w0 & 0xFFFFFu;
    // 1. candidate: the smallest positive member of suffix 0's group (SA positions head .. while the head stays)
    {
        uint32_t mn = 0xFFFFFFFFu;
        for (uint32_t e = head + tid; e < n && headp[e] == head; e += 1024) {
            const uint32_t i = sa[e];
            if (i > 0) mn = min(mn, i);
        }
        if (mn != 0xFFFFFFFFu) atomicMin(&sh_min, mn);
    }
    __syncthreads();
    const uint32_t p = sh_min;
    if (p == 0xFFFFFFFFu || 2ull * p > n) return;
    const uint32_t r = n % p;
    if (r == 0) return;
    // 2. is p a period of the text?  (S[u] == S[u + p] for every u < n - p)  16 bytes a step (the block's base is 16-byte
    //    aligned; the second operand is read unaligned), four independent steps in flight: one workgroup per block walks
    //    0.9 MB here, and a byte per dependent load was most of the kernel's 1.4 ms
    {
        bool bad = false;
        const uint32_t lim = n - p;
        for (uint32_t u0 = tid * 16u; u0 < lim && !bad; u0 += 1024u * 16u * 4u) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t u = u0 + (uint32_t)q * 1024u * 16u;
                if (u + 16u <= lim) {
                    uint4 x, y;
                    __builtin_memcpy(&x, s + u, 16);
                    __builtin_memcpy(&y, s + u + p, 16);
                    bad |= x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
                } else {
                    for (uint32_t v = u; v < lim; v++) bad |= s[v] != s[v + p];
                }
            }
        }
        if (bad) sh_bad = 1;
    }
    __syncthreads();
    if (sh_bad) return;
    // 3. direction: first d with w[(r + d) mod p] != w[d]
    {
        for (uint32_t d = tid; d < p; d += 1024) {
            uint32_t x = r + d;
            if (x >= p) x -= p;
            if (s[x] != s[d]) {
                atomicMin(&sh_d0, d);
                break;
            }
        }
    }
    // flags of the tail rotations n-p+1 .. n-1 (slot a - (n-p+1)): tf[k] = leads the order, tf[p + k] = has a plainly
    // congruent neighbour (2p <= n <= S)
    for (uint32_t k = tid * 16u; k < 2 * p; k += 1024u * 16u) { // (tf = the block's flag bytes: 16-byte aligned, 2p <= n <= S)
        if (k + 16u <= 2 * p)
            *reinterpret_cast<uint4 *>(tf + k) = make_uint4(0u, 0u, 0u, 0u);
        else
            for (uint32_t v = k; v < 2 * p; v++) tf[v] = 0;
    }
    __syncthreads();
    const uint32_t d0 = sh_d0;
    if (d0 == 0xFFFFFFFFu) return; // (cannot happen for a minimal period with r != 0)
    const uint32_t xr = r + d0 >= p ? r + d0 - p : r + d0;
    const bool asc = s[xr] < s[d0]; // the rotation that keeps reading (smaller index) is the smaller one
    // 4. every group a chain?  adjacent members x, y of a group: x = y (mod p), or one of them a wrapped tail whose
    //    effective index (a - n) is congruent to the other
    const uint32_t tail0 = n - p + 1;
    {
        bool bad = false;
        // x mod p without a division: p is fixed for the block (x < 2^20, p >= 1: floor(x * ceil(2^32 / p) / 2^32) is the
        // quotient or one less... exact here because x * p < 2^40 keeps the error below one: corrected by one compare)
        const uint32_t pinv = (uint32_t)((0x100000000ull + p - 1u) / p);
        auto modp = [&](uint32_t x) -> uint32_t {
            uint32_t q = __umulhi(x, pinv);
            uint32_t rem = x - q * p;       // q may be one too large: then rem wrapped
            if ((int32_t)rem < 0) rem += p;
            if (rem >= p) rem -= p;
            return rem;
        };
        for (uint32_t e0 = 1 + tid; e0 < n && !bad; e0 += 1024u * 8u) {
            uint32_t hx[8], hy[8], sx[8], sy[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { // eight independent steps: the loads of all of them are in flight together
                const uint32_t e = e0 + (uint32_t)q * 1024u;
                const bool in = e < n;
                hy[q] = in ? headp[e] : 0u;
                hx[q] = in ? headp[e - 1] : 1u;
                sx[q] = in ? sa[e - 1] : 0u;
                sy[q] = in ? sa[e] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (hy[q] != hx[q]) continue; // (also the steps past the end)
                const uint32_t x = sx[q], y = sy[q];
                const uint32_t mx = modp(x), my = modp(y);
                if (mx == my) {
                    if (x >= tail0) tf[p + x - tail0] = 1;
                    if (y >= tail0) tf[p + y - tail0] = 1;
                    continue;
                }
                // x - n = x + (p - r) (mod p)
                const bool xt = x >= tail0 && n - x <= h, yt = y >= tail0 && n - y <= h;
                uint32_t ex = mx + p - r, ey = my + p - r;
                if (ex >= p) ex -= p;
                if (ey >= p) ey -= p;
                if (xt && !(y >= tail0) && ex == my) {
                    tf[x - tail0] = 1;
                } else if (yt && !(x >= tail0) && ey == mx) {
                    tf[y - tail0] = 1;
                } else {
                    bad = true;
                }
            }
        }
        if (bad) sh_bad = 1;
    }
    __syncthreads();
    if (sh_bad) return;
    // a leading tail must not ALSO have a plainly congruent neighbour (its group would mix two phases through it)
    {
        bool bad = false;
        for (uint32_t k = tid; k < p - 1 && !bad; k += 1024) bad = tf[k] && tf[p + k];
        if (bad) sh_bad = 1;
    }
    __syncthreads();
    if (sh_bad) return;
    // 5. the tails in two ascending lists: leaders at cl[0 ..), the others at cl[p ..)
    uint32_t c2 = 0, c1 = 0;
    for (uint32_t k0 = 0; k0 < p - 1; k0 += 1024) {
        const uint32_t k = k0 + tid;
        const bool valid = k < p - 1;
        const bool lead = valid && tf[k] != 0;
        uint32_t t2, t1;
        const uint32_t o2 = block_excl_add(lead ? 1u : 0u, ls, &t2);
        const uint32_t o1 = block_excl_add(valid && !lead ? 1u : 0u, ls, &t1);
        if (lead) cl[c2 + o2] = tail0 + k;
        if (valid && !lead) cl[p + c1 + o1] = tail0 + k;
        c2 += t2;
        c1 += t1;
    }
    if (tid == 0) {
        a.chain[b * 4 + 1] = p;
        a.chain[b * 4 + 2] = c2;
        a.chain[b * 4] = 1u | (asc ? 2u : 0u);
        a.st_h[b] = H_DONE;
    }
}

// ---- near-periodic blocks, from the start: sort eight periods, expand -------------------------------------------------
// period_probe above finishes a block S = w^k w' (|w| = p, 0 < |w'| = r < p) in one round -- AFTER the 8-pass sort of all its
// n rotations and a SWEEP round over all of them: 3.5 of the 4.3 ms a batch of such blocks takes.  But the order of the
// rotations of S follows from the order of the rotations of the short block S' = w^m w' (its first n' = m p + r bytes: the
// text needs no copy), m = 8:
//   * rotation i' of S' and rotation i' + D of S (D = n - n' = (k - m) p) are the same distance from the end of the block,
//     where the phase jumps from r to 0 -- the one place a rotation of S differs from w repeated for ever; every comparison
//     between two rotations is decided within fewer than 3 p characters behind the seam of the one that reaches it first, so
//     rotations that are both within n' of the end compare in S as they do in S';
//   * the k - m rotations c, c + p, .. of phase c that S has and S' has not are further from the end than any of them: they
//     read w from phase c for more than n' characters, as rotation c of S' does, and sort next to it -- before it in ascending
//     index order if a rotation that keeps reading (phase r) sorts before one that has wrapped (phase 0), else behind it in
//     descending order -- with nothing in between (a rotation that compares equal to w^inf from phase c for that long is a
//     rotation of phase c, or a tail that reads like one after its wrap and lies beyond all of them).
// (Checked against a naive sort of all rotations for 36,000 random (w, k, r) over alphabets of 2-4 letters, periods up to 30
// and m = 3, 4, 8: no mismatch from m = 3 on -- /tests/test_period_model.py keeps a smaller run of that model; on the GPU
// test_near_periodic_from_the_start compares whole streams with the oracle.)
// period_detect (one workgroup a block, before anything else reads bt.n): the smallest p <= PD_PMAX with S[u] = S[u + p] for
// all u < n - p -- candidates are the places where the block's first 16 bytes recur, verified in ascending order --; if the
// block is at least PD_M + 2 periods long and r != 0 it leaves (flags, p, n, m) in bt.pshrink and SHRINKS bt.n[b] to n': every
// kernel of the sort then sees a block of a few periods.  period_expand (behind bwt_emit) writes the last column and the
// origin pointer of S from the ranks of S' and restores bt.n[b].  Exactly periodic blocks (r = 0) are left to round_begin's
// "nothing was refined" rule, as before.
constexpr uint32_t PD_PMAX = 8192, PD_M = 8, PD_MINLEN = 64;

__global__ void __launch_bounds__(1024) period_detect(const uint8_t *blk, uint32_t *nn, uint32_t *pshrink, uint32_t S)
{
    const uint32_t b = blockIdx.x, tid = threadIdx.x, n = nn[b];
    const uint8_t *s = blk + (size_t)b * S;
    __shared__ uint32_t s_cand, s_d;
    uint32_t *rec = pshrink + (size_t)b * 4;
    if (tid < 4) rec[tid] = 0u;
    if (n < 20u * 16u) return; // (too short to be worth it; also keeps the 16-byte probes inside the block)
    const uint32_t pmax = min(PD_PMAX, n / (PD_M + 2u));
    u64 f0, f1;
    __builtin_memcpy(&f0, s, 8);
    __builtin_memcpy(&f1, s + 8, 8);
    uint32_t lower = 1;
    for (int attempt = 0; attempt < 4; attempt++) {
        if (tid == 0) s_cand = 0xFFFFFFFFu;
        __syncthreads();
        uint32_t mine = 0xFFFFFFFFu;
        for (uint32_t p = lower + tid; p <= pmax && mine == 0xFFFFFFFFu; p += 1024) {
            u64 g0, g1;
            __builtin_memcpy(&g0, s + p, 8);
            __builtin_memcpy(&g1, s + p + 8, 8);
            if (g0 == f0 && g1 == f1) mine = p;
        }
        if (mine != 0xFFFFFFFFu) atomicMin(&s_cand, mine);
        __syncthreads();
        const uint32_t p = s_cand;
        if (p == 0xFFFFFFFFu) return; // the first 16 bytes do not recur: no period up to pmax
        // every u < n - p: 16 bytes a thread and step, four steps between two looks at the verdict
        const uint32_t lim = n - p;
        bool failed = false;
        for (uint32_t u0 = 0; u0 < lim; u0 += 4u * 16384u) {
            bool bad = false;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t u = u0 + q * 16384u + tid * 16u;
                if (u + 16u <= lim) {
                    u64 a0, a1, c0, c1;
                    __builtin_memcpy(&a0, s + u, 8);
                    __builtin_memcpy(&a1, s + u + 8, 8);
                    __builtin_memcpy(&c0, s + u + p, 8);
                    __builtin_memcpy(&c1, s + u + p + 8, 8);
                    bad |= a0 != c0 || a1 != c1;
                } else {
                    for (uint32_t v = u; v < lim; v++) bad |= s[v] != s[v + p];
                }
            }
            if (__syncthreads_or(bad ? 1 : 0)) { // (the same verdict in every thread: the loop is left by all of them together)
                failed = true;
                break;
            }
        }
        if (!failed) {
            const uint32_t k = n / p, r = n - k * p;
            const uint32_t m = max(PD_M, (PD_MINLEN + p - 1u) / p);
            if (r == 0u || k < m + 2u) return;
            // which way a phase's rotations run: does a rotation that keeps reading (w from phase r) sort before one that has
            // wrapped (w from phase 0)?  First difference of the two, within p characters (w is primitive, r != 0)
            if (tid == 0) s_d = 0xFFFFFFFFu;
            __syncthreads();
            uint32_t dmin = 0xFFFFFFFFu;
            for (uint32_t d = tid; d < p && dmin == 0xFFFFFFFFu; d += 1024) {
                uint32_t x = r + d;
                if (x >= p) x -= p;
                if (s[x] != s[d]) dmin = d;
            }
            if (dmin != 0xFFFFFFFFu) atomicMin(&s_d, dmin);
            __syncthreads();
            const uint32_t d = s_d;
            if (d == 0xFFFFFFFFu) return; // (cannot happen for a minimal period; the block is then left to the general sort)
            if (tid == 0) {
                uint32_t x = r + d;
                if (x >= p) x -= p;
                const bool asc = s[x] < s[d];
                rec[1] = p;
                rec[2] = n;
                rec[3] = m;
                rec[0] = 1u | (asc ? 2u : 0u);
                nn[b] = m * p + r;
            }
            return;
        }
        lower = p + 1u; // that recurrence was no period: the next one
        __syncthreads();
    }
}

constexpr uint32_t PX_WGS = 8; // workgroups that share the expansion of one block
__global__ void __launch_bounds__(1024) period_expand(Batch bt, const uint32_t *pshrink)
{
    const uint32_t b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const uint32_t *rec = pshrink + (size_t)b * 4;
    if (!(rec[0] & 1u)) return;
    const bool asc = (rec[0] & 2u) != 0u;
    const uint32_t p = rec[1], n = rec[2], m = rec[3];
    const uint32_t k = n / p, r = n - k * p, np = m * p + r, X = k - m; // X: rotations of every phase that S has and S' has not
    const size_t base = (size_t)b * bt.S;
    const uint8_t *s = bt.rle + base;
    const uint32_t *rank = bt.rank + base;
    uint8_t *out = bt.bwt + base;
    constexpr uint32_t MW = (PD_M * PD_PMAX + 2u * PD_PMAX + PD_MINLEN) / 32u + 8u; // words of a bit per position of S' (n' < (m + 1) p)
    __shared__ uint32_t mask[MW], pre[MW];
    __shared__ uint32_t ls[20];
    const uint32_t words = (np + 1u + 31u) / 32u;
    for (uint32_t w = tid; w < words; w += 1024) mask[w] = 0u;
    __syncthreads();
    // where the extra rotations of every phase go: in front of rotation c of S' (ascending) or behind it (descending)
    for (uint32_t c = tid; c < p; c += 1024) {
        const uint32_t q = rank_final(rank[rslot(c)]) + (asc ? 0u : 1u);
        atomicOr(&mask[q >> 5], 1u << (q & 31u));
    }
    __syncthreads();
    // pre[w] = insertion points in the words before w
    uint32_t carry = 0;
    for (uint32_t w0 = 0; w0 < words; w0 += 1024) {
        const uint32_t w = w0 + tid;
        const uint32_t c = w < words ? (uint32_t)__popc(mask[w]) : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_add(c, ls, &tot);
        if (w < words) pre[w] = carry + ex;
        carry += tot;
    }
    __syncthreads();
    auto upto = [&](uint32_t q) { return pre[q >> 5] + (uint32_t)__popc(mask[q >> 5] & (0xFFFFFFFFu >> (31u - (q & 31u)))); }; // points at positions <= q
    // the rotations S' has: rotation i' of S' is rotation i' + D of S, behind every run of extras whose point is at or before it
    for (uint32_t i = part * 1024u + tid; i < np; i += PX_WGS * 1024u) {
        const uint32_t q = rank_final(rank[rslot(i)]);
        out[q + X * upto(q)] = i ? s[i - 1u] : s[p - 1u]; // (rotation D of S is preceded by the end of a period, not by S'[n' - 1])
    }
    // the extras of every phase: X copies of the phase's predecessor byte -- rotation 0 of S, the furthest of phase 0, is
    // preceded by the block's last byte and is where the origin pointer points.  Short periods: the workgroups share every
    // phase's run; long ones: a wavefront a phase.
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (X >= 4096u) {
        for (uint32_t c = 0; c < p; c++) {
            const uint32_t q = rank_final(rank[rslot(c)]) + (asc ? 0u : 1u);
            const uint32_t at = q + X * (upto(q) - 1u);
            const uint8_t ch = c ? s[c - 1u] : s[p - 1u];
            for (uint32_t j = part * 1024u + tid; j < X; j += PX_WGS * 1024u)
                if (!(c == 0u && j == (asc ? 0u : X - 1u))) out[at + j] = ch; // (phase 0's one exception is written below)
        }
    } else {
        for (uint32_t c = part * 16u + wave; c < p; c += PX_WGS * 16u) {
            const uint32_t q = rank_final(rank[rslot(c)]) + (asc ? 0u : 1u);
            const uint32_t at = q + X * (upto(q) - 1u);
            const uint8_t ch = c ? s[c - 1u] : s[p - 1u];
            for (uint32_t j = lane; j < X; j += 64)
                if (!(c == 0u && j == (asc ? 0u : X - 1u))) out[at + j] = ch;
        }
    }
    if (part == 0 && tid == 0) {
        const uint32_t q = rank_final(rank[rslot(0)]) + (asc ? 0u : 1u);
        const uint32_t at = q + X * (upto(q) - 1u) + (asc ? 0u : X - 1u);
        out[at] = s[n - 1u];
        bt.ptr[b] = at;
        bt.n[b] = n;
    }
}

// ---- round bookkeeping on the device ---------------------------------------------------------------
// One workgroup, one thread per block.  Consumes the counters the previous round left (c_big, c_small,
// c_tail, c_prog; after the initial refine: `first`), decides every block's mode and depth, builds this
// round's work lists and gates, clears the counters and the digit totals of the blocks on the big-list
// path, and writes the summary the host reads one round late.
// summary words: 0 round, 1 nS, 2 nA, 3 nT, 4 nQ, 5 maxS, 6 maxA, 7 maxT, 8 total unresolved, 9 of them sitting the round out,
//                10 sum of the S lists, 11 this round sorts its big lists on ranks (five passes; else on group numbers: four), 12 sum of the A lists, 13 (round 0) members of small groups that entered the first doubling step inside chunk_finish, 14 error flag, 15 largest depth in use,
//                16 S blocks whose refine writes lists this round (the only ones that can leave SWEEP mode next round),
//                17/18 sum over the rounds so far of the unresolved suffixes entering them, 19 the part of word 12 that mid_sort orders,
//                20-22 unused, 23 sequence word (last)
#ifndef QUAD_DIV
#define QUAD_DIV 10u
#endif
constexpr uint32_t QUAD_BIT = 0x80000000u; // in st_ntail-derived gates: this round runs at depth x4
enum ListId : int { L_S = 0, L_A = 1, L_R = 2, L_T = 3, L_Q = 4, L_P = 5 };

__global__ void __launch_bounds__(1024) round_begin(Batch bt, uint32_t B, uint32_t round, uint32_t *actP, uint32_t *hsum, uint32_t seq, uint32_t sweep_div,
                                                    uint32_t r0_fused, uint32_t mid_on)
{
    const uint32_t b = threadIdx.x;
    const bool valid = b < B;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (valid) bt.gcount[b] = 0u; // numbers are drawn afresh for the lists this round's refinement writes
    if (threadIdx.x == 0) bt.gwide[(round + 1u) & 1u] = 0u; // ... and so is that round's "a block ran out of numbers"
    // per-wavefront partial results: sums (all, S, A, n, sconv), maxima (S, A, T, depth), list counts
    __shared__ uint32_t psum[7][16], pmax[4][16], pcnt[6][16];
    uint32_t gS = 0, gA = 0, gT = 0, h = 0, n = 0, conv = 0;
    bool sit = false; // the block's small groups sit this round out (below)
    if (valid) {
        // everything this thread needs, loaded at once
        n = bt.n[b];
        const uint32_t groups = bt.c_groups[b], mode_in = bt.st_mode[b], h_in = bt.st_h[b];
        const uint32_t nbig = bt.c_big[b], ntail = bt.c_tail[b] + bt.c_small[b];
        const uint32_t gR = bt.gateR[b], gTin = bt.gateT[b], prog = bt.c_prog[b], nolist = bt.c_nolist[b];
        // Round 0: with fewer than one group per `sweep_div` (256) suffixes after the 8-byte sort the block is run-heavy /
        // periodic and starts in SWEEP mode; text-like blocks start in SPLIT mode and never pay for SA order by position.
        // (a block that took the bucket-first initial sort -- bwt_msd.h -- has no SA order by position: SPLIT mode, and
        // its rotations are ordered by their first 7 bytes, not 8)
        const bool msd = round == 0 && bt.ms_np[b] != 0u;
        uint32_t mode = round == 0 ? ((!msd && (uint64_t)groups * sweep_div < n) ? 0u : 1u) : mode_in;
        h = round == 0 ? (msd ? 7u : 8u) : h_in; // the initial sort ordered the rotations by their first 8 (7) bytes
        // A bucket-first block whose small groups chunk_finish has already ordered by bytes 7..14 of their rotations
        // (bwt_msd.h: the first doubling step, from the text, inside the kernel that found the groups): those groups
        // stand at depth 15, only the large groups need round 0 (depth 7 -> 14); the small-group list stays where it is
        // and joins in round 1, whose depth is 14 whatever round 0 did or did not refine (mixed depths are sound: every
        // rank is a refinement of the order by 7 bytes that agrees with the final order, and a group is only ever
        // keyed at a depth its members are known to share).
        const bool fusedblk = r0_fused != 0u && bt.ms_np[b] != 0u;
        sit = round == 0 && fusedblk;
        if (round == 1 && fusedblk) {
            h = h_in << 1;
        } else
        if (round > 0 && (gR | gTin)) { // the block had work in the round before
            const bool wasquad = (gTin & QUAD_BIT) != 0;
            if (!prog && h < n)
                h = H_DONE; // nothing was refined: equal ranks at depth h are equal at every depth (block = w^k)
            else if (h < H_DONE)
                h = (h << (wasquad ? 2 : 1)) > H_DONE ? H_DONE : (h << (wasquad ? 2 : 1));
        }
        // SWEEP mode pays a sweep of all n positions + 3 passes, the big-list path ~6 passes over the large
        // groups only: leave SWEEP mode, for good, once those hold less than a third of the block
        if (mode == 0u && (uint64_t)nbig * 3u < n && !nolist) mode = 1u;
        if (mode == 0u) {
            gS = nbig + ntail;
            conv = (gS && (uint64_t)nbig * 2u < n) ? 1u : 0u; // (refine's `nolist` rule, negated)
        } else {
            gA = nbig;
            gT = ntail;
        }
        bt.st_mode[b] = mode;
        bt.st_h[b] = h;
        bt.st_nbig[b] = nbig;
        bt.st_ntail[b] = ntail;
        bt.gateS[b] = gS;
        bt.gateA[b] = gA;
        bt.gateR[b] = gS | gA; // one of them is 0
        bt.c_big[b] = 0;
        bt.c_small[b] = 0;
        // where the block's small-group records of this round go: tail_round reads the list from the buffer the round
        // before left it in and writes the survivors to the other one, refine appends behind them; a block without a
        // tail_round this round keeps its list (and its length: refine appends behind c_tail) where it is
        const uint32_t loc = round == 0 ? 0u : bt.st_tdst[b];
        const bool runs = gT != 0u && !sit;
        bt.st_tdst[b] = runs ? loc ^ 1u : loc;
        bt.c_tail[b] = sit ? gT : 0u;
        bt.c_prog[b] = 0;
        bt.c_nolist[b] = 0;
    }
    const uint32_t gTl = sit ? 0u : gT; // what this round's small-group kernels see
    // the records of the big lists that mid_sort orders in LDS (bwt_msd.h) instead of the global passes
    const uint32_t gAm = (valid && mid_on && ms_block_is_mid(bt.ms_np, msc_row(bt.ms_cnt, B, MSR_SPANS), b)) ? gA : 0u;
    if (valid && mid_on) msc_row(bt.ms_cnt, B, MSR_RUNQ + ((round + 1u) & 1u))[b] = 0u; // the runs this round's refinement writes: the next round's list
    { // (every sum stays below 2^30: at most 1024 blocks of fewer than 2^20 suffixes)
        const uint32_t v[7] = {gS + gA + gT, gS, gA, n, conv, sit ? gT : 0u, gAm};
        const uint32_t m[4] = {gS, gA, gT, (gS | gA | gT) ? h : 0u};
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const uint32_t r = wave_reduce_add(v[k]);
            if (lane == 0) psum[k][wave] = r;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t r = m[k];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) r = max(r, (uint32_t)__shfl_xor((int)r, d, 64));
            if (lane == 0) pmax[k][wave] = r;
        }
    }
    __syncthreads();
    uint32_t sum[7] = {0, 0, 0, 0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
    for (uint32_t w = 0; w < nw; w++) {
#pragma unroll
        for (int k = 0; k < 7; k++) sum[k] += psum[k][w];
#pragma unroll
        for (int k = 0; k < 4; k++) mx[k] = max(mx[k], pmax[k][w]);
    }
    // a block with small groups only may look three h-blocks ahead (depth 4h, three gathers per suffix):
    // worth it once few suffixes are left -- in the block, or in the whole batch -- when rounds are latency-bound
    const bool few = (uint64_t)gT * QUAD_DIV < n || (uint64_t)sum[0] * QUAD_DIV < sum[3];
    // (h == H_DONE -- identical rotations, keyed on the index -- takes the depth x4 kernel as well: it keys them the same way,
    // and once every block with small groups is on it the host stops launching the plain form at all)
    const uint32_t quad = (valid && gA == 0u && gTl != 0u && few && (h < (1u << 28) || h == H_DONE)) ? 1u : 0u;
    if (valid) bt.gateT[b] = gTl | (quad ? QUAD_BIT : 0u);
    // order-preserving lists: position = listed blocks in lower lanes + in earlier wavefronts
    // (list 3 only counts: blocks that HOLD small groups, sitting out or not -- the host bounds the next round with it)
    const bool fl[6] = {gS != 0u, gA != 0u, (gS | gA) != 0u, gT != 0u, gTl != 0u && quad, gTl != 0u && !quad};
    uint32_t *dst[6] = {bt.actS, bt.actA, bt.actR, bt.actT, bt.actQ, actP};
    uint32_t pre[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const u64 m = __ballot(fl[k]);
        pre[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0) pcnt[k][wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t tot[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t off = 0, t = 0;
        for (uint32_t w = 0; w < nw; w++) {
            const uint32_t c = pcnt[k][w];
            if (w < wave) off += c;
            t += c;
        }
        tot[k] = t;
        if (fl[k]) dst[k][off + pre[k]] = b;
    }
    if (wave == 0) { // the summary, one word per lane, straight into pinned host memory
        // unresolved suffixes entering the rounds so far.  The first doubling step of a bucket-first block's small groups
        // runs inside chunk_finish (depth 7 -> 15): the members that ENTERED it count for round 0 (chunk_finish's counter),
        // not the survivors that sit round 0 out.
        const unsigned long long asum = *bt.stat_A + sum[0] + ((round == 0 && r0_fused) ? (unsigned long long)bt.ms_cnt[6] - sum[5] : 0ull);
        const uint32_t words[SUMMARY_WORDS] = {round,  tot[L_S], tot[L_A], tot[L_T], tot[L_Q], mx[0],  mx[1], mx[2], sum[0], sum[5],
                                               sum[1], bt.gwide[round & 1u], sum[2],   (round == 0 && r0_fused) ? bt.ms_cnt[6] : 0u, *bt.errflag, mx[3], sum[4], (uint32_t)asum,
                                               (uint32_t)(asum >> 32), sum[6], 0u, 0u, 0u, 0u};
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < SUMMARY_WORDS - 1; k++) mine = lane == (uint32_t)k ? words[k] : mine;
        if (lane < (uint32_t)SUMMARY_WORDS - 1) hsum[lane] = mine;
        __threadfence_system(); // every lane's word is out before lane 0 says so (the release below orders lane 0's own stores)
        if (lane < 6) bt.nlist[lane] = lane == 0 ? tot[0] : lane == 1 ? tot[1] : lane == 2 ? tot[2] : lane == 3 ? tot[3] : lane == 4 ? tot[4] : tot[5];
        if (lane == 0) {
            *bt.stat_A = asum;
            // the record is complete before its last word says so
            __hip_atomic_store(hsum + SUMMARY_WORDS - 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- one launch clears everything a sort starts from ---------------------------------------------------------------
// (a hipMemsetAsync is a kernel of its own, about 5 us each back to back: eight of them were 45 us of every step)
constexpr int CLEAR_MAX = 16;
struct ClearArgs {
    uint4 *p[CLEAR_MAX];         // 16-byte aligned regions (carved at 256-byte boundaries: layout_batch, api.hip -- a length
                                 // rounded up to 16 bytes stays inside the region's own carve)
    unsigned long long q[CLEAR_MAX]; // their lengths in 16-byte words, as a running total (region k = [q[k-1], q[k]))
    int n;
};
__global__ void __launch_bounds__(256) bwt_clear(ClearArgs c)
{
    const unsigned long long total = c.q[c.n - 1];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256) {
        int k = 0;
        while (i >= c.q[k]) k++;
        c.p[k][i - (k ? c.q[k - 1] : 0ull)] = make_uint4(0u, 0u, 0u, 0u);
    }
}
struct ClearList {
    ClearArgs a{};
    bool overflow = false; // a region that found no room: the caller must not launch (state would stay uncleared)
    void add(void *p, size_t bytes)
    {
        if (!bytes) return;
        if (a.n >= CLEAR_MAX) {
            overflow = true;
            return;
        }
        a.p[a.n] = reinterpret_cast<uint4 *>(p);
        a.q[a.n] = (a.n ? a.q[a.n - 1] : 0ull) + (bytes + 15) / 16; // (regions end at 256-byte boundaries of the arena: rounding up stays inside)
        a.n++;
    }
    void launch(hipStream_t st)
    {
        if (!a.n) return;
        const unsigned long long total = a.q[a.n - 1];
        const unsigned grid = (unsigned)std::min<unsigned long long>(4096ull, (total + 255) / 256);
        bwt_clear<<<dim3(grid), 256, 0, st>>>(a);
    }
};

#include "bwt_msd.h"

// ---- host driver -----------------------------------------------------------------------------------
// one radix pass = one kernel (look-back scatter)
template <int BITS, int MODE>
static void launch_pass(bzh_ctx *ctx, SortArgs &a, uint32_t NB, uint32_t maxcnt)
{
    const uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0 || NB == 0) return;
    a.T = tiles | ((few_blocks(NB) && !ctx->no_spread) ? WG_SPREAD : 0u); // (look-back: see bwt_run on no_spread)
    a.pass++;
    radix_scatter<BITS, MODE><<<dim3(xcd_grid(a.T, NB)), SORT_THREADS, 0, ctx->stream>>>(a);
    if (ctx->profiling) ctx->stats.bwt_sort_launches++; // every launch issued, also the ones that find their list empty
}

// Profiling: HIP events bracket each RUN of consecutive radix_scatter launches (the 8 initial passes,
// the 3 of a SWEEP round, the 5 of a big-list round -- nothing else runs in between), not every
// launch: an event pair costs about as much idle time as a small kernel.
static hipEvent_t span_begin(bzh_ctx *ctx)
{
    if (!ctx->profiling) return nullptr;
    hipEvent_t e = bzh_event(ctx);
    hipEventRecord(e, ctx->stream);
    return e;
}
static void span_end(bzh_ctx *ctx, hipEvent_t e0)
{
    if (!ctx->profiling || !e0) return;
    hipEvent_t e1 = bzh_event(ctx);
    hipEventRecord(e1, ctx->stream);
    ctx->sort_spans.push_back({e0, e1});
}

// One workgroup per listed block: column sums of refine's digit rows, then the exclusive scan inside each
// of the three digits -> dbase[b][k*128 + d] = first list slot of digit d in SWEEP pass k.
__global__ void __launch_bounds__(768) sweep_bases(RefineArgs a, uint32_t *dbase)
{
    const uint32_t kb = blockIdx.x;
    if (kb >= (a.lst.ids ? *a.lst.cnt : a.lst.B)) return;
    const uint32_t b = a.lst.ids ? a.lst.ids[kb] : kb;
    if (a.mode[b] != 0u) return; // rows are only written for blocks in SWEEP mode
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    const uint32_t col = threadIdx.x % 384, seg = threadIdx.x / 384;
    const uint32_t half = (ntile + 1) / 2;
    const uint32_t t0 = seg ? half : 0u, t1 = seg ? ntile : half;
    const uint32_t *p = a.dig + (size_t)b * (a.dig_stride ? a.dig_stride : a.TPB * 512u) + col;
    uint32_t sum = 0, t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[(size_t)(t + k) * 512];
#pragma unroll
        for (int k = 0; k < 8; k++) sum += v[k];
    }
    for (; t < t1; t++) sum += p[(size_t)t * 512];
    __shared__ uint32_t part[384];
    __shared__ uint32_t ls[16];
    if (seg) part[col] = sum;
    __syncthreads();
    const uint32_t c = seg ? 0u : sum + part[col];
    uint32_t tot;
    const uint32_t ex = block_excl_add(c, ls, &tot); // threads 0..383 in (digit, value) order, the rest add 0
    if (!seg) part[col] = ex;
    __syncthreads();
    if (!seg) dbase[(size_t)b * DB_STRIDE + col] = ex - part[col & ~127u];
}

static void launch_refine(bzh_ctx *ctx, RefineArgs &r, uint32_t NB, uint32_t maxcnt, bool bases)
{
    const uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0 || NB == 0) return;
    r.T = tiles | ((few_blocks(NB) && !ctx->no_spread) ? WG_SPREAD : 0u);
    flag_tiles<<<dim3(xcd_grid(r.T, NB)), SORT_THREADS, 0, ctx->stream>>>(r);
    flag_carry<<<dim3(NB), 1024, 0, ctx->stream>>>(r);
    refine<<<dim3(xcd_grid(r.T, NB)), SORT_THREADS, 0, ctx->stream>>>(r);
    if (bases) sweep_bases<<<dim3(NB), 768, 0, ctx->stream>>>(r, ctx->bt.dbase);
}

// ---- initial ranks: binned by destination, then applied -------------------------------------------------------
// A 4-byte store to a random slot of a block's 3.6 MB rank array leaves the XCD as a partial 64-byte write: the
// 100 M stores of the initial refinement cost 6.4 GB of fabric writes.  Instead refine_one<init> bins its (rank word,
// suffix) pairs by suffix bits 12..19 on the way out, exactly as a radix pass would (counts per tile, look-back over
// the tiles, runs of a bin leave together; the bins are full by construction, so their bases are known up front):
// the pairs of every 4096-suffix window of the rank array end up together, and rank_apply places a window's words
// in LDS and stores them as whole lines.
constexpr uint32_t APPLY_W = 4096;
__global__ void __launch_bounds__(256) rank_apply(const u64 *binned, const uint32_t *nn, uint32_t *rank, uint32_t S)
{
    const uint32_t b = blockIdx.y, n = nn[b];
    const uint32_t w0 = blockIdx.x * APPLY_W;
    if (w0 >= n) return;
    const uint32_t cnt = min(APPLY_W, n - w0);
    const size_t base = (size_t)b * S;
    __shared__ uint32_t win[APPLY_W];
    constexpr int PER = APPLY_W / 256;
#pragma unroll
    for (int k = 0; k < PER; k++) win[k * 256 + threadIdx.x] = 0xFFFFFFFFu; // (a last window may be short)
    __syncthreads();
    u64 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t e = k * 256 + threadIdx.x;
        v[k] = e < cnt ? binned[base + w0 + e] : ~0ull;
    }
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (v[k] != ~0ull) win[rslot((uint32_t)v[k] & (uint32_t)SUF_MASK) - w0] = (uint32_t)(v[k] >> 32); // rslot stays inside the aligned 1024-block
    __syncthreads();
    uint32_t *dst = rank + base + w0;
#pragma unroll
    for (int k = 0; k < PER / 4; k++) {
        const uint32_t e = (k * 256 + threadIdx.x) * 4;
        const uint4 q = *reinterpret_cast<const uint4 *>(&win[e]);
        if (q.x != 0xFFFFFFFFu && q.y != 0xFFFFFFFFu && q.z != 0xFFFFFFFFu && q.w != 0xFFFFFFFFu) {
            *reinterpret_cast<uint4 *>(dst + e) = q;
        } else {
            if (q.x != 0xFFFFFFFFu) dst[e] = q.x;
            if (q.y != 0xFFFFFFFFu) dst[e + 1] = q.y;
            if (q.z != 0xFFFFFFFFu) dst[e + 2] = q.z;
            if (q.w != 0xFFFFFFFFu) dst[e + 3] = q.w;
        }
    }
}

template <bool INIT>
static void launch_refine_one(bzh_ctx *ctx, RefineArgs &r, uint32_t NB, uint32_t maxcnt, u64 *recs = nullptr)
{
    const uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0 || NB == 0) return;
    r.T = tiles | ((few_blocks(NB) && !ctx->no_spread) ? WG_SPREAD : 0u);
    refine_one<INIT><<<dim3(xcd_grid(r.T, NB)), SORT_THREADS, 0, ctx->stream>>>(r, ctx->bt.c_groups, recs);
}

// Suffix-sort and emit the last column for blocks 0..B-1 of the batch (bt.rle / bt.n filled).
// nmax = largest block length in the batch, ntotal = sum of block lengths (statistics only).
int bwt_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal, bool is_retry)
{
    Batch &bt = ctx->bt;
    if (B == 0) return BZH_OK;
    hipStream_t st = ctx->stream;
    const uint32_t mb = ctx->max_batch;
    // (what a retry must not count twice: see the look-back retry at the end)
    const bzh_stats stats_in = ctx->stats;
    const size_t spans_in = ctx->sort_spans.size(), kspans_in = ctx->kspans.size();
    uint64_t kb_in[K_COUNT], kl_in[K_COUNT];
    memcpy(kb_in, ctx->k_bytes, sizeof kb_in);
    memcpy(kl_in, ctx->k_launch, sizeof kl_in);
    u64 *bufA = reinterpret_cast<u64 *>(bt.listA), *bufB = reinterpret_cast<u64 *>(bt.listB);
    u64 *bufC = reinterpret_cast<u64 *>(bt.listC), *bufD = reinterpret_cast<u64 *>(bt.listD);
    const Lst all{nullptr, nullptr, B};
    uint32_t *actP = bt.actQ + bt.B; // sixth list, behind the five named ones (rows of the layout are bt.B apart)

    SortArgs a{};
    a.blk = bt.rle;
    a.n = bt.n;
    a.rank = bt.rank;
    a.sa = bt.sa;
    a.headp = bt.headp;
    a.hb = bt.st_h;
    a.S = bt.S;
    a.TPB = bt.TPB;
    a.gwide = bt.gwide;
    a.gidof = bt.gidof;

    // ---- initial sort on the 8-byte cyclic prefix: LSD, 8 passes of 8 bits.  Passes 0-4 order by bytes 3..7 of
    // the rotation (the element carries all five); pass 5 finds each element's 5-byte group on the way in (the list
    // is in that order), replaces the key by bytes 0..2 + the dense index of the group (one gather from the
    // L2-resident text) and orders by byte 2; passes 6-7 order by bytes 1 and 0.  The refinement that follows
    // compares whole elements.  A plain pass over every suffix costs far less than a doubling round does per
    // suffix, so this replaces "4-byte sort + refine + first doubling round".
    // The passes run as single look-back kernels (no histogram / scan launches): their digit bases
    // are the block's byte counts.
    a.cnt = bt.n;
    a.lst = all;
    a.shift = 20;
    a.gst = reinterpret_cast<u64 *>(bt.tagg); // (flag_tiles / flag_carry use it after the initial sort only)
    a.chain = bt.chain;
    a.fault = ctx->debug_fault; // (one batch only)
    a.patient = ctx->no_spread ? 1u : 0u;
    const bool had_fault = a.fault == 1u; // (kind 2: the fault of a shared GPU -- the sort is expected to recover by itself)
    ctx->debug_fault = 0;
    a.clist = reinterpret_cast<const uint32_t *>(bt.listD); // (a block in SWEEP mode has no small-group lists)
    a.src = nullptr;
    a.dst = bufA;
    a.look = reinterpret_cast<u64 *>(bt.hist);
    a.dbase = bt.dbase;
    a.doff = 0;
    a.err = bt.errflag;
    a.pass = 0;
    ClearList clr; // launched below, once it is known whether the bucket tables are used
    clr.add(bt.errflag, sizeof(uint32_t));
    clr.add(a.look, (size_t)B * bt.TPB * NBMAX * sizeof(u64));
    clr.add(bt.tagg, (size_t)B * bt.TPB * sizeof(int4));
    clr.add(bt.dtot, (size_t)B * DB_STRIDE * sizeof(uint32_t));
    // round state: everything from st_mode to the 64-bit counter (one contiguous carve, see layout_batch)
    clr.add(bt.st_mode, (size_t)((uint8_t *)(bt.stat_A + 1) - (uint8_t *)bt.st_mode));
    clr.add(bt.hasbyte, (size_t)B * 256); // (bwt_emit ORs into it at the very end)
    clr.add(bt.gcount, (size_t)bt.B * sizeof(uint32_t)); // (no large group numbered yet; the two "ran out of numbers" words
    clr.add(bt.gwide, 16);                                //  lie behind it)
    // The second stream of the suffix sort (created once): the big-list path of a round runs on it beside the small-group
    // kernel, and so do the 8 passes of the blocks that keep them when the rest of the batch takes the bucket-first
    // initial sort.  (With profiling on everything stays on one stream, or the per-kernel spans would overlap.)
    static const bool no_overlap = getenv("BZH_NO_OVERLAP") != nullptr;
    hipStream_t side = nullptr;
    if (!ctx->profiling && !no_overlap) {
        side = bzh_side_stream(ctx);
        if (side && !ctx->side2_stream) { // (optional: without it the global passes follow mid_sort on the second stream)
            if (hipStreamCreateWithFlags(&ctx->side2_stream, hipStreamNonBlocking) != hipSuccess) ctx->side2_stream = nullptr;
            if (ctx->side2_stream && hipEventCreateWithFlags(&ctx->side_ev[2], hipEventDisableTiming) != hipSuccess) {
                hipStreamDestroy(ctx->side2_stream);
                ctx->side2_stream = nullptr;
            }
        }
    }
    hipStream_t side2 = side ? ctx->side2_stream : nullptr;
    // ---- which blocks take which initial sort (bwt_msd.h): text-like blocks the bucket-first one (the plan decides per
    // block, on the device), repetitive, random and binary blocks the 8 passes below (`oldl`; all blocks at level 1,
    // whose blocks are too short for the tables to pay).  BZH_INIT=lsd keeps every block on the 8 passes (A/B timing).
    static const bool init_lsd = []() {
        const char *e = getenv("BZH_INIT");
        return e && !strcmp(e, "lsd");
    }();
    static const bool init_msd = []() {
        const char *e = getenv("BZH_INIT");
        return e && !strcmp(e, "msd");
    }();
    // (a batch of fewer than 12 blocks keeps the 8 passes: the bucket tables' fixed work -- histogram, plan, five levels --
    // does not pay below that: 1 / 2 / 4 / 8 / 16 text blocks 1.56 / 1.62 / 2.08 / 2.55 / 3.07 ms with the buckets, 1.45 / 1.51 /
    // 1.94 / 2.47 / 3.11 ms with the 8 passes, one random block 1.36 / 1.26 ms; BZH_INIT=msd overrides)
    const bool use_msd = ctx->M >= MS_MIN_N && !a.fault && !init_lsd && (B >= 12u || init_msd);
    // chunk_finish takes the first doubling step of the small groups itself, keyed on the text (BZH_R0=0: A/B timing)
    static const bool r0_off = getenv("BZH_R0") && !strcmp(getenv("BZH_R0"), "0");
    const uint32_t r0_fused = (use_msd && !r0_off) ? 1u : 0u;
    // round 0 orders the large groups of a bucket-first block unit by unit in LDS (mid_sort; BZH_MID=0: the global passes, A/B timing)
    static const bool mid_off = getenv("BZH_MID") && !strcmp(getenv("BZH_MID"), "0");
    const bool mid_on = use_msd && !mid_off;
    bool msd_deeper = false; // the levels behind the first ran: a block may hold a group that spans several units
    uint32_t nOld = B;
    Lst oldl = all;
    u64 *const binned = reinterpret_cast<u64 *>(bt.binned);
    volatile uint32_t *const hrec0 = ctx->h_pinned + (size_t)mb * 8 + 64; // [MAX_ROUNDS + 1][SUMMARY_WORDS]
    const uint32_t epoch = (++ctx->bwt_epoch & 0xFFFFFFu) << 6;
    Msd msd_keep{};
    if (use_msd) { // (its counters, rank-window cursors and bigram counts join the one clearing launch)
        clr.add(bt.ms_cnt, (MS_CNT_WORDS + (size_t)(MS_LEVELS + 7) * B) * sizeof(uint32_t)); // (the counters; the tables of runs and tiles behind them are written before they are read)
        clr.add(bt.ms_bincur, (size_t)B * 256 * sizeof(uint32_t));
        clr.add(bt.ms_bgcur, (size_t)B * MS_BG * sizeof(uint32_t));
    } else {
        clr.add(bt.ms_np, (size_t)B * sizeof(uint32_t));
    }
    if (clr.overflow) { // (more regions than the clearing kernel takes: sort state would stay dirty -- loud, not wrong bytes)
        bzh_set_error(ctx, "BWT: the clearing list is full (internal error)");
        return BZH_E_HIP;
    }
    clr.launch(st);
    // near-periodic blocks are sorted as eight of their periods (period_detect shrinks bt.n[b] before anything reads it,
    // period_expand behind bwt_emit writes the whole block's last column); a retry after a look-back gave up finds the blocks
    // shrunk already
    if (!is_retry) period_detect<<<dim3(B), 1024, 0, st>>>(bt.rle, bt.n, bt.pshrink, bt.S);
    if (use_msd) {
        BZH_TRY(msd_sort_begin(ctx, B, nmax, ntotal, bufB, bufD, bufA, bufC, binned, false, r0_fused,
                               hrec0 + (size_t)MAX_ROUNDS * SUMMARY_WORDS, epoch + 63u, &nOld, side ? ctx->side_ev[0] : nullptr, &msd_keep));
        oldl = Lst{bt.ms_old, bt.ms_cnt + MC_OLD, B};
        a.lst = oldl;
    }
    const uint64_t ntotal_old = nOld == B ? ntotal : ntotal * nOld / B; // (statistics only)
    u64 *cur = bufA, *oth = bufB;
    hipEvent_t ev_init = span_begin(ctx);
    // A mixed batch: the few blocks on the 8 passes cannot fill the device (one block alone takes ~3 ms for them), so
    // they run on the second stream beside the bucket-first kernels of the others, from the plan's end on.
    const bool old_beside = use_msd && side && nOld && nOld < B;
    hipStream_t so = st;
    if (old_beside) {
        hipStreamWaitEvent(side, ctx->side_ev[0], 0);
        so = side;
        ctx->stream = side; // (launch_pass / launch_refine_one launch on the context's stream)
    }
    if (nOld) {
        {
            KSpan ks(ctx, K_BYTE_COUNT, ntotal_old, 2);
            byte_count<<<dim3(BYTE_SEGS, B), 1024, 0, so>>>(bt.rle, bt.n, bt.dtot, bt.S);
            active_bases<<<dim3(B), 256, 0, so>>>(bt.dtot, bt.dbase, all, 1);
        }
        {
            KSpan ks(ctx, K_RADIX_INIT, 13 * ntotal_old); // 5 text bytes in, one element out
            launch_pass<8, GEN_BYTES5>(ctx, a, nOld, nmax);
        }
        for (int p = 1; p < 8; p++) {
            a.shift = p < 5 ? 20 + 8 * p : 40 + 8 * (p - 5);
            a.src = cur;
            a.dst = oth;
            KSpan ks(ctx, p == 5 ? K_RADIX_GID : K_RADIX_INIT, (p == 5 ? 20 : 16) * ntotal_old); // (re-key: + one 4-byte gather)
            if (p == 5)
                launch_pass<8, GEN_GID>(ctx, a, nOld, nmax);
            else
                launch_pass<8, GEN_LIST>(ctx, a, nOld, nmax);
            u64 *t = cur;
            cur = oth;
            oth = t;
        }
        if (ctx->profiling) ctx->stats.bwt_sort_elems += 8 * ntotal_old;
    } else {
        cur = bufB; // (where the 8 passes leave the sorted list; the big lists are in bufA either way)
        oth = bufA;
    }
    span_end(ctx, ev_init);

    RefineArgs r{};
    r.n = bt.n;
    r.cnt = bt.n;
    r.list = cur;
    r.big = oth;
    r.tail0 = bufC;
    r.tail1 = bufD;
    r.tdst = bt.st_tdst;
    r.tbase = bt.c_tail;
    r.mode = bt.st_mode;
    r.blk = bt.rle;
    r.bwt = bt.bwt;
    r.rank = bt.rank;
    r.sa = bt.sa;
    r.headp = bt.headp;
    r.flg = bt.flg;
    r.tagg = bt.tagg;
    r.dig = bt.hist;
    r.c_big = bt.c_big;
    r.c_small = bt.c_small;
    r.c_prog = bt.c_prog;
    r.c_nolist = bt.c_nolist;
    r.nbig_in = bt.st_nbig;
    r.S = bt.S;
    r.TPB = bt.TPB;
    r.gout = GidOut{bt.gidof, bt.grank, bt.gcount, bt.gwide, bt.S, bt.B, 0u}; // (the initial refinement writes round 0's lists)
    r.grank = bt.grank;
    r.gwide = bt.gwide;
    r.init = 1;
    r.cstat = reinterpret_cast<u64 *>(bt.hist);
    r.carry = reinterpret_cast<u64 *>(bt.tagg);
    r.cpass = ++a.pass;
    r.err = bt.errflag;
    r.patient = a.patient;
    r.lst = oldl;
    // lists of every block + (rank word, suffix) pairs in list order; the blocks that start in SWEEP mode get their
    // SA order and digit bases in round 0 (below)
    {
        // list in, one (rank word, suffix) pair out per suffix -- binned by 4096-suffix window of the rank array, into
        // sa|headp, which nobody needs before round 0 -- plus the list records of the unresolved ones (their bytes are
        // added when round 0's summary is in); the blocks that start in SWEEP mode get SA order and digit bases in
        // round 0 (below).  rank_apply then writes the rank array as whole lines.
        r.bpass = ++a.pass;
        if (nOld) {
            KSpan ks(ctx, K_REFINE_INIT, 24 * ntotal_old);
            r.dig = reinterpret_cast<uint32_t *>(bt.flg); // (the digit rows of the initial refinement: the flag bytes are free, `hist` holds its look-back words)
            r.dig_stride = bt.S / 4;
            launch_refine_one<true>(ctx, r, nOld, nmax, binned);
            r.dig = bt.hist;
            r.dig_stride = 0;
        }
        if (old_beside) {
            ctx->stream = st;
            hipEventRecord(ctx->side_ev[1], side);
        }
        if (use_msd) { // the rest of the bucket-first sort: deeper levels if level 1 left any, the finishing kernel
            BZH_TRY(msd_sort_finish(ctx, st, msd_keep, ntotal, hrec0 + (size_t)MAX_ROUNDS * SUMMARY_WORDS, epoch + 63u, &msd_deeper));
        static const bool trace_init = getenv("BZH_TRACE_ROUNDS") != nullptr;
        if (trace_init) { // (debugging aid: waits for the device)
            uint32_t c[MS_CNT_WORDS];
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(c, bt.ms_cnt, sizeof c, hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[bzhip] initial sort: %u blocks bucket-first, %u blocks 8-pass; %u units; %.1f %% of the suffixes in oversized 2-byte buckets; oversized buckets per level %u %u %u %u %u (tiles %u %u %u %u %u)\n",
                        c[MC_NEW], c[MC_OLD], c[MC_UNITS], 100.0 * 1024.0 * c[23] / (double)std::max<uint64_t>(1, ntotal), c[MC_SEGS + 1], c[MC_SEGS + 2], c[MC_SEGS + 3], c[MC_SEGS + 4], c[MC_SEGS + 5],
                        c[MC_ITEMS + 1], c[MC_ITEMS + 2], c[MC_ITEMS + 3], c[MC_ITEMS + 4], c[MC_ITEMS + 5]);
            if (c[32] | c[35])
                fprintf(stderr, "[bzhip] chunk_finish, 16-cycle ticks over all workgroups: ticket+load+bucket index %u, ranking %u, stage scatter+barrier %u, reload %u, heads+suffix table %u, suffixes+extents+bins %u, keys+bin scan %u, all pairs %u, rank pairs out %u, lists out %u\n",
                        c[32], c[33], c[34], c[35], c[36], c[37], c[38], c[39], c[40], c[41]);
        }
        }
        if (old_beside) hipStreamWaitEvent(st, ctx->side_ev[1], 0); // the main stream goes on only behind the second stream's work
        KSpan ks(ctx, K_RANK_APPLY, 12 * ntotal);
        rank_apply<<<dim3((nmax + APPLY_W - 1) / APPLY_W, B), 256, 0, st>>>(binned, bt.n, bt.rank, bt.S);
    }
    { // the big lists of the first round are in `oth`
        u64 *t = cur;
        cur = oth;
        oth = t;
    }
    r.init = 0;
    r.cnt = bt.gateR;
    if (mid_on) {
        r.mid_np = bt.ms_np;
        r.mid_spans = msc_row(bt.ms_cnt, B, MSR_SPANS);
        r.mid_tiles = msc_tiles(bt.ms_cnt, B);
        r.mid_ntiles = msc_row(bt.ms_cnt, B, MSR_MTILES);
    }
    // (greedy packing: two neighbouring tiles hold more than SORT_TILE records, so a list of L records makes at most
    // 2 L / SORT_TILE + 1 tiles -- the bound refine_one's launch is sized with when such blocks exist)
    auto refine_bound = [&](uint32_t maxcnt) { return mid_on ? (2u * ((maxcnt + SORT_TILE - 1) / SORT_TILE) + 1u) * SORT_TILE : maxcnt; };

    TailArgs ta{};
    ta.n = bt.n;
    ta.len = bt.st_ntail; // (gateT carries the QUAD bit; the plain length lives here)
    ta.buf0 = bufC; // a block's small-group list moves between bufC and bufD, one hop per round in which it is worked on
    ta.buf1 = bufD; // (round_begin keeps track per block: st_tdst)
    ta.tdst = bt.st_tdst;
    ta.rank = bt.rank;
    ta.c_tail = bt.c_tail;
    ta.c_prog = bt.c_prog;
    ta.err = bt.errflag;
    ta.hb = bt.st_h;
    ta.S = bt.S;
#ifdef BZH_EXPERIMENTS
    ta.dbg = getenv("BZH_TAIL_DBG") ? (uint32_t)atoi(getenv("BZH_TAIL_DBG")) : 0u;
#endif

    // ---- doubling rounds, queued one ahead of the summaries ---------------------------------------------
    // round_begin writes its summary straight into pinned host memory and sets the record's last word to
    // round + 1 behind a system-scope release: no copy, no event -- the host looks at the word.
    // (The word also carries the number of this call: the round_begin queued last by the call before may still
    // be on its way when this one starts.)
    volatile uint32_t *hsum = hrec0; // [MAX_ROUNDS][SUMMARY_WORDS] (+ one record of the initial sort's plan)
    auto wait_summary = [&](uint32_t rd, uint32_t *out) -> hipError_t { // normally there already
        volatile uint32_t *rec = hsum + (size_t)rd * SUMMARY_WORDS;
        const uint32_t want = epoch + rd + 1u;
        int idle = 0;
        for (uint64_t it = 0; rec[SUMMARY_WORDS - 1] != want; it++) {
            if ((it & 0xFFFu) == 0xFFFu) { // a kernel fault would leave us here for ever: ask the stream now and then
                const hipError_t e = hipStreamQuery(st);
                if (e != hipSuccess && e != hipErrorNotReady) return e;
                if (e == hipSuccess && ++idle > 64) return hipErrorUnknown; // stream drained, still no record
            }
            __builtin_ia32_pause();
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int k = 0; k < SUMMARY_WORDS; k++) out[k] = rec[k];
        return hipSuccess;
    };
    static const bool trace = getenv("BZH_TRACE_ROUNDS") != nullptr;
    // A block on the 8 passes with fewer than n / 256 groups after them starts in SWEEP mode (BZH_SWEEP_DIV: A/B timing).
    // Rounds 2-3 drew the line at n / 8; measured in round 4 (scripts/gpu_sweep_ab.py): the periodic and run-heavy inputs
    // SWEEP mode exists for have a few hundred to a thousand times fewer groups than suffixes and do not care (they
    // break at n / 2048), while merely repetitive text is 3-9 % faster in SPLIT mode (real text with every block on the
    // 8 passes 15.85 -> 14.60 ms, with the buckets 14.73 -> 14.36 ms: two of its blocks keep the 8 passes).
    static const uint32_t sweep_div = getenv("BZH_SWEEP_DIV") ? (uint32_t)atoi(getenv("BZH_SWEEP_DIV")) : 256u;
    // bounds for the launches of the round being queued (exact lists live on the device)
    uint32_t nS = B, nA = B, nT = B, nQ = B, maxS = nmax, maxA = nmax, maxT = nmax;
    // do the blocks of mid_sort / the blocks of the global passes have big lists?  (round 0: what the initial sort's plan says;
    // later: the sums of the last summary -- a big list only shrinks, and only a block in SWEEP mode can still join the others)
    bool have_mid = mid_on, have_glob = !mid_on || nOld != 0u || msd_deeper;
    uint32_t err = 0;
    bool finished = false;
    uint32_t s[SUMMARY_WORDS];

    // -- blocks in SWEEP mode: three look-back passes; the last refine left the digit bases (sweep_bases)
    auto run_S = [&](uint32_t round) {
        if (!nS) return;
        KSpan ks(ctx, K_SWEEP, 0, 3);
        if (round == 0) {
            // round_begin has just picked the blocks that start in SWEEP mode: SA order, heads by position and digit rows are
            // there (refine_one<init> leaves them for every block on the 8 passes); the rows become the bases of the passes
            RefineArgs r0 = r;
            r0.cnt = bt.n;
            r0.dig = reinterpret_cast<uint32_t *>(bt.flg);
            r0.dig_stride = bt.S / 4;
            r0.lst = Lst{bt.actS, bt.nlist + L_S, B};
            sweep_bases<<<dim3(nS), 768, 0, st>>>(r0, bt.dbase); // (before period_probe: it uses the flag bytes next)
        }
        {
            // near-periodic blocks are finished in this round (period_probe sets their depth to "h >= n")
            ProbeArgs pr{bt.rle, bt.n, bt.rank, bt.sa, bt.headp, bt.st_h, bt.chain, reinterpret_cast<uint32_t *>(bt.listD),
                         bt.flg, bt.S, Lst{bt.actS, bt.nlist + L_S, B}};
            period_probe<<<dim3(nS), 1024, 0, st>>>(pr);
        }
        a.lst = Lst{bt.actS, bt.nlist + L_S, B};
        a.cnt = bt.n; // enumerate SA positions
        a.shift = 40;
        a.doff = 0;
        a.src = nullptr;
        a.dst = cur;
        hipEvent_t e0 = span_begin(ctx);
        launch_pass<7, GEN_SWEEP>(ctx, a, nS, nmax);
        a.cnt = bt.gateS;
        a.shift = 47;
        a.doff = 128;
        a.src = cur;
        a.dst = oth;
        launch_pass<7, GEN_LIST>(ctx, a, nS, maxS);
        a.shift = 54;
        a.doff = 256;
        a.src = oth;
        a.dst = cur;
        launch_pass<7, GEN_LIST>(ctx, a, nS, maxS);
        span_end(ctx, e0);
    };
    // -- SPLIT-mode blocks with large groups: re-key the big list once, then five look-back passes on bits
    //    20..59; gen: cur -> oth, passes: oth -> cur -> oth -> cur -> oth -> cur
    // The big-list path (re-key + five passes) and the small-group kernel of a round touch different lists, and the
    // one thing they share -- the rank array, read by the first, written by the second -- keeps both versions of a
    // word (rank_at), so the two run side by side: the big-list path on a second stream between two events.  (With
    // profiling on everything stays on one stream, or the per-kernel spans would overlap.)
    bool side_busy = false, side2_busy = false;
    auto join_side = [&]() { // the main stream goes on only behind what the side streams were given
        if (side_busy) hipStreamWaitEvent(st, ctx->side_ev[1], 0);
        if (side2_busy) hipStreamWaitEvent(st, ctx->side_ev[2], 0);
        side_busy = side2_busy = false;
    };
    // An error return hands the context back: nothing of this call may still be running on the second stream
    // against the arena when the next call queues its memsets (bzh_debug_fault promises a usable context).
    auto fail_wait = [&](hipError_t e) -> int {
        if (side_busy && side) hipStreamSynchronize(side);
        if (side2_busy && side2) hipStreamSynchronize(side2);
        side_busy = side2_busy = false;
        ctx->stream = st;
        bzh_set_error(ctx, "%s:%d waiting for a round summary -> %s", __FILE__, __LINE__, hipGetErrorString(e));
        return BZH_E_HIP;
    };
    // The tiles of the NEXT round's big lists, planned as soon as this round's refinement has written them -- on the second
    // stream, beside round_begin: mid_sort can then start with its round (a plan queued in front of it started it late, when
    // tail_round held every CU's LDS, and it waited for room: 374 us for 2 M records).
    auto plan_ahead = [&](uint32_t round) {
        if (!(mid_on && have_mid)) return;
        KSpan ks(ctx, K_MID_SORT, 0);
        Msd mp = msd_keep;
        mp.runq = msc_row(bt.ms_cnt, B, MSR_RUNQ + ((round + 1u) & 1u));
        mp.runs = msc_runs(bt.ms_cnt, B, (round + 1u) & 1u);
        hipStream_t sp = st;
        if (side) {
            hipEventRecord(ctx->side_ev[0], st);
            hipStreamWaitEvent(side, ctx->side_ev[0], 0);
            sp = side;
        }
        mid_plan<<<dim3(B), 256, 0, sp>>>(mp, 0u);
        if (side) { // (the plan is part of what the second stream was given: join_side, fail_wait and the end of the sort wait for it)
            hipEventRecord(ctx->side_ev[1], side);
            side_busy = true;
        }
    };
    auto run_A = [&](uint32_t round) {
        const uint32_t gt = (maxA + SORT_TILE - 1) / SORT_TILE;
        if (!nA || !gt) return;
        hipStream_t sa = st;
        // (round 0 with mid_sort alone -- no block for the global passes, no small-group kernel beside it: it stays on the main
        // stream, a fork and a join between streams are 20 us)
        const bool alone = round == 0 && mid_on && have_mid && !have_glob && r0_fused && nOld == 0;
        const bool forked = side && !alone;
        if (forked) {
            hipEventRecord(ctx->side_ev[0], st);
            hipStreamWaitEvent(side, ctx->side_ev[0], 0);
            sa = side;
            ctx->stream = side; // (launch_pass launches on the context's stream)
        }
        a.lst = Lst{bt.actA, bt.nlist + L_A, B};
        a.cnt = bt.gateA;
        a.src = cur;
        a.dst = oth;
        a.T = gt | (few_blocks(nA) ? WG_SPREAD : 0u);
        // The bucket-first blocks without a group that spans several units: their big lists are ordered tile by tile in LDS, in
        // place (the sorted list is wanted in `cur` either way); the global passes below skip them, and are not even launched
        // when no block can need them (none on the 8 passes or in SWEEP mode, no spanning group: only level 5 of the initial
        // sort makes those) -- nor is mid_sort once no block of its kind has a big list left (the sums of the last summary).
        // Round 0 has the device to itself (the small groups of a bucket-first block sit it out): two workgroups a CU, and
        // the few blocks left to the global passes run them on the main stream; from round 1 on tail_round runs beside it
        // and needs room on every CU: one workgroup a CU.
        const bool mid = mid_on && have_mid;
        if (mid) {
            KSpan ks(ctx, K_MID_SORT, 0); // (the tiles were planned behind the refinement that wrote the lists: plan_ahead)
            MidArgs ma{cur, bt.rank, bt.st_h, bt.gateA, a.tag, msc_tiles(bt.ms_cnt, B), msc_row(bt.ms_cnt, B, MSR_MTILES), msc_row(bt.ms_cnt, B, MSR_SPANS),
                       bt.ms_np, bt.n, msc_row(bt.ms_cnt, B, MSR_MTICKET), bt.errflag, bt.S, B, 0u};
#ifdef BZH_EXPERIMENTS
            ma.dbg = getenv("BZH_MID_DBG") ? (uint32_t)atoi(getenv("BZH_MID_DBG")) : 0u;
#endif
            mid_sort<<<dim3(round ? 256 : 512), MS_THREADS, 0, sa>>>(ma);
        }
        const bool global_path = !mid_on || have_glob;
        a.mid_np = mid_on ? bt.ms_np : nullptr;
        a.mid_spans = mid_on ? msc_row(bt.ms_cnt, B, MSR_SPANS) : nullptr;
        bool on_side2 = false;
        if (mid && forked && round == 0) {
            sa = st;
            ctx->stream = st;
        } else if (mid && forked && side2 && global_path) { // (from round 1 on the main stream is tail_round's: a third stream)
            hipStreamWaitEvent(side2, ctx->side_ev[0], 0);
            sa = side2;
            ctx->stream = side2;
            on_side2 = true;
        }
        if (global_path) {
        // A large group has more than TAIL_G members (and a group that spans several units of the initial sort draws one
        // number a unit: at most two more per 8192 records), so a list of at most 250,000 records cannot run out of
        // GID_MAX numbers: the fifth pass is then not even launched.
        const int npass = maxA <= 250000u ? 4 : 5;
        a.only4 = npass == 4 ? 1u : 0u;
        {
            KSpan ks(ctx, K_ACTIVE_GEN, 0, 2);
            active_gen<<<dim3(xcd_grid(a.T, nA)), SORT_THREADS, 0, sa>>>(a, bt.dtot);
            active_bases<<<dim3(nA), 256, 0, sa>>>(bt.dtot, bt.dbase, a.lst, 5);
        }
        // five passes on [rank][key2] from `oth` (where active_gen put the re-keyed list) -- or, the groups being numbered
        // densely (the usual case, decided on the device: *gwide), four passes on [number][key2] from `cur` (active_gen wrote
        // in place); either way the sorted list ends in `cur`
        u64 *c = oth, *o = cur;
        hipEvent_t e0 = span_begin(ctx);
        KSpan ks(ctx, K_RADIX_ROUNDS, 0, 5);
        a.has_n = 1u;
        // (a large group has more than TAIL_G members: lists of at most GID_MAX * (TAIL_G + 1) records cannot run out of
        // numbers, and the fifth pass is not even launched)
        for (int p = 0; p < npass; p++) {
            a.shift = 20 + 8 * p;
            a.doff = 256 * p;
            a.src = c;
            a.dst = o;
            a.src_n = p < 4 ? o : nullptr; // (the narrow chain runs the other way round: cur -> oth -> cur -> oth -> cur)
            a.dst_n = p < 4 ? c : nullptr;
            launch_pass<8, GEN_LIST>(ctx, a, nA, maxA);
            u64 *t = c;
            c = o;
            o = t;
        }
        a.has_n = 0u;
        span_end(ctx, e0);
        } // (global_path)
        a.mid_np = a.mid_spans = nullptr;
        if (forked) {
            ctx->stream = st;
            hipEventRecord(ctx->side_ev[1], side);
            side_busy = true;
            if (on_side2) {
                hipEventRecord(ctx->side_ev[2], side2);
                side2_busy = true;
            }
        }
    };
    // -- small groups: one kernel per form (depth x2 / depth x4); survivors move to the other list buffer
    // (all_quad: the last summary read shows no block in SWEEP mode, none with a big list and every block that holds small
    // groups on the depth x4 form -- all of which only ever stays so --: the plain form has no block to work on and is not
    // launched; the depth x4 kernel checks the list it would have had)
    bool all_quad = false;
    auto run_T = [&]() {
        const uint32_t tt = (maxT + TR_T - 1) / TR_T;
        if (!nT || !tt) return;
        KSpan ks(ctx, K_TAIL_ROUND, 0, nQ ? 2 : 1);
        ta.T = tt | (few_blocks(nT) ? WG_SPREAD : 0u);
        ta.tag = a.tag;
        ta.nplain = nullptr;
        if (!(all_quad && nQ)) {
            ta.lst = Lst{actP, bt.nlist + L_P, B};
            tail_round<false><<<dim3(xcd_grid(ta.T, nT)), TR_THREADS, 0, st>>>(ta);
        } else {
            ta.nplain = bt.nlist + L_P;
        }
        if (nQ) {
            ta.lst = Lst{bt.actQ, bt.nlist + L_Q, B};
            tail_round<true><<<dim3(xcd_grid(ta.T, nQ)), TR_THREADS, 0, st>>>(ta);
        }
    };
    uint64_t emitted = 0; // rotations whose last-column byte the initial sort wrote (statistics)
    auto account = [&](const uint32_t *sm) { // statistics of the round a summary describes
        if (!ctx->profiling) return;
        const uint64_t tot = (uint64_t)sm[8], eS = (uint64_t)sm[10], eAm = (uint64_t)sm[19], eA = (uint64_t)sm[12] - eAm; // (eAm: mid_sort's records, round 0)
        if (sm[2]) { // the big lists of this round: four passes on group numbers, five on ranks (word 11)
            const uint64_t passes = sm[11] ? 5u : 4u;
            ctx->stats.bwt_sort_elems += passes * eA;
            ctx->k_bytes[K_RADIX_ROUNDS] += eA * passes * 16;
        }
        ctx->k_bytes[K_MID_SORT] += eAm * 20; // record in, ordered record out, one rank gather
        if (sm[1]) ctx->stats.bwt_sort_elems += 3 * eS;
        // algorithmic bytes of the round's kernels (per-element figures: DESIGN.md section 4)
        ctx->k_bytes[K_SWEEP] += eS * (3 * 16 + 12 + 20);           // enumeration + gather, 3 passes, flags + refine
        ctx->k_bytes[K_ACTIVE_GEN] += eA * 20;                      // record in, re-keyed record out, one rank gather
        ctx->k_bytes[K_REFINE_ROUNDS] += (eA + eAm) * 20;           // record in, rank word, list record out
        ctx->k_bytes[K_TAIL_ROUND] += (tot - eS - eA - eAm - sm[9]) * 24; // record in, key gather, rank word, survivor out (not the groups that sit the round out)
        if (sm[0] == 0) {
            // (+ the last-column bytes of the rotations the initial sort resolved -- everything that is in no list now: one
            // byte gathered, one byte stored each; bwt_emit stores the rest)
            emitted = ntotal > tot ? ntotal - tot : 0;
            ctx->k_bytes[nOld == B ? K_REFINE_INIT : K_MSD_FINISH] += emitted * 2;
        }
        if (sm[0] == 0) ctx->k_bytes[nOld == B ? K_REFINE_INIT : K_MSD_FINISH] += tot * 8 + (uint64_t)sm[13] * 8; // the list records the initial refinement wrote + the 8 text bytes each member of a small group was keyed on
    };

    for (uint32_t round = 0; round < (uint32_t)MAX_ROUNDS; round++) {
        a.tag = 1u + round % 31u; // (a block is at work for fewer than 31 rounds: its depth doubles every time)
        // the numbers of the large groups: this round's lists carry the half written by the round before; this round's
        // refinement draws the numbers of the next round's groups into the other half
        a.gwide = r.gwide = bt.gwide + (round & 1u);
        r.grank = bt.grank + (size_t)(round & 1u) * bt.B * GID_MAX;
        r.gout.par = (round + 1u) & 1u;
        if (mid_on) { // (the runs this round's refinement writes are the next round's list)
            r.mid_runq = msc_row(bt.ms_cnt, B, MSR_RUNQ + ((round + 1u) & 1u));
            r.mid_runs = msc_runs(bt.ms_cnt, B, (round + 1u) & 1u);
        }
        {
        KSpan ks(ctx, K_ROUND_BEGIN, 0);
        round_begin<<<1, (B + 63u) / 64u * 64u, 0, st>>>(bt, B, round, actP,
                                                          const_cast<uint32_t *>(hsum) + (size_t)round * SUMMARY_WORDS, epoch + round + 1u,
                                                          sweep_div, r0_fused, mid_on ? 1u : 0u);
        }
        if (round > 1) {
            // the summary of the PREVIOUS round: where every block stood when that round began
            if (const hipError_t we = wait_summary(round - 1, s); we != hipSuccess) return fail_wait(we);
            const uint64_t total = (uint64_t)s[8];
            err |= s[14];
            if (trace)
                fprintf(stderr, "[bzhip] round %u h<=%u unresolved=%llu  S blocks=%u (max %u)  A blocks=%u (max %u, %s)  T blocks=%u (quad %u, max %u)\n",
                        s[0], s[15], (unsigned long long)total, s[1], s[5], s[2], s[6], s[11] ? "on ranks" : "on group numbers", s[3], s[4], s[7]);
            if (err) break; // a kernel reported an internal error: nothing after it can be trusted
            if (total == 0) { // nothing was left when round-1 began: it and this round_begin were no-ops
                finished = true;
                break;
            }
            ctx->stats.bwt_rounds = (uint64_t)round > ctx->stats.bwt_rounds ? (uint64_t)round : ctx->stats.bwt_rounds;
            account(s);
            // One round on: SWEEP blocks can only leave; their unresolved suffixes may turn up in the big or the
            // small lists; big lists shrink, small lists gain at most what the big lists lose.
            // SWEEP blocks only leave that mode with lists in hand: cS of them wrote lists in the round before.
            const uint32_t pS = s[1], pA = s[2], pT = s[3], mS = s[5], mA = s[6], mT = s[7], cS = s[16];
            have_mid = mid_on && s[19] != 0u;
            have_glob = !mid_on || s[12] != s[19] || pS != 0u || cS != 0u;
            nS = pS;
            maxS = mS;
            nA = std::min(B, pA + cS);
            maxA = std::max(mA, cS ? mS : 0u);
            nT = std::min(B, pT + pA + cS);
            nQ = nT;
            maxT = std::min(nmax, mT + std::max(mA, cS ? mS : 0u));
            all_quad = pS == 0u && pA == 0u && cS == 0u && pT != 0u && s[4] == pT;
        }
        if (round == 0) {
            // Nothing is known yet, and text-like batches have no block in SWEEP mode: the big-list and small-group
            // paths go first with full-size launches (they take a millisecond), by then round 0's own summary is
            // there and the SWEEP path runs with exact sizes -- or, mostly, not at all.
            run_A(0);
            if (!(r0_fused && nOld == 0)) run_T(); // (every block on the bucket-first sort: all small groups sit round 0 out -- not even an empty launch)
            if (const hipError_t we = wait_summary(0, s); we != hipSuccess) return fail_wait(we);
            const uint64_t total = (uint64_t)s[8];
            err |= s[14];
            if (trace)
                fprintf(stderr, "[bzhip] round %u h<=%u unresolved=%llu  S blocks=%u (max %u)  A blocks=%u (max %u, %s)  T blocks=%u (quad %u, max %u)\n",
                        s[0], s[15], (unsigned long long)total, s[1], s[5], s[2], s[6], s[11] ? "on ranks" : "on group numbers", s[3], s[4], s[7]);
            if (total == 0) { // the initial sort resolved everything
                finished = true;
                break;
            }
            ctx->stats.bwt_rounds = std::max<uint64_t>(ctx->stats.bwt_rounds, 1);
            account(s);
            nS = s[1];
            maxS = s[5];
            run_S(0);
            // exact figures of round 0 from here on (they also bound round 1, as above)
            const uint32_t pA = s[2], pT = s[3], mA = s[6], mT = s[7], cS = s[16];
            nA = pA;
            maxA = mA;
            have_mid = mid_on && s[19] != 0u;
            have_glob = !mid_on || s[12] != s[19] || s[1] != 0u || cS != 0u;
            join_side();
            if (nS | nA) {
                r.list = cur;
                r.big = oth;
                if (nS) {
                    r.cpass = ++a.pass;
                    r.lst = Lst{bt.actS, bt.nlist + L_S, B};
                    KSpan ks(ctx, K_SWEEP, 0, 4);
                    launch_refine(ctx, r, nS, maxS, true);
                }
                if (nA) {
                    r.cpass = ++a.pass;
                    r.lst = Lst{bt.actA, bt.nlist + L_A, B};
                    KSpan ks(ctx, K_REFINE_ROUNDS, 0);
                    launch_refine_one<false>(ctx, r, nA, refine_bound(maxA));
                    plan_ahead(0);
                }
                std::swap(cur, oth);
            }
            // bounds of round 1
            nA = std::min(B, pA + cS);
            maxA = std::max(mA, cS ? maxS : 0u);
            nT = std::min(B, pT + pA + cS);
            nQ = nT;
            maxT = std::min(nmax, mT + std::max(mA, cS ? maxS : 0u));
            continue;
        }
        run_S(round);
        run_A(round);
        run_T();
        join_side();
        // -- every block that went through radix passes: flags, group extents, ranks, routing (SWEEP-mode blocks in
        //    three kernels with SA order and digit counts, the big lists of SPLIT-mode blocks in one)
        if (nS | nA) {
            r.list = cur;
            r.big = oth;
            if (nS) {
                r.cpass = ++a.pass;
                r.lst = Lst{bt.actS, bt.nlist + L_S, B};
                KSpan ks2(ctx, K_SWEEP, 0, 4);
                launch_refine(ctx, r, nS, maxS, true);
            }
            if (nA) {
                r.cpass = ++a.pass;
                r.lst = Lst{bt.actA, bt.nlist + L_A, B};
                KSpan ks(ctx, K_REFINE_ROUNDS, 0);
                launch_refine_one<false>(ctx, r, nA, refine_bound(maxA));
                plan_ahead(round);
            }
            std::swap(cur, oth);
        }
    }
    join_side();
    if (!finished && !err) { // MAX_ROUNDS is far beyond log2(n) + the rounds queued ahead
        HIP_TRY(ctx, bzh_stream_wait(st));
        bzh_set_error(ctx, "BWT: the doubling rounds did not terminate (internal error)");
        return BZH_E_HIP;
    }
    if (err) HIP_TRY(ctx, bzh_stream_wait(st)); // (what was queued behind the faulty kernel ends before the error is reported)
    // A look-back that gave up although nothing was injected: seen only when several PROCESSES compute on this GPU at once
    // and a launch covers few blocks (few_blocks: fewer than 6; fewer than 32 when this was seen).  Their tiles are dealt over
    // all XCDs, so a tile may wait for a predecessor that is still queued on another XCD -- whose slots another process's workgroups hold, waiting in the same
    // way for tiles queued behind ours.  Blocks pinned to one XCD each (the mapping of larger batches) only ever wait for
    // workgroups that are resident already.  The sort starts from bt.rle and re-initialises everything it uses, so it is
    // simply run again with that mapping, which this context then keeps.
    // (Whatever else the kernels behind the give-up reported -- they ran on garbage lists -- does not matter: the sort
    // restarts from bt.rle and clears the error word.)
    if ((err & 2u) && !had_fault && !ctx->no_spread) { // (late rounds of a large batch also launch over few blocks)
        ctx->no_spread = true;
        ctx->stats = stats_in; // the abandoned attempt is not counted
        ctx->sort_spans.resize(spans_in);
        ctx->kspans.resize(kspans_in);
        memcpy(ctx->k_bytes, kb_in, sizeof kb_in);
        memcpy(ctx->k_launch, kl_in, sizeof kl_in);
        static const bool say = getenv("BZH_TRACE_ROUNDS") != nullptr;
        if (say) fprintf(stderr, "[bzhip] a look-back gave up: the suffix sort runs again with every block on one XCD\n");
        return bwt_run(ctx, B, nmax, ntotal, true);
    }
    // (the last summary read is the one of a round that found nothing to do: every kernel before it has run, so
    // its error word and its sum of unresolved suffixes are final)
    ctx->stats.bwt_active_sum += (uint64_t)s[17] | ((uint64_t)s[18] << 32);
    if (err) {
        bzh_set_error(ctx, err & 8   ? "BWT: the bucket-first initial sort broke one of its invariants (internal error)"
                           : err & 4 ? "BWT: a launch was sized for fewer blocks or tiles than the round had (internal error)"
                           : err & 2 ? "BWT: a look-back gave up waiting (internal error)"
                                     : "BWT: a small-group window saw a group larger than its guarantee (internal error)");
        return BZH_E_HIP;
    }

    uint32_t gx = (nmax + 1023) / 1024;
    if (gx > 256) gx = 256;
    if (gx == 0) gx = 1;
    if (few_blocks(B)) gx |= WG_SPREAD;
    KSpan ks(ctx, K_BWT_EMIT, 6 * ntotal - emitted); // rank word and text byte in for every rotation, one byte out for those not yet written
    bwt_emit<<<dim3(xcd_grid(gx, B)), 256, 0, st>>>(bt, gx, B);
    period_expand<<<dim3(PX_WGS, B), 1024, 0, st>>>(bt, bt.pshrink);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}


// ---- inverse transform (verification tooling) -------------------------------------------
// The reference has no decoder (README.md:9); its fuzz target round-trips through libbz2
// (fuzz/fuzz_targets/round_trip.rs:8-22).  This is the device-side counterpart for the dominant stage: given a
// last column L and the origin pointer, rebuild the block.  One stable radix pass keyed on L's bytes (the same
// radix_scatter as the forward sort) yields T = LF^-1 (row r -> row of the rotation one byte further on);
// the walk X[i] = T^i(ptr) is then computed for all i at once by doubling: with X[0..m) and P = T^m known,
// X[m+i] = P[X[i]] and T^2m = P o P -- log2 n rounds of gathers, no serial list traversal, and blocks made of
// repeated words (several cycles in T) need no special care.  S[i] = L[X[i+1]].
struct UnbwtArgs {
    const uint8_t *L;    // [B][S] last column
    const uint32_t *n;   // [B]
    const uint32_t *ptr; // [B]
    const u64 *list;     // [B][S] sorted (byte, position) pairs
    uint32_t *P, *P2;    // [B][S] T^m and T^2m
    uint32_t *X;         // [B][S] X[i] = T^i(ptr)
    uint8_t *out;        // [B][S]
    uint32_t S, m;
};

__global__ void __launch_bounds__(256) unbwt_init(UnbwtArgs a)
{
    const uint32_t b = blockIdx.y, n = a.n[b];
    const size_t base = (size_t)b * a.S;
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) a.P[base + r] = (uint32_t)(a.list[base + r] & 0xFFFFFFFFull);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.X[base] = a.ptr[b];
}

__global__ void __launch_bounds__(256) unbwt_round(UnbwtArgs a)
{
    const uint32_t b = blockIdx.y, n = a.n[b], m = a.m;
    if (m >= n) return;
    const size_t base = (size_t)b * a.S;
    const uint32_t *P = a.P + base;
    const bool need_sq = 2u * m < n; // T^2m is only needed if another round follows
    const uint32_t ext = min(m, n - m);
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
        if (need_sq) a.P2[base + r] = P[P[r]];
        if (r < ext) a.X[base + m + r] = P[a.X[base + r]];
    }
}

__global__ void __launch_bounds__(256) unbwt_emit(UnbwtArgs a)
{
    const uint32_t b = blockIdx.y, n = a.n[b];
    const size_t base = (size_t)b * a.S;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t nx = i + 1 < n ? a.X[base + i + 1] : a.ptr[b];
        a.out[base + i] = a.L[base + nx];
    }
}

__global__ void __launch_bounds__(256) count_mismatch(const uint8_t *x, const uint8_t *y, const uint32_t *nn, uint32_t S, unsigned long long *acc)
{
    const uint32_t b = blockIdx.y, n = nn[b];
    const size_t base = (size_t)b * S;
    uint32_t bad = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) bad += x[base + i] != y[base + i];
    bad = wave_reduce_add(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(acc, (unsigned long long)bad);
}

// Inverse BWT of blocks 0..B-1 of the batch: reads bt.bwt / bt.ptr / bt.n, leaves the blocks in bt.mtfpos
// (scratch of the forward path).  Uses the sort buffers and rank / sa / headp as work arrays.
int unbwt_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax)
{
    Batch &bt = ctx->bt;
    if (B == 0 || nmax == 0) return BZH_OK;
    hipStream_t st = ctx->stream;
    const Lst all{nullptr, nullptr, B};
    u64 *bufA = reinterpret_cast<u64 *>(bt.listA);
    SortArgs a{};
    a.blk = bt.bwt;
    a.n = bt.n;
    a.cnt = bt.n;
    a.lst = all;
    a.S = bt.S;
    a.TPB = bt.TPB;
    a.shift = 32;
    a.src = nullptr;
    a.dst = bufA;
    a.look = reinterpret_cast<u64 *>(bt.hist);
    a.dbase = bt.dbase;
    a.doff = 0;
    a.err = bt.errflag;
    a.pass = 0;
    HIP_TRY(ctx, hipMemsetAsync(bt.errflag, 0, sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(a.look, 0, (size_t)B * bt.TPB * NBMAX * sizeof(u64), st));
    HIP_TRY(ctx, hipMemsetAsync(bt.dtot, 0, (size_t)B * DB_STRIDE * sizeof(uint32_t), st));
    byte_count<<<dim3(BYTE_SEGS, B), 1024, 0, st>>>(bt.bwt, bt.n, bt.dtot, bt.S);
    active_bases<<<dim3(B), 256, 0, st>>>(bt.dtot, bt.dbase, all, 1);
    launch_pass<8, GEN_LCOL>(ctx, a, B, nmax);
    UnbwtArgs u{};
    u.L = bt.bwt;
    u.n = bt.n;
    u.ptr = bt.ptr;
    u.list = bufA;
    u.P = bt.rank;
    u.P2 = bt.sa;
    u.X = bt.headp;
    u.out = bt.mtfpos;
    u.S = bt.S;
    const dim3 grid(std::min<uint32_t>((nmax + 1023) / 1024, 512), B);
    unbwt_init<<<grid, 256, 0, st>>>(u);
    for (uint32_t m = 1; m < nmax; m <<= 1) {
        u.m = m;
        unbwt_round<<<grid, 256, 0, st>>>(u);
        uint32_t *t = u.P;
        u.P = u.P2;
        u.P2 = t;
    }
    unbwt_emit<<<grid, 256, 0, st>>>(u);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

// mismatching bytes between bt.rle and the inverse transform's output over blocks 0..B-1 (added to *d_acc)
int unbwt_compare(bzh_ctx *ctx, uint32_t B, uint32_t nmax, unsigned long long *d_acc)
{
    Batch &bt = ctx->bt;
    if (B == 0 || nmax == 0) return BZH_OK;
    const dim3 grid(std::min<uint32_t>((nmax + 1023) / 1024, 512), B);
    count_mismatch<<<grid, 256, 0, ctx->stream>>>(bt.rle, bt.mtfpos, bt.n, bt.S, d_acc);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
