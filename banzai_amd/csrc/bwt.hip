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
// then doubling rounds from depth h = 8, each in one of three forms:
//   SWEEP  (most suffixes unresolved): unresolved suffixes are ENUMERATED in SA order of suffix
//          i+h (a coalesced sweep of SA + rank gathers) and stably sorted by r only -- 3 passes of
//          7 bits; stability leaves every group in key2 order, key2 is carried only for flagging;
//   ACTIVE (few unresolved): the previous round's sorted list is re-keyed once (active_gen) and
//          sorted on (r, key2) -- 5 passes of 8 bits over the unresolved suffixes only;
//   TAIL   (per block, once all its groups are small): groups are ranked locally (tail_*).
// Then boundary flags, max-scan of group heads, rank/SA update; resolved suffixes drop out.
// When h >= n the survivors are identical rotations (block = w^k): key2 becomes n-1-i (ACTIVE) or
// the enumeration runs over descending i (SWEEP), the reference's tie rule.
//
// Every radix pass is ONE kernel (radix_scatter): a tile publishes its digit counts and finds its
// first slots by decoupled look-back over the earlier tiles of its block; the digit totals a pass
// needs up front are a by-product of the step before it (byte_count / refine + sweep_bases /
// active_gen + active_bases), so nothing is ever read just to be counted.
//
// Launch geometry: workgroup ids are mapped so that all tiles of bzip2 block b run on XCD b mod 8
// (wg_map), keeping the block's rank/SA arrays (3.6 MB each) inside one 4 MiB L2.
// Kernels (integer only, HBM/LDS bound, no MFMA):
//   byte_count     digit totals of the 8 initial passes (= byte counts of the cyclic block)
//   radix_scatter  stable single-pass scatter: wave match-any ranking, per-wave LDS cursors,
//                  look-back for the tile's global offsets, elements reordered in LDS so each
//                  digit's run leaves the CU as coalesced stores
//   flag_tiles / flag_carry / refine   boundary flags, max-scan of group heads, rank + SA update
//                  (tiles staged through LDS: coalesced global access, blocked per-thread scans)
//   sweep_bases / active_gen / active_bases   digit bases of the SWEEP / ACTIVE passes
//   tail_sort / tail_finish   TAIL rounds
//   bwt_emit       last column, ptr, has_byte
#include <vector>

#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned long long u64;

enum GenMode : int {
    GEN_BYTES4 = 0, // element e is suffix e keyed by its 4-byte cyclic prefix
    GEN_SWEEP = 1,  // doubling round, SA-order enumeration
    GEN_ACTIVE = 2, // doubling round, re-key the previous sorted list
    GEN_LIST = 3,   // element e is src[e]
    GEN_LISTH = 4   // element e is src[e] unless that is LIST_INVALID (a hole)
};

constexpr u64 SUF_MASK = 0xFFFFFull;
constexpr uint32_t RANK_MASK = 0x7FFFFFFFu;

struct SortArgs {
    const uint8_t *blk;   // [B][S]
    const uint32_t *n;    // [B]
    const uint32_t *cnt;  // [B] elements enumerated this pass
    const uint32_t *gate; // [B] skip block when 0
    const uint32_t *rank; // [B][S]
    const uint32_t *sa;   // [B][S]
    const uint32_t *headp; // [B][S] group rank of the suffix at each SA position
    const u64 *src;       // [B][S]
    u64 *dst;             // [B][S]
    uint32_t S, TPB, h, shift;
    uint32_t T, B; // launch geometry: tiles per block in this launch, blocks
    uint32_t recrank; // GEN_ACTIVE: src records carry the suffix's current group rank (refine wrote it back)
    // single-pass (look-back) scatter of the initial sort only:
    u64 *look;             // [B][TPB][256] tile status words  [pass:32][state:2][count:30]
    const uint32_t *dbase; // [B][DB_STRIDE] first slot of every digit (exclusive scan of the pass's digit totals)
    uint32_t doff;         // which 128/256-entry group of dbase this pass uses
    uint32_t *err;         // bit 1: a look-back gave up (internal error, never a hang)
    uint32_t pass;         // id of this pass in the status words (stale words read as "not there yet")
};

constexpr int NBMAX = 256;
constexpr u64 LIST_INVALID = ~0ull; // list slot without an unresolved suffix

// XCD-aware workgroup -> (bzip2 block, tile) map.  Workgroups are dealt round-robin over the 8
// XCDs (observed dispatch behaviour, used for speed only): ids congruent mod 8 share an XCD and
// its private 4 MiB L2.  All tiles of block b get ids = b (mod 8).  Grid = 8*ceil(B/8)*T.
// With fewer than 8 active blocks that would leave XCDs idle (a single block would run on 32 of the
// 256 CUs): launches that know they have few active blocks set WG_SPREAD in T and get the plain
// mapping, consecutive workgroup ids = consecutive tiles of one block, i.e. every block on all XCDs.
// Either way tile t-1 of a block has a lower workgroup id than tile t (the look-backs rely on it).
constexpr uint32_t WG_SPREAD = 0x80000000u;
__device__ __forceinline__ bool wg_map(uint32_t T, uint32_t B, uint32_t &b, uint32_t &tile)
{
    const uint32_t L = blockIdx.x;
    if (T & WG_SPREAD) {
        T &= ~WG_SPREAD;
        b = L / T;
        tile = L - b * T;
        return b < B;
    }
    const uint32_t slot = L >> 3;
    const uint32_t k = slot / T;
    tile = slot - k * T;
    b = k * 8u + (L & 7u);
    return b < B;
}

static inline uint32_t xcd_grid(uint32_t tiles, uint32_t B)
{
    return (tiles & WG_SPREAD) ? (tiles & ~WG_SPREAD) * B : 8u * ((B + 7u) / 8u) * tiles;
}

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

// WANT_K2 = false for histogram passes that only look at the r digits.
template <int MODE, bool WANT_K2>
__device__ __forceinline__ bool gen_elem(const SortArgs &a, uint32_t b, uint32_t e, uint32_t n, u64 &v)
{
    const size_t base = (size_t)b * a.S;
    if (MODE == GEN_BYTES4) {
        // a.h = byte offset of the key inside the rotation (4 for the low half of the 8-byte prefix)
        uint32_t i = e + a.h;
        if (i >= n) i = n > 4 ? i - n : i % n;
        const uint32_t key = text4(a.blk + base, i, n);
        v = ((u64)key << 32) | e;
        return true;
    } else if (MODE == GEN_SWEEP) {
        uint32_t i, k2;
        if (a.h < n) { // suffix j = sa[e] is the e-th smallest; i = j - h has it as its second half
            const uint32_t j = a.sa[base + e];
            i = j >= a.h ? j - a.h : j + n - a.h;
            const uint32_t r = a.rank[base + i];
            if (r & RANK_RESOLVED) return false;
            k2 = WANT_K2 ? a.headp[base + e] : 0u; // = rank[j] without the gather
            v = ((u64)r << 40) | ((u64)k2 << 20) | i;
            return true;
        }
        i = n - 1 - e; // identical rotations: larger index first; e doubles as a distinct key2
        const uint32_t r = a.rank[base + i];
        if (r & RANK_RESOLVED) return false;
        v = ((u64)r << 40) | ((u64)e << 20) | i;
        return true;
    } else if (MODE == GEN_ACTIVE) {
        const u64 x = a.src[base + e];
        const uint32_t i = (uint32_t)(x & SUF_MASK);
        uint32_t r;
        if (a.recrank) { // the last refine left [new rank][.][i], or LIST_INVALID for resolved suffixes
            if (x == LIST_INVALID) return false;
            r = (uint32_t)(x >> 40) & 0xFFFFFu;
        } else {
            r = a.rank[base + i];
            if (r & RANK_RESOLVED) return false;
        }
        uint32_t k2;
        if (a.h < n) {
            uint32_t i2 = i + a.h;
            if (i2 >= n) i2 -= n;
            k2 = a.rank[base + i2] & RANK_MASK;
        } else {
            k2 = n - 1 - i;
        }
        v = ((u64)r << 40) | ((u64)k2 << 20) | i;
        return true;
    } else if (MODE == GEN_LISTH) {
        v = a.src[base + e];
        return v != LIST_INVALID;
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
    const uint32_t per = ((n + BYTE_SEGS - 1) / BYTE_SEGS + 3u) & ~3u; // keeps the dword loads aligned
    const uint32_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    if (lo >= hi) return;
    __shared__ uint32_t h[16][256];
    for (int k = threadIdx.x; k < 16 * 256; k += 1024) (&h[0][0])[k] = 0;
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
    for (uint32_t i = lo + threadIdx.x * 4; i < hi; i += 4096) {
        if (i + 4 <= hi) {
            uint32_t w;
            __builtin_memcpy(&w, s + i, 4);
            atomicAdd(&mine[w & 255u], 1u);
            atomicAdd(&mine[(w >> 8) & 255u], 1u);
            atomicAdd(&mine[(w >> 16) & 255u], 1u);
            atomicAdd(&mine[w >> 24], 1u);
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

// ACTIVE round, step 1: re-key the previous sorted list ONCE (gen_elem<GEN_ACTIVE>: one gather per
// unresolved suffix) into dst -- same slot, LIST_INVALID where the suffix is resolved -- and count
// all five 8-bit digits of the new keys (bits 20..59) into the block's totals, from which
// active_bases makes the bases of the five look-back passes that follow.
__global__ void __launch_bounds__(SORT_THREADS) active_gen(SortArgs a, uint32_t *dtot)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.B, b, tile)) return;
    if (a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b], n = a.n[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    __shared__ uint32_t h[5 * 256];
    for (int k = threadIdx.x; k < 5 * 256; k += SORT_THREADS) h[k] = 0;
    __syncthreads();
    u64 *dst = a.dst + (size_t)b * a.S;
    const int lane = threadIdx.x & 63;
#pragma unroll 4
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t e = tile * SORT_TILE + k * SORT_THREADS + threadIdx.x;
        u64 v = LIST_INVALID;
        const bool ok = e < cnt && gen_elem<GEN_ACTIVE, true>(a, b, e, n, v);
        if (e < cnt) dst[e] = ok ? v : LIST_INVALID;
        if (ok) {
            atomicAdd(&h[(uint32_t)(v >> 20) & 255u], 1u);
            atomicAdd(&h[256 + ((uint32_t)(v >> 28) & 255u)], 1u);
            atomicAdd(&h[512 + ((uint32_t)(v >> 36) & 255u)], 1u);
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
                atomicAdd(&h[768 + (u0 & 255u)], c);
                atomicAdd(&h[1024 + (u0 >> 8)], c);
            }
            todo &= ~same;
        }
    }
    __syncthreads();
    uint32_t *tot = dtot + (size_t)b * DB_STRIDE;
    for (int k = threadIdx.x; k < 5 * 256; k += SORT_THREADS)
        if (h[k]) atomicAdd(&tot[k], h[k]);
}

// One workgroup per block: exclusive scan inside each of the ndig digit groups of dtot.
__global__ void __launch_bounds__(256) active_bases(const uint32_t *dtot, uint32_t *dbase, const uint32_t *gate, int ndig)
{
    const uint32_t b = blockIdx.x;
    if (gate[b] == 0) return;
    __shared__ uint32_t ls[8];
#pragma unroll 1
    for (int p = 0; p < ndig; p++) {
        const uint32_t v = dtot[(size_t)b * DB_STRIDE + p * 256 + threadIdx.x];
        uint32_t tot;
        const uint32_t ex = block_excl_add(v, ls, &tot);
        dbase[(size_t)b * DB_STRIDE + p * 256 + threadIdx.x] = ex;
    }
}

constexpr uint32_t LOOK_LOCAL = 1u, LOOK_GLOBAL = 2u;
__device__ __forceinline__ u64 look_word(uint32_t pass, uint32_t state, uint32_t count)
{
    return ((u64)pass << 32) | ((u64)state << 30) | count;
}

// REKEY: the element leaves with the key of the NEXT key half (bytes i..i+3 of the rotation) --
// used by the last pass over the low half of the 8-byte prefix.
// Single pass: no histogram / scan launches before this kernel.  The tile publishes its digit
// counts, then each digit's thread looks back over the earlier tiles of the block (decoupled
// look-back: a predecessor offers either its own counts or, once known, its inclusive prefix) for
// the tile's first slot; digit bases come from a.dbase.  A status word is one 64-bit atomic, so no
// fences are needed; tiles only ever wait for LOWER workgroup ids, which the dispatcher starts
// first; waits are bounded (a.err) so that a logic error cannot hang the device.
template <int BITS, int MODE, bool REKEY = false>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter(SortArgs a)
{
    constexpr int NB = 1 << BITS;
    constexpr int NW = SORT_THREADS / 64;
    uint32_t b, tile;
    if (!wg_map(a.T, a.B, b, tile)) return;
    if (a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b], n = a.n[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    __shared__ uint32_t cur[NW][NB];  // per-wave counters, then cursors (tile-local positions)
    __shared__ uint32_t binstart[NB]; // tile-local start of each digit's run
    __shared__ uint32_t goff[NB];     // global offset of this tile's run of each digit
    __shared__ u64 stage[SORT_TILE];  // tile in digit order
    __shared__ uint32_t ls[NW + 2];
    for (int k = threadIdx.x; k < NW * NB; k += SORT_THREADS) (&cur[0][0])[k] = 0;

    // wave w owns the contiguous run [w*ITEMS*64, (w+1)*ITEMS*64) of the tile, 64 elements a step
    u64 v[SORT_ITEMS];
    uint32_t actmask = 0;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t e = tile * SORT_TILE + wave * (SORT_ITEMS * 64) + k * 64 + lane;
        v[k] = 0;
        if (e < cnt && gen_elem<MODE, true>(a, b, e, n, v[k])) actmask |= 1u << k;
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
    if (threadIdx.x < NB) {
        const uint32_t bin = threadIdx.x;
        binstart[bin] = ex;
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
    if (threadIdx.x < NB) { // look back for the counts of digit `bin` in tiles 0 .. tile-1
        const uint32_t bin = threadIdx.x;
        u64 *col = a.look + (size_t)b * a.TPB * NBMAX + bin;
        uint32_t acc = 0, spins = 0;
        int t = (int)tile - 1;
        while (t >= 0) {
            const u64 w = __hip_atomic_load(col + (size_t)t * NBMAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t state = (uint32_t)(w >> 30) & 3u;
            if ((uint32_t)(w >> 32) != a.pass || state == 0) { // predecessor has not published yet
                if (++spins > (1u << 26)) { // seconds: only a logic error gets here
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
        __hip_atomic_store(col + (size_t)tile * NBMAX, look_word(a.pass, LOOK_GLOBAL, acc + mytot), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        goff[bin] = a.dbase[(size_t)b * DB_STRIDE + a.doff + bin] + acc;
    }
    __syncthreads();
    u64 *dst = a.dst + (size_t)b * a.S;
    for (uint32_t e = threadIdx.x; e < tile_total; e += SORT_THREADS) {
        u64 x = stage[e];
        const uint32_t d = (uint32_t)(x >> a.shift) & (NB - 1);
        if (REKEY) {
            const uint32_t i = (uint32_t)(x & SUF_MASK);
            x = ((u64)text4(a.blk + (size_t)b * a.S, i, n) << 32) | i;
        }
        dst[goff[d] + (e - binstart[d])] = x;
    }
}

// ---- group refinement ---------------------------------------------------------------------------
struct RefineArgs {
    const uint32_t *n;   // [B]
    const uint32_t *cnt; // [B] list length (n for the init pass, unresolved count in rounds)
    const u64 *list;     // [B][S] sorted elements
    u64 *wb;             // [B][S] or nullptr: the block's OTHER list buffer; receives the still unresolved suffixes,
                         // compacted and in order, as [new rank:20 @40][0][suffix:20]
    u64 *cstat;          // tile status words of the compaction's look-back (word 192 of the tile's hist row)
    uint32_t cpass;      // pass id in those words
    uint32_t *err;       // bit 1: a look-back gave up
    const uint8_t *blk;  // [B][S] the text (init pass: low half of the 8-byte prefix is compared from it)
    uint32_t *rank;      // [B][S]
    uint32_t *sa;        // [B][S]
    uint32_t *headp;     // [B][S]
    uint8_t *flg;        // [B][S]
    int2 *tagg;          // [B][TPB]
    uint32_t *dig;       // [B][TPB][512] or nullptr: per tile, counts of the three 7-bit digits of the new rank over
                         // the suffixes left unresolved (bases of the next SWEEP round's look-back passes)
    uint32_t *nact_next; // [B]
    uint32_t *maxgrp;    // [B] largest refined group (members), atomicMax
    const uint32_t *gate; // [B] skip block when 0 (nullptr = no gating)
    uint32_t S, TPB;
    int init;
    uint32_t T, B;
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

// lo/plo: bytes 4..7 of the rotations of cur / prev (init pass only)
__device__ __forceinline__ void elem_flags(const RefineArgs &a, uint32_t q, u64 cur, u64 prev, uint32_t lo, uint32_t plo,
                                           bool &gs, bool &bd)
{
    if (a.init) {
        gs = (q == 0);
        bd = gs || (cur >> 32) != (prev >> 32) || lo != plo;
    } else {
        gs = (q == 0) || (cur >> 40) != (prev >> 40);
        bd = gs || (cur >> 20) != (prev >> 20);
    }
}

// flag bit0: first element of its (old) group; bit1: first element of its refined group
__global__ void __launch_bounds__(SORT_THREADS) flag_tiles(RefineArgs a)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.B, b, tile)) return;
    if (a.gate && a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    const size_t base = (size_t)b * a.S;
    const u64 *list = a.list + base;
    const uint32_t tile0 = tile * SORT_TILE;
    __shared__ u64 lds[STAGE_SLOTS];
    stage_tile(list, tile0, cnt, lds);
    __syncthreads();
    const uint32_t e0 = threadIdx.x * SORT_ITEMS, q0 = tile0 + e0;
    uint32_t packed[SORT_ITEMS / 4] = {0, 0, 0, 0};
    int lastgs = -1, lastbd = -1;
    if (q0 < cnt) {
        u64 prev = e0 ? lds[slot_of(e0 - 1)] : (q0 ? list[q0 - 1] : 0ull);
        const uint32_t n = a.n[b];
        const uint8_t *txt = a.blk + base;
        uint32_t plo = 0;
        if (a.init && q0) plo = text4(txt, wrap_add(prev, n), n);
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (q < cnt) {
                const u64 cur = lds[slot_of(e0 + k)];
                uint32_t lo = 0;
                if (a.init) lo = text4(txt, wrap_add(cur, n), n);
                bool gs, bd;
                elem_flags(a, q, cur, prev, lo, plo, gs, bd);
                plo = lo;
                if (gs) lastgs = (int)q;
                if (bd) lastbd = (int)q;
                packed[k >> 2] |= ((gs ? 1u : 0u) | (bd ? 2u : 0u)) << ((k & 3) * 8);
                prev = cur;
            }
        }
        *reinterpret_cast<uint4 *>(a.flg + base + q0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
    __shared__ int red[2][SORT_THREADS / 64];
    int g = lastgs, d = lastbd;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        g = max(g, __shfl_xor(g, s, 64));
        d = max(d, __shfl_xor(d, s, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = g;
        red[1][threadIdx.x >> 6] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SORT_THREADS / 64; w++) {
            g = max(g, red[0][w]);
            d = max(d, red[1][w]);
        }
        a.tagg[(size_t)b * a.TPB + tile] = make_int2(g, d);
    }
}

// One workgroup per bzip2 block: exclusive max-scan of the tile aggregates (carry into each tile).
// ntile <= TPB <= 1024 (checked at context creation), so one sweep of 1024 threads covers it.
__global__ void __launch_bounds__(1024) flag_carry(RefineArgs a)
{
    const uint32_t b = blockIdx.x;
    if (a.gate && a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (ntile == 0) return;
    int2 *t = a.tagg + (size_t)b * a.TPB;
    __shared__ int l01[32];
    __shared__ int inc0[1024], inc1[1024];
    const uint32_t e = threadIdx.x;
    const int2 v = e < ntile ? t[e] : make_int2(-1, -1);
    int s0 = v.x, s1 = v.y;
    block_incl_max2(s0, s1, l01);
    inc0[e] = s0;
    inc1[e] = s1;
    __syncthreads();
    if (e < ntile) t[e] = e == 0 ? make_int2(-1, -1) : make_int2(inc0[e - 1], inc1[e - 1]);
}

// WB: also write the compacted list of the suffixes that stay unresolved (a.wb)
template <bool WB>
__global__ void __launch_bounds__(SORT_THREADS) refine(RefineArgs a)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.B, b, tile)) return;
    if (a.gate && a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    const size_t base = (size_t)b * a.S;
    const uint32_t tile0 = tile * SORT_TILE;
    __shared__ u64 lds[STAGE_SLOTS];
    const uint32_t e0 = threadIdx.x * SORT_ITEMS, q0 = tile0 + e0;

    // flags first: the tile's unresolved count depends on nothing else, and publishing it before the
    // (slow: the memory system is saturated by rank scatters) staging of the tile lets the later tiles'
    // look-back find it in place
    uint32_t packed[4] = {0, 0, 0, 0};
    uint32_t nextflag = 2; // flag of element q0+16 (end of list counts as a boundary)
    if (q0 < cnt) {
        const uint4 f = *reinterpret_cast<const uint4 *>(a.flg + base + q0);
        packed[0] = f.x;
        packed[1] = f.y;
        packed[2] = f.z;
        packed[3] = f.w;
        if (q0 + SORT_ITEMS < cnt) nextflag = a.flg[base + q0 + SORT_ITEMS];
    }
    int tg = -1, td = -1;
    uint32_t ucnt = 0; // own elements that stay unresolved (a singleton = boundary followed by a boundary)
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
        const uint32_t fn = (k + 1 < SORT_ITEMS) ? ((packed[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 3u) : nextflag;
        if (q0 + k < cnt) {
            if (f & 1u) tg = (int)(q0 + k);
            if (f & 2u) td = (int)(q0 + k);
            if (WB) ucnt += ((f & 2u) && ((q0 + k + 1 == cnt) || (fn & 2u))) ? 0u : 1u;
        }
    }
    // compaction of the unresolved records: slot = (unresolved in earlier tiles: look-back over the
    // tile counts) + (in earlier threads of the tile) + (before the element in the thread)
    __shared__ uint32_t lsu[SORT_THREADS / 64 + 2];
    __shared__ uint32_t cpre;
    uint32_t utot = 0, uoff = 0;
    u64 *cst = a.cstat + (size_t)b * a.TPB * NBMAX + 192;
    if (WB) {
        uoff = block_excl_add(ucnt, lsu, &utot);
        if (threadIdx.x == 0)
            __hip_atomic_store(cst + (size_t)tile * NBMAX, look_word(a.cpass, tile ? LOOK_LOCAL : LOOK_GLOBAL, utot),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stage_tile(a.list + base, tile0, cnt, lds);
    __shared__ int l01[2 * SORT_THREADS / 64];
    __shared__ int ex0[SORT_THREADS], ex1[SORT_THREADS];
    block_incl_max2(tg, td, l01);
    ex0[threadIdx.x] = tg;
    ex1[threadIdx.x] = td;
    __syncthreads(); // also orders stage_tile's stores before the blocked reads below
    const int2 tc = a.tagg[(size_t)b * a.TPB + tile];
    int cg = tc.x, cd = tc.y;
    if (threadIdx.x > 0) {
        cg = max(cg, ex0[threadIdx.x - 1]);
        cd = max(cd, ex1[threadIdx.x - 1]);
    }
    uint32_t wslot = 0;
    if (WB) {
        if (threadIdx.x == 0) {
            uint32_t acc = 0, spins = 0;
            if (tile > 0) {
                int t = (int)tile - 1;
                while (t >= 0) {
                    const u64 w = __hip_atomic_load(cst + (size_t)t * NBMAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t state = (uint32_t)(w >> 30) & 3u;
                    if ((uint32_t)(w >> 32) != a.cpass || state == 0) {
                        if (++spins > (1u << 26)) {
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
                __hip_atomic_store(cst + (size_t)tile * NBMAX, look_word(a.cpass, LOOK_GLOBAL, acc + utot), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            cpre = acc;
        }
        __syncthreads();
        wslot = cpre + uoff;
    }
    uint32_t *rank = a.rank + base;
    uint32_t *sa = a.sa + base;
    uint32_t *headp = a.headp + base;
    u64 *wb = WB ? a.wb + base : nullptr;
    __shared__ uint32_t dh[384];
    if (a.dig) {
        for (int k = threadIdx.x; k < 384; k += SORT_THREADS) dh[k] = 0;
        __syncthreads();
    }
    uint32_t unresolved = 0, biggest = 0;
    uint32_t ph = 0, pc = 0, ph7 = 0, pc7 = 0; // open runs of the digit counting
    u64 outv[SORT_ITEMS]; // SA position : group rank : suffix (20 bits each), all ones = none
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) outv[k] = ~0ull;
    if (q0 < cnt) {
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (q < cnt) {
                const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
                const uint32_t fn = (k + 1 < SORT_ITEMS) ? ((packed[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 3u) : nextflag;
                if (f & 1u) cg = (int)q;
                if (f & 2u) cd = (int)q;
                const u64 cur = lds[slot_of(e0 + k)];
                const uint32_t i = (uint32_t)(cur & SUF_MASK);
                // SA position of the group's first list entry, minus that entry's list index
                const uint32_t gbase = a.init ? 0u : ((uint32_t)(cur >> 40) - (uint32_t)cg);
                const uint32_t pos = gbase + q;
                const uint32_t head = gbase + (uint32_t)cd;
                const bool single = (f & 2u) && ((q + 1 == cnt) || (fn & 2u));
                rank[i] = single ? (head | RANK_RESOLVED) : head;
                outv[k] = ((u64)(single ? 1u : 0u) << 60) | ((u64)pos << 40) | ((u64)head << 20) | i;
                unresolved += single ? 0u : 1u;
                if (WB && !single) wb[wslot++] = ((u64)head << 40) | i; // ranked record for ACTIVE re-keying / TAIL
                if (a.dig && !single) {
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
                if ((q + 1 == cnt) || (fn & 2u)) biggest = max(biggest, q - (uint32_t)cd + 1u); // last of its group
            }
        }
    }
    if (pc) atomicAdd(&dh[ph & 127u], pc);
    if (pc7) {
        atomicAdd(&dh[128 + (ph7 & 127u)], pc7);
        atomicAdd(&dh[256 + (ph7 >> 7)], pc7);
    }
    // SA update through LDS so that consecutive lanes store consecutive positions
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) lds[slot_of(e0 + k)] = outv[k];
    __syncthreads();
    if (a.dig) {
        uint32_t *row = a.dig + ((size_t)b * a.TPB + tile) * 512;
        for (int k = threadIdx.x; k < 384; k += SORT_THREADS) row[k] = dh[k];
    }
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t e = k * SORT_THREADS + threadIdx.x;
        const u64 x = lds[slot_of(e)];
        if (x != ~0ull) {
            const uint32_t pos = (uint32_t)(x >> 40) & 0xFFFFFu;
            const uint32_t head = (uint32_t)(x >> 20) & 0xFFFFFu;
            sa[pos] = (uint32_t)(x & SUF_MASK);
            headp[pos] = head;
        }
    }
    unresolved = wave_reduce_add(unresolved);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, d, 64));
    // one atomic pair per workgroup: all tiles of a block hit the same two counters
    __shared__ uint32_t wsum[SORT_THREADS / 64], wmax[SORT_THREADS / 64];
    if ((threadIdx.x & 63) == 0) {
        wsum[threadIdx.x >> 6] = unresolved;
        wmax[threadIdx.x >> 6] = biggest;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t su = 0, mx = 0;
#pragma unroll
        for (int w = 0; w < SORT_THREADS / 64; w++) {
            su += wsum[w];
            mx = max(mx, wmax[w]);
        }
        if (su) {
            atomicAdd(&a.nact_next[b], su);
            atomicMax(&a.maxgrp[b], mx);
        }
    }
}

// ---- TAIL rounds: every group small -> sort groups locally, 4 small launches per round -------------
// Precondition (tracked by refine's maxgrp): every unresolved group of the block has at most TAIL_G
// members.  The block's unresolved suffixes sit, grouped and in SA order, in `len` slots of one list
// buffer.  tail_sort: a workgroup owns the groups whose first member lies in its range of TAIL_T
// slots and sees TAIL_G slots either side, so every owned group is complete in its window; it ranks
// the members of each group by key2 and writes a record for the slot of the u-th member into the
// block's OTHER list buffer (same slot numbering), so the input stays untouched during the kernel
// and its records can be trusted: each carries the suffix and its current group rank (written by
// refine, or by the previous tail round), so nothing has to be gathered but key2.  tail_sort reads
// only OLD ranks; tail_finish stores the new ranks / SA entries (the kernel boundary keeps rank reads
// consistent) and moves the still-unresolved records, order preserved, back to the block's own
// buffer, so the next round touches only what is left.
constexpr int TAIL_T = 2048, TAIL_G = 512, TAIL_W = TAIL_T + 2 * TAIL_G, TAIL_THREADS = 512;
constexpr int TAIL_PER = TAIL_W / TAIL_THREADS; // 6 slots per thread
constexpr int TAIL_SMALL = 32; // groups up to this size are ranked by the threads holding their members
constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t TAIL_BUF_B = 0x80000000u; // gateT bit: the block's list lives in listB
constexpr uint32_t TAIL_LEN = 0x3FFFFFFFu;

struct TailArgs {
    const uint32_t *n;   // [B]
    const uint32_t *len; // [B] slot count | TAIL_BUF_B, 0 = block not in tail mode this round
    uint32_t recrank;    // records carry the current group rank (always, except right after the initial sort)
    u64 *bufA, *bufB;    // [B][S]
    uint32_t *rank;      // [B][S]
    uint32_t *sa;        // [B][S]
    uint32_t *nact_next; // [B]
    u64 *stat;           // [B][TT] tile status words of tail_finish's look-back
    uint32_t pass;       // pass id in those words
    uint32_t *err;       // [1] precondition violations
    const uint32_t *hb;  // [B] depth h of each block (TAIL blocks advance on their own: x4 while the radix path doubles)
    uint32_t S, T, B, TT;
};

// record: [resolved:1 @60][new rank:20 @40][SA position:20 @20][suffix:20 @0]
template <bool QUAD>
__global__ void __launch_bounds__(TAIL_THREADS) tail_sort(TailArgs a)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.B, b, tile)) return;
    const uint32_t lenw = a.len[b];
    const uint32_t len = lenw & TAIL_LEN;
    const uint32_t r0 = tile * TAIL_T;
    if (r0 >= len) return;
    const uint32_t r1 = min(len, r0 + (uint32_t)TAIL_T);
    const uint32_t s_lo = r0 >= (uint32_t)TAIL_G ? r0 - TAIL_G : 0u;
    const uint32_t s_hi = min(len, r1 + (uint32_t)TAIL_G);
    const uint32_t nwin = s_hi - s_lo;
    const uint32_t n = a.n[b], h = a.hb[b];
    const size_t base = (size_t)b * a.S;
    const u64 *buf = ((lenw & TAIL_BUF_B) ? a.bufB : a.bufA) + base; // read only during this kernel
    u64 *out = ((lenw & TAIL_BUF_B) ? a.bufA : a.bufB) + base;       // records, same slot numbering
    const uint32_t *rank = a.rank + base;
    // A0 group rank, A1 suffix | window offset of the slot << 20, A3 (A4, A5) keys, GE group end (at the head)
    __shared__ uint32_t A0[TAIL_W], A1[TAIL_W], A3[TAIL_W];
    __shared__ uint32_t A4[QUAD ? TAIL_W : 1], A5[QUAD ? TAIL_W : 1]; // extra keys of the 4h form only
    __shared__ uint16_t GE[TAIL_W];
    __shared__ uint32_t ls[TAIL_THREADS / 64 + 2];
    __shared__ int lm[TAIL_THREADS / 64];
    __shared__ int exh[TAIL_THREADS];

    // load by slot (coalesced); slots without an unresolved suffix drop out
#pragma unroll
    for (int k = 0; k < TAIL_PER; k++) {
        const uint32_t w = k * TAIL_THREADS + threadIdx.x;
        uint32_t r = NONE32, i = 0;
        if (w < nwin) {
            const uint32_t sl = s_lo + w;
            const u64 x = buf[sl];
            if (x != LIST_INVALID) {
                i = (uint32_t)(x & SUF_MASK);
                if (a.recrank) { // refine / the last tail round left the group rank in the record
                    r = (uint32_t)(x >> 40) & 0xFFFFFu;
                } else { // list straight from the initial sort: rank and resolvedness from the array
                    const uint32_t rr = rank[i];
                    if (!(rr & RANK_RESOLVED)) r = rr;
                }
            }
            // no unresolved suffix here: a hole for tail_finish
            if (r == NONE32 && sl >= r0 && sl < r1) out[sl] = LIST_INVALID;
        }
        A0[w] = r;
        A1[w] = i;
    }
    __syncthreads();
    // order-preserving compaction: a thread owns TAIL_PER consecutive slots and, afterwards, the
    // (consecutive) compacted elements they hold
    const uint32_t w0 = threadIdx.x * TAIL_PER;
    uint32_t cr[TAIL_PER], ci[TAIL_PER], cnt = 0;
#pragma unroll
    for (int k = 0; k < TAIL_PER; k++) {
        const uint32_t r = A0[w0 + k], i = A1[w0 + k];
        // keep registers dense: element j of this thread is its j-th unresolved slot
#pragma unroll
        for (int j = 0; j < TAIL_PER; j++) {
            if (r != NONE32 && (uint32_t)j == cnt) {
                cr[j] = r;
                ci[j] = i | ((w0 + k) << 20);
            }
        }
        cnt += r != NONE32;
    }
    uint32_t V;
    const uint32_t idx0 = block_excl_add(cnt, ls, &V);
#pragma unroll
    for (int j = 0; j < TAIL_PER; j++) {
        if ((uint32_t)j < cnt) {
            A0[idx0 + j] = cr[j];
            A1[idx0 + j] = ci[j];
        }
    }
    __syncthreads();
    // group structure by scan: start of every element's group, end of the group stored at its head
    const uint32_t prevr = (cnt && idx0 > 0) ? A0[idx0 - 1] : NONE32;
    const uint32_t nextr = (cnt && idx0 + cnt < V) ? A0[idx0 + cnt] : NONE32;
    int lasthead = -1;
#pragma unroll
    for (int j = 0; j < TAIL_PER; j++)
        if ((uint32_t)j < cnt && cr[j] != (j ? cr[j - 1] : prevr)) lasthead = (int)(idx0 + j);
    const int inc = block_incl_max(lasthead, lm);
    exh[threadIdx.x] = inc;
    __syncthreads();
    int run = threadIdx.x ? exh[threadIdx.x - 1] : -1;
    uint32_t gstart[TAIL_PER];
#pragma unroll
    for (int j = 0; j < TAIL_PER; j++) {
        gstart[j] = 0;
        if ((uint32_t)j < cnt) {
            if (cr[j] != (j ? cr[j - 1] : prevr)) run = (int)(idx0 + j);
            gstart[j] = (uint32_t)run;
            const uint32_t nx = ((uint32_t)(j + 1) < cnt) ? cr[(j + 1 < TAIL_PER) ? j + 1 : j] : nextr;
            if (nx != cr[j]) GE[run] = (uint16_t)(idx0 + j + 1);
        }
    }
    // ownership and the keys of owned elements
    uint32_t owned = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < TAIL_PER; j++) {
        if ((uint32_t)j < cnt) {
            const uint32_t e = idx0 + j;
            const uint32_t fs = s_lo + (A1[gstart[j]] >> 20);
            if (fs >= r0 && fs < r1) {
                owned |= 1u << j;
                if (s_lo + (ci[j] >> 20) - fs >= (uint32_t)TAIL_G) bad = true; // span beyond the window guarantee
                const uint32_t i = ci[j] & (uint32_t)SUF_MASK;
                uint32_t k2, k3 = 0, k4 = 0;
                if (h < n) {
                    uint32_t i2 = i + h;
                    if (i2 >= n) i2 -= n;
                    k2 = rank[i2] & RANK_MASK;
                    if (QUAD) { // two more h-blocks of the (cyclic) rotation
                        uint32_t i3 = i2 + h;
                        if (i3 >= n) i3 -= n;
                        uint32_t i4 = i3 + h;
                        if (i4 >= n) i4 -= n;
                        k3 = rank[i3] & RANK_MASK;
                        k4 = rank[i4] & RANK_MASK;
                    }
                } else {
                    k2 = n - 1 - i; // identical rotations: larger index first
                }
                A3[e] = k2;
                if (QUAD) {
                    A4[e] = k3;
                    A5[e] = k4;
                }
            }
        }
    }
    // large owned groups are listed for the cooperative pass below (each by the thread holding its head)
    __shared__ uint32_t nbig;
    __shared__ uint32_t bigs[TAIL_W / (TAIL_SMALL + 1) + 1]; // start | size << 16
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TAIL_PER; j++) {
        if ((uint32_t)j < cnt && (owned & (1u << j))) {
            const uint32_t e = idx0 + j, g = gstart[j];
            const uint32_t sz = (uint32_t)GE[g] - g;
            if (e == g && sz > (uint32_t)TAIL_SMALL) bigs[atomicAdd(&nbig, 1u)] = g | (sz << 16);
        }
    }
    // small groups: every member counts its own predecessors (at most TAIL_SMALL steps)
#pragma unroll
    for (int j = 0; j < TAIL_PER; j++) {
        if ((uint32_t)j < cnt && (owned & (1u << j))) {
            const uint32_t e = idx0 + j, r = cr[j], g = gstart[j];
            const uint32_t ge = GE[g];
            if (ge - g > (uint32_t)TAIL_SMALL) continue;
            // lexicographic key (k2, k3, k4): two 64-bit words compare it
            const uint32_t my_hi = A3[e];
            const u64 my_lo = QUAD ? (((u64)A4[e] << 32) | A5[e]) : 0ull;
            uint32_t less = 0, eq_before = 0, eq = 0;
#pragma unroll 4
            for (uint32_t f = g; f < ge; f++) { // bounds known up front: the LDS reads pipeline
                const uint32_t f_hi = A3[f];
                const u64 f_lo = QUAD ? (((u64)A4[f] << 32) | A5[f]) : 0ull;
                const bool same = f_hi == my_hi && f_lo == my_lo;
                less += (f_hi < my_hi) || (f_hi == my_hi && f_lo < my_lo);
                eq += same;
                eq_before += same && (f < e);
            }
            const uint32_t u = less + eq_before;
            const bool single = eq == 1;
            const u64 rec = ((u64)(single ? 1u : 0u) << 60) | ((u64)(r + less) << 40) | ((u64)(r + u) << 20) |
                            (ci[j] & (uint32_t)SUF_MASK);
            out[s_lo + (A1[g + u] >> 20)] = rec; // the u-th smallest member takes the slot of the u-th member
        }
    }
    // large groups, one after the other, by the whole workgroup: the g x g comparisons are tiled as
    // (member, slice of the partners) over all threads and the partial counts meet in LDS -- left to
    // the few threads holding the members, a 512-member group would keep them busy for 3000 steps
    // while everyone else idles
    __shared__ uint32_t cless[TAIL_THREADS], ceq[TAIL_THREADS], cbef[TAIL_THREADS];
    __syncthreads();
    const uint32_t nb = nbig;
    for (uint32_t q = 0; q < nb; q++) {
        const uint32_t g = bigs[q] & 0xFFFFu, sz = bigs[q] >> 16; // sz <= TAIL_G <= TAIL_THREADS
        uint32_t c = 1; // partner slices per member: the largest power of two with sz * c <= threads
        while (sz * (c << 1) <= (uint32_t)TAIL_THREADS) c <<= 1;
        if (threadIdx.x < sz) {
            cless[threadIdx.x] = 0;
            ceq[threadIdx.x] = 0;
            cbef[threadIdx.x] = 0;
        }
        __syncthreads();
        const uint32_t mbr = threadIdx.x / c, sl = threadIdx.x % c;
        if (mbr < sz) {
            const uint32_t e = g + mbr;
            const uint32_t per = (sz + c - 1) / c;
            const uint32_t f0 = g + sl * per, f1 = min(g + sz, f0 + per);
            const uint32_t my_hi = A3[e];
            const u64 my_lo = QUAD ? (((u64)A4[e] << 32) | A5[e]) : 0ull;
            uint32_t less = 0, eq_before = 0, eq = 0;
#pragma unroll 4
            for (uint32_t f = f0; f < f1; f++) {
                const uint32_t f_hi = A3[f];
                const u64 f_lo = QUAD ? (((u64)A4[f] << 32) | A5[f]) : 0ull;
                const bool same = f_hi == my_hi && f_lo == my_lo;
                less += (f_hi < my_hi) || (f_hi == my_hi && f_lo < my_lo);
                eq += same;
                eq_before += same && (f < e);
            }
            if (c == 1) {
                cless[mbr] = less;
                ceq[mbr] = eq;
                cbef[mbr] = eq_before;
            } else {
                if (less) atomicAdd(&cless[mbr], less);
                if (eq) atomicAdd(&ceq[mbr], eq);
                if (eq_before) atomicAdd(&cbef[mbr], eq_before);
            }
        }
        __syncthreads();
        if (threadIdx.x < sz) {
            const uint32_t e = g + threadIdx.x, r = A0[e];
            const uint32_t less = cless[threadIdx.x], u = less + cbef[threadIdx.x];
            const bool single = ceq[threadIdx.x] == 1;
            const u64 rec = ((u64)(single ? 1u : 0u) << 60) | ((u64)(r + less) << 40) | ((u64)(r + u) << 20) |
                            (A1[e] & (uint32_t)SUF_MASK);
            out[s_lo + (A1[g + u] >> 20)] = rec;
        }
        __syncthreads();
    }
    // (the block's survivor count is produced by tail_finish: per-wave atomics onto one counter per
    // block cost more than the rest of this kernel)
    if (bad) atomicOr(a.err, 1u);
}

// After tail_sort, one kernel: applies the records of a tile (new ranks / SA entries -- the kernel
// boundary after tail_sort keeps its rank reads consistent), and moves the records that are still
// unresolved, order preserved, back to the block's own buffer: a thread owns 8 consecutive slots,
// a block scan gives the offsets inside the tile, a look-back over the tile counts the offset of the
// tile.  The last tile leaves the block's survivor count for the host.
__global__ void __launch_bounds__(256) tail_finish(TailArgs a)
{
    uint32_t b, tile;
    if (!wg_map(a.T, a.B, b, tile)) return;
    const uint32_t lenw = a.len[b];
    const uint32_t len = lenw & TAIL_LEN;
    const uint32_t r0 = tile * TAIL_T;
    if (r0 >= len) return;
    const size_t base = (size_t)b * a.S;
    const u64 *rec = ((lenw & TAIL_BUF_B) ? a.bufA : a.bufB) + base; // tail_sort's output
    u64 *dst = ((lenw & TAIL_BUF_B) ? a.bufB : a.bufA) + base;       // back home, compacted
    uint32_t *rank = a.rank + base;
    uint32_t *sa = a.sa + base;
    constexpr int PER = TAIL_T / 256, NWV = 256 / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // slot = r0 + k*256 + thread: coalesced reads; survivors keep slot order = (k, wave, lane) order
    __shared__ uint32_t wc[PER * NWV + 1];
    __shared__ uint32_t tpre;
    u64 x[PER];
    uint32_t lo[PER]; // survivors of my wavefront's row k in lower lanes
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t sl = r0 + k * 256 + threadIdx.x;
        x[k] = sl < len ? rec[sl] : LIST_INVALID;
        if (x[k] != LIST_INVALID) {
            const uint32_t i = (uint32_t)(x[k] & SUF_MASK);
            const uint32_t nr = (uint32_t)(x[k] >> 40) & 0xFFFFFu;
            const bool res = (x[k] >> 60) & 1ull;
            rank[i] = res ? (nr | RANK_RESOLVED) : nr;
            // SA entries matter once final: a block in TAIL mode never runs a SWEEP again (the only
            // reader of provisional SA order), so unresolved suffixes write theirs when they resolve
            if (res) {
                sa[(uint32_t)(x[k] >> 20) & 0xFFFFFu] = i;
                x[k] = LIST_INVALID;
            }
        }
        const u64 m = __ballot(x[k] != LIST_INVALID);
        lo[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0) wc[k * NWV + wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int q = 0; q < PER * NWV; q++) { // exclusive scan in slot order
            const uint32_t c = wc[q];
            wc[q] = tot;
            tot += c;
        }
        u64 *st = a.stat + (size_t)b * a.TT; // one word per tail tile
        uint32_t acc = 0, spins = 0;
        if (tile > 0) {
            __hip_atomic_store(st + tile, look_word(a.pass, LOOK_LOCAL, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int t = (int)tile - 1;
            while (t >= 0) {
                const u64 w = __hip_atomic_load(st + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t state = (uint32_t)(w >> 30) & 3u;
                if ((uint32_t)(w >> 32) != a.pass || state == 0) {
                    if (++spins > (1u << 26)) {
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
        }
        __hip_atomic_store(st + tile, look_word(a.pass, LOOK_GLOBAL, acc + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tpre = acc;
        if (r0 + TAIL_T >= len) a.nact_next[b] = acc + tot; // last tile: unresolved suffixes of the block after this round
    }
    __syncthreads();
    const uint32_t pre = tpre;
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (x[k] != LIST_INVALID) dst[pre + wc[k * NWV + wave] + lo[k]] = x[k];
}

// ---- last column ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bwt_emit(Batch bt, uint32_t T, uint32_t B)
{
    uint32_t b, tile;
    if (!wg_map(T, B, b, tile)) return;
    T &= ~WG_SPREAD;
    const uint32_t n = bt.n[b];
    const size_t base = (size_t)b * bt.S;
    const uint8_t *s = bt.rle + base;
    const uint32_t *sa = bt.sa + base;
    uint8_t *out = bt.bwt + base;
    __shared__ uint32_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    // 4 consecutive positions per thread -> one 32-bit store
    for (uint32_t p0 = (tile * 256 + threadIdx.x) * 4; p0 < n; p0 += T * 256 * 4) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t p = p0 + k;
            if (p < n) {
                const uint32_t j = sa[p];
                if (j == 0) bt.ptr[b] = p;
                const uint32_t c = s[j ? j - 1 : n - 1];
                w |= c << (8 * k);
                seen[c] = 1; // every byte of S appears exactly once in the last column
            }
        }
        if (p0 + 3 < n)
            *reinterpret_cast<uint32_t *>(out + p0) = w;
        else
            for (int k = 0; k < 4 && p0 + k < n; k++) out[p0 + k] = (uint8_t)(w >> (8 * k));
    }
    __syncthreads();
    if (seen[threadIdx.x]) bt.hasbyte[(size_t)b * 256 + threadIdx.x] = 1;
}

// Start of a round in one launch: the gates travel as kernel arguments (<= 2 x 256 words) and the
// next round's counters are cleared -- instead of a memset plus a host-to-device copy, each of which
// costs a launch-sized gap.
constexpr uint32_t SETUP_MAX = 256;
struct RoundSetup {
    uint32_t gates[3 * SETUP_MAX]; // [0, mb): radix gates, [mb, 2 mb): tail gates, [2 mb, 3 mb): depth h of TAIL blocks
};
__global__ void __launch_bounds__(256) round_setup(RoundSetup rs, uint32_t *gateR, uint32_t *zero, uint32_t mb, uint32_t dtot_words,
                                                   uint32_t *dtot)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < 3 * mb) gateR[t] = rs.gates[t]; // gateT and gateH follow gateR in memory
    if (t < 2 * mb) zero[t] = 0;
    for (uint32_t k = t; k < dtot_words; k += gridDim.x * 256) dtot[k] = 0; // ACTIVE rounds: digit totals
}

// ---- host driver -----------------------------------------------------------------------------------
// one radix pass = one kernel (look-back scatter)
template <int BITS, int MODE, bool REKEY>
static void launch_pass(bzh_ctx *ctx, SortArgs &a, uint32_t B, uint32_t maxcnt, uint64_t elems)
{
    const uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0) return;
    a.T = tiles | ctx->wgflag;
    a.B = B;
    a.pass++;
    radix_scatter<BITS, MODE, REKEY><<<dim3(xcd_grid(a.T, B)), SORT_THREADS, 0, ctx->stream>>>(a);
    if (ctx->profiling) {
        ctx->stats.bwt_sort_launches += 1;
        ctx->stats.bwt_sort_elems += elems;
    }
}

// Profiling: HIP events bracket each RUN of consecutive radix_scatter launches (the 8 initial passes,
// the 3 of a SWEEP round, the 5 of an ACTIVE round -- nothing else runs in between), not every
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

// One workgroup per block: column sums of refine's digit rows, then the exclusive scan inside each
// of the three digits -> dbase[b][k*128 + d] = first list slot of digit d in SWEEP pass k.
__global__ void __launch_bounds__(768) sweep_bases(RefineArgs a, uint32_t *dbase)
{
    const uint32_t b = blockIdx.x;
    if (a.gate && a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    const uint32_t col = threadIdx.x % 384, seg = threadIdx.x / 384;
    const uint32_t half = (ntile + 1) / 2;
    const uint32_t t0 = seg ? half : 0u, t1 = seg ? ntile : half;
    const uint32_t *p = a.dig + (size_t)b * a.TPB * 512 + col;
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

static void launch_refine(bzh_ctx *ctx, RefineArgs &r, uint32_t B, uint32_t maxcnt)
{
    const uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0) return;
    r.T = tiles | ctx->wgflag;
    r.B = B;
    flag_tiles<<<dim3(xcd_grid(r.T, B)), SORT_THREADS, 0, ctx->stream>>>(r);
    flag_carry<<<dim3(B), 1024, 0, ctx->stream>>>(r);
    if (r.wb)
        refine<true><<<dim3(xcd_grid(r.T, B)), SORT_THREADS, 0, ctx->stream>>>(r);
    else
        refine<false><<<dim3(xcd_grid(r.T, B)), SORT_THREADS, 0, ctx->stream>>>(r);
    if (r.dig) sweep_bases<<<dim3(B), 768, 0, ctx->stream>>>(r, ctx->bt.dbase);
}

// Suffix-sort and emit the last column for blocks 0..B-1 of the batch (bt.rle / bt.n filled).
// nmax = largest block length in the batch, ntotal = sum of block lengths (statistics only).
int bwt_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal)
{
    Batch &bt = ctx->bt;
    if (B == 0) return BZH_OK;
    hipStream_t st = ctx->stream;
    u64 *bufA = reinterpret_cast<u64 *>(bt.listA), *bufB = reinterpret_cast<u64 *>(bt.listB);

    SortArgs a{};
    a.blk = bt.rle;
    a.n = bt.n;
    a.rank = bt.rank;
    a.sa = bt.sa;
    a.headp = bt.headp;
    a.S = bt.S;
    a.TPB = bt.TPB;
    a.h = 0;

    // ---- initial sort on the 8-byte cyclic prefix: LSD, 8 passes of 8 bits.  Passes 0-3 order by
    // bytes 4..7 of the rotation (the key field holds them), the scatter of pass 3 swaps in bytes 0..3,
    // passes 4-7 order by those.  A plain pass over every suffix costs far less than a doubling round
    // does per suffix, so this replaces "4-byte sort + refine + first doubling round".
    // The passes run as single look-back kernels (no histogram / scan launches): their digit bases
    // are the block's byte counts.
    a.cnt = bt.n;
    a.gate = bt.n;
    a.shift = 32;
    a.h = 4; // key offset for GEN_BYTES4
    a.src = nullptr;
    a.dst = bufA;
    a.look = reinterpret_cast<u64 *>(bt.hist);
    a.dbase = bt.dbase;
    a.doff = 0;
    a.err = bt.errflag;
    a.pass = 0;
    HIP_TRY(ctx, hipMemsetAsync(bt.errflag, 0, sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(a.look, 0, (size_t)B * bt.TPB * NBMAX * sizeof(u64), st));
    HIP_TRY(ctx, hipMemsetAsync(bt.alive, 0, (size_t)B * ((bt.S + TAIL_T - 1) / TAIL_T) * sizeof(u64), st));
    HIP_TRY(ctx, hipMemsetAsync(bt.dtot, 0, (size_t)B * DB_STRIDE * sizeof(uint32_t), st));
    byte_count<<<dim3(BYTE_SEGS, B), 1024, 0, st>>>(bt.rle, bt.n, bt.dtot, bt.S);
    active_bases<<<dim3(B), 256, 0, st>>>(bt.dtot, bt.dbase, bt.n, 1);
    ctx->wgflag = B < 8 ? WG_SPREAD : 0u; // the initial sort and its refine run on every block
    hipEvent_t ev_init = span_begin(ctx);
    launch_pass<8, GEN_BYTES4, false>(ctx, a, B, nmax, ntotal);
    u64 *cur = bufA, *oth = bufB;
    for (int p = 1; p < 8; p++) {
        a.shift = 32 + 8 * (p & 3);
        a.src = cur;
        a.dst = oth;
        if (p == 3)
            launch_pass<8, GEN_LIST, true>(ctx, a, B, nmax, ntotal);
        else
            launch_pass<8, GEN_LIST, false>(ctx, a, B, nmax, ntotal);
        u64 *t = cur;
        cur = oth;
        oth = t;
    }
    span_end(ctx, ev_init);
    a.h = 0;

    // three rotating count arrays: length of the list in `cur` (prevcnt), unresolved counts of the
    // round being sorted (nact), counts that round's refine accumulates (nact_next)
    uint32_t *cnts[3] = {bt.nactA, bt.nactB, bt.nactC};
    int inext = 0;
    uint32_t *nact = nullptr, *nact_next = cnts[inext];
    const uint32_t mb = ctx->max_batch; // pair layout: counts at +0, largest group at +mb
    HIP_TRY(ctx, hipMemsetAsync(nact_next, 0, 2 * mb * sizeof(uint32_t), st));

    RefineArgs r{};
    r.n = bt.n;
    r.cnt = bt.n;
    r.list = cur;
    r.blk = bt.rle;
    r.rank = bt.rank;
    r.sa = bt.sa;
    r.headp = bt.headp;
    r.flg = bt.flg;
    r.tagg = bt.tagg;
    r.nact_next = nact_next;
    r.maxgrp = nact_next + mb;
    r.S = bt.S;
    r.TPB = bt.TPB;
    r.init = 1;
    r.wb = nullptr; // the first round reads the list in its initial-sort format (every suffix, unranked)
    r.cstat = reinterpret_cast<u64 *>(bt.hist);
    r.err = bt.errflag;
    r.dig = bt.hist; // the first round may be a SWEEP
    launch_refine(ctx, r, B, nmax);

    // ---- doubling rounds ---------------------------------------------------------------------------
    uint32_t *hact = ctx->h_pinned;       // unresolved counts read back this round
    uint32_t *hmax = ctx->h_pinned + mb;     // largest group per block after the last radix round
    uint32_t *hn = ctx->h_pinned + 2 * mb;   // block lengths
    uint32_t *hgR = ctx->h_pinned + 3 * mb;  // gates uploaded each round
    uint32_t *hgT = ctx->h_pinned + 4 * mb;
    uint32_t *hgH = ctx->h_pinned + 5 * mb;  // depth of every TAIL block
    HIP_TRY(ctx, hipMemcpyAsync(hn, bt.n, B * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    std::vector<uint32_t> hprev(B), taillen(B, 0); // frozen slots once in TAIL mode
    std::vector<uint8_t> tailmode(B, 0);
    std::vector<uint32_t> htail(B, 0); // a TAIL block's own depth: it may advance x4 while the radix path doubles
    bool active_mode = false, have_n = false;
    bool have_list = true; // `cur` holds the list ACTIVE / TAIL need (the initial one, or refine's compacted one)
    const uint32_t *prevcnt = nullptr; // (the list in `cur` is dense after the first round: its length is nact)
    uint32_t h = 8; // the initial sort ordered the rotations by their first 8 bytes
    TailArgs ta{};
    ta.n = bt.n;
    ta.len = bt.gateT;
    ta.bufA = bufA;
    ta.bufB = bufB;
    ta.rank = bt.rank;
    ta.sa = bt.sa;
    ta.err = bt.errflag;
    ta.stat = reinterpret_cast<u64 *>(bt.alive);
    ta.S = bt.S;
    ta.TT = (bt.S + TAIL_T - 1) / TAIL_T; // <= 512 (S <= 2^20)
    for (int round = 0; round < 48; round++) {
        HIP_TRY(ctx, hipMemcpyAsync(hact, nact_next, 2 * mb * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, bzh_stream_wait(st));
        if (!have_n) {
            for (uint32_t b = 0; b < B; b++) hprev[b] = hn[b];
            have_n = true;
        }
        // per-block mode: a block whose groups all fit a tail window leaves the radix path for good
        uint32_t maxact = 0, prevmax = 0, maxtail = 0;
        uint64_t sum = 0, nsum = 0, tot = 0, tailtot = 0;
        uint32_t maxh = 0, nrad_act = 0, ntail_act = 0;
        for (uint32_t b = 0; b < B; b++) {
            tot += hact[b];
            if (!tailmode[b] && have_list && hact[b] && hmax[b] <= (uint32_t)TAIL_G) {
                tailmode[b] = 1;
                htail[b] = h;
                // the sorted list in `cur` keeps this many slots, in this buffer, from now on
                taillen[b] = (round == 0 ? hn[b] : hact[b]) | (cur == bufB ? TAIL_BUF_B : 0u);
            }
            if (tailmode[b] == 2) // compacted last round (back into its own buffer)
                taillen[b] = hact[b] | (taillen[b] & TAIL_BUF_B);
            if (tailmode[b]) {
                tailmode[b] = 2;
                hgR[b] = 0;
                hgT[b] = hact[b] ? taillen[b] : 0;
                hgH[b] = htail[b];
                tailtot += hact[b];
                ntail_act += hact[b] != 0;
                if (hact[b]) maxh = htail[b] > maxh ? htail[b] : maxh;
                maxtail = (hgT[b] & TAIL_LEN) > maxtail ? (hgT[b] & TAIL_LEN) : maxtail;
            } else {
                hgR[b] = hact[b];
                hgT[b] = 0;
                hgH[b] = 0;
                maxact = hact[b] > maxact ? hact[b] : maxact;
                nrad_act += hact[b] != 0;
                if (hact[b]) {
                    const uint32_t L = round == 0 ? hn[b] : hact[b]; // length of the list in `cur`
                    prevmax = L > prevmax ? L : prevmax;
                    sum += hact[b];
                    nsum += hn[b];
                }
                hprev[b] = hact[b]; // length of the list this round's sort produces
            }
        }
        if (tot == 0) break;
        static const bool trace = getenv("BZH_TRACE_ROUNDS") != nullptr;
        if (trace) { // diagnostic: group-size picture of the round about to run
            uint32_t nrad = 0, ntail = 0, le512 = 0, le1k = 0, le4k = 0;
            for (uint32_t b = 0; b < B; b++) {
                if (!hact[b]) continue;
                if (tailmode[b]) {
                    ntail++;
                    continue;
                }
                nrad++;
                le512 += hmax[b] <= 512;
                le1k += hmax[b] <= 1024;
                le4k += hmax[b] <= 4096;
            }
            fprintf(stderr, "[bzhip] round %d h=%u unresolved=%llu radix blocks=%u (maxgrp<=512:%u <=1k:%u <=4k:%u) tail blocks=%u\n",
                    round, h, (unsigned long long)tot, nrad, le512, le1k, le4k, ntail);
        }
        ctx->stats.bwt_active_sum += tot;
        ctx->stats.bwt_rounds = (uint64_t)(round + 1) > ctx->stats.bwt_rounds ? (uint64_t)(round + 1) : ctx->stats.bwt_rounds;
        { // rotate: this round's counts stay readable next round as the length of `cur`
            const int icur = inext;
            int inew = (icur + 1) % 3;
            if (cnts[inew] == prevcnt) inew = (icur + 2) % 3;
            nact = cnts[icur];
            nact_next = cnts[inew];
            inext = inew;
        }
        // ACTIVE costs ~5 list passes over the unresolved suffixes, SWEEP a full SA sweep plus 3
        // passes: switch once the unresolved fraction is small; never switch back.
        if (!active_mode && have_list && nsum && sum * 3 < nsum) active_mode = true;
        const bool active_round = maxact && active_mode;
        if (mb <= SETUP_MAX) {
            RoundSetup rs;
            memcpy(rs.gates, hgR, 3 * mb * sizeof(uint32_t)); // hgT and hgH follow hgR in the pinned block
            const uint32_t dw = active_round ? B * DB_STRIDE : 0u;
            round_setup<<<dim3(active_round ? 64 : (3 * mb + 255) / 256), 256, 0, st>>>(rs, bt.gateR, nact_next, mb, dw, bt.dtot);
        } else {
            HIP_TRY(ctx, hipMemsetAsync(nact_next, 0, 2 * mb * sizeof(uint32_t), st));
            HIP_TRY(ctx, hipMemcpyAsync(bt.gateR, hgR, 3 * mb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            if (active_round) HIP_TRY(ctx, hipMemsetAsync(bt.dtot, 0, (size_t)B * DB_STRIDE * sizeof(uint32_t), st));
        }

        ctx->wgflag = nrad_act < 8 ? WG_SPREAD : 0u; // radix launches of this round
        a.h = h;
        a.recrank = round > 0; // every refine after the initial one writes ranks back into the list
        a.gate = bt.gateR;
        // a TAIL round may look three h-blocks ahead: depth 4h instead of 2h (three gathers per suffix:
        // only once few suffixes are left in TAIL blocks, where rounds are latency-bound).  TAIL blocks
        // carry their own depth, so blocks still on the radix path do not hold them back.
        const bool quad = maxh < (1u << 28) && tailtot * 10 < ntotal;
        u64 *next_cur = cur, *next_oth = oth;
        if (!maxact) {
            // every unresolved block is in TAIL mode
        } else if (!active_mode) {
            // three look-back passes; the last refine left the digit bases (sweep_bases)
            a.cnt = bt.n; // enumerate SA positions
            a.shift = 40;
            a.doff = 0;
            a.src = nullptr;
            a.dst = bufA;
            hipEvent_t ev = span_begin(ctx);
            launch_pass<7, GEN_SWEEP, false>(ctx, a, B, nmax, sum);
            a.cnt = nact;
            a.shift = 47;
            a.doff = 128;
            a.src = bufA;
            a.dst = bufB;
            launch_pass<7, GEN_LIST, false>(ctx, a, B, maxact, sum);
            a.shift = 54;
            a.doff = 256;
            a.src = bufB;
            a.dst = bufA;
            launch_pass<7, GEN_LIST, false>(ctx, a, B, maxact, sum);
            span_end(ctx, ev);
            next_cur = bufA;
            next_oth = bufB;
        } else {
            // re-key once (active_gen), then five look-back passes on bits 20..59; the list ends in `cur`
            {
                const uint32_t gt = (prevmax + SORT_TILE - 1) / SORT_TILE;
                a.cnt = round == 0 ? bt.n : nact; // the initial list holds every suffix, later lists are dense
                a.src = cur;
                a.dst = oth;
                a.T = gt | ctx->wgflag;
                a.B = B;
                if (gt) {
                    active_gen<<<dim3(xcd_grid(a.T, B)), SORT_THREADS, 0, st>>>(a, bt.dtot);
                    active_bases<<<dim3(B), 256, 0, st>>>(bt.dtot, bt.dbase, bt.gateR, 5);
                }
                u64 *c = oth, *o = cur;
                a.shift = 20;
                a.doff = 0;
                a.src = c;
                a.dst = o;
                hipEvent_t ev = span_begin(ctx);
                launch_pass<8, GEN_LISTH, false>(ctx, a, B, prevmax, sum); // skips the holes
                a.cnt = nact;
                for (int p = 1; p < 5; p++) {
                    u64 *t = c;
                    c = o;
                    o = t;
                    a.shift = 20 + 8 * p;
                    a.doff = 256 * p;
                    a.src = c;
                    a.dst = o;
                    launch_pass<8, GEN_LIST, false>(ctx, a, B, maxact, sum);
                }
                span_end(ctx, ev);
                // gen: cur -> oth; passes: oth -> cur -> oth -> cur -> oth -> cur
                next_cur = cur;
                next_oth = oth;
            }
        }
        if (maxtail) { // blocks in TAIL mode: in place in their own buffer, independent of cur/oth
            ta.nact_next = nact_next;
            ta.hb = bt.gateR + 2 * mb;
            ta.recrank = round > 0;
            ta.T = ((maxtail + TAIL_T - 1) / TAIL_T) | (ntail_act < 8 ? WG_SPREAD : 0u);
            ta.B = B;
            if (quad)
                tail_sort<true><<<dim3(xcd_grid(ta.T, B)), TAIL_THREADS, 0, st>>>(ta);
            else
                tail_sort<false><<<dim3(xcd_grid(ta.T, B)), TAIL_THREADS, 0, st>>>(ta);
            ta.pass = ++a.pass;
            tail_finish<<<dim3(xcd_grid(ta.T, B)), 256, 0, st>>>(ta);
        }
        if (maxact) {
            cur = next_cur;
            oth = next_oth;
        }

        if (maxact) {
            r.cnt = nact;
            r.list = cur;
            r.nact_next = nact_next;
            r.maxgrp = nact_next + mb;
            r.init = 0;
            r.dig = active_mode ? nullptr : bt.hist; // only a SWEEP round needs the digit bases
            // The compacted list is only read by ACTIVE re-keying and by blocks entering TAIL mode.  While
            // nearly everything is still unresolved (periodic inputs: many SWEEP rounds in a row) it is not
            // written; without it the next round is a SWEEP and no block changes mode (always correct).
            const bool wbk = active_mode || sum * 10 <= nsum * 9;
            r.wb = wbk ? oth : nullptr; // the still unresolved suffixes, ranked, compacted, in order
            r.cpass = ++a.pass;
            r.gate = bt.gateR;
            launch_refine(ctx, r, B, maxact);
            if (wbk) { // the compacted list is the next round's `cur`
                u64 *t = cur;
                cur = oth;
                oth = t;
            }
            have_list = wbk;
        }

        for (uint32_t b = 0; b < B; b++)
            if (tailmode[b] && htail[b] < (1u << 30)) htail[b] <<= quad ? 2 : 1;
        if (h < (1u << 30)) h <<= 1;
    }
    {
        uint32_t err = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&err, bt.errflag, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, bzh_stream_wait(st));
        if (err) {
            bzh_set_error(ctx, err & 2 ? "BWT initial sort: a look-back gave up waiting (internal error)"
                                       : "BWT tail rounds saw a group larger than their window (internal error)");
            return BZH_E_HIP;
        }
    }

    HIP_TRY(ctx, hipMemsetAsync(bt.hasbyte, 0, (size_t)B * 256, st));
    uint32_t gx = (nmax + 1023) / 1024;
    if (gx > 256) gx = 256;
    if (gx == 0) gx = 1;
    if (B < 8) gx |= WG_SPREAD;
    bwt_emit<<<dim3(xcd_grid(gx, B)), 256, 0, st>>>(bt, gx, B);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
