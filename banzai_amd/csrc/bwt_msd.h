// bwt_msd.h -- bucket-first initial sort of the suffix sorter (included by bwt.hip; gfx950 only).
//
// Replaces, for text-like blocks, the 8 global radix passes + re-key gather + refine_one<init> of the initial sort
// (reference contract unchanged: lib/bwt.rs:526-756, tie rule :564-573 -- the initial sort only has to leave an
// order-consistent ranking of the rotations by a prefix of known minimum depth, the doubling rounds do the rest).
//
//   bigram_hist      65,536-bin histogram of the block's 2-byte cyclic prefixes (packed 16-bit LDS counters)
//   bigram_plan      one workgroup per block: bucket starts (scan); which blocks qualify -- a sampled "is this block
//                    repetitive" test (such blocks keep the 8-pass path and its SWEEP mode) and at most MS_NE_MAX
//                    non-empty buckets (text-like: random / binary blocks keep it too); the work list: buckets of at
//                    most MS_TILE suffixes become UNITS (neighbours packed together greedily), larger ones go to level 1
//   bigram_scatter   ONE pass from the text to 2-byte buckets: a tile's suffixes are ordered by bigram inside LDS
//                    (two local counting passes), every run of equal bigrams claims its room in the bucket with one
//                    atomic add (a first pass has no order to keep) and leaves as a coalesced store.
//                    element = [bytes 2..6 of the rotation : 40 @20][suffix : 20]
//   seg_count / seg_plan / seg_scatter   level L = 1..5: every bucket still larger than a tile is split by byte
//                    1 + L of the rotation (same claim scheme), its sub-buckets become units or level L+1 buckets; what
//                    is still oversized after byte 6 has all 7 bytes equal: one group, nothing left to sort
//   chunk_finish     one workgroup per unit: up to 5 + 1 counting passes inside LDS (passes whose digit does not vary
//                    inside the unit are skipped), group heads and sizes from ballots, then exactly what
//                    refine_one<init> leaves: (rank word, suffix) pairs binned by 4096-suffix window of the rank array
//                    (rank_apply turns them into whole lines), small groups to the small-group list, large groups to
//                    the big list.  Units are whole buckets, so no group crosses a unit: no carries, no look-back.
// Depth: 2 + 5 = 7 bytes, so a block on this path enters the doubling rounds at h = 7 (round_begin).
#pragma once

constexpr int MS_THREADS = 512, MS_ITEMS = 16, MS_TILE = 8192, MS_NW = 8;
constexpr uint32_t MS_BG = 65536;
constexpr uint32_t MS_SAMPLES = 4096;
enum : uint32_t { MC_UNITS = 0, MC_TICKET = 1, MC_PLAN_DONE = 2, MC_OLD = 3, MC_NEW = 4, MC_SEGS = 8, MC_ITEMS = 16 };
constexpr uint32_t ERR_MSD = 8u; // bit 3 of the error word: an invariant of this file did not hold

struct Msd {
    const uint8_t *blk;  // [B][S]
    const uint32_t *n;   // [B]
    uint32_t S, B;
    uint32_t *bgcur, *pool, *segcur;
    uint4 *units, *segs;
    uint32_t *items, *cnt, *np, *act_old, *act_new, *bincur;
    u64 *bufX, *bufY;    // partition output of levels 0, 2, 4 / 1, 3, 5
    u64 *big, *tail, *binned;
    uint8_t *bwt;        // [B][S] the last column (chunk_finish writes the bytes of the rotations it resolves)
    uint32_t *c_big, *c_small, *c_groups;
    uint32_t *err;
    uint32_t midcap;     // records a tile of mid_sort holds at most (mid_plan; a longer run is a tile of its own)
    // the big lists' runs (bwt.hip: msc_*): the counters records | runs << 20 and the table chunk_finish fills (round 0's
    // half); mid_plan: the half it reads, the tiles and their number per block it writes, mid_sort's tickets it clears
    uint32_t *runq;
    uint2 *runs;
    uint2 *tiles;
    uint32_t *ntiles, *mticket;
    GidOut gout;         // numbers for the large groups (bwt.hip: round 0 sorts its big lists on them)
    uint32_t force_old;  // every block keeps the 8-pass path (BZH_INIT=lsd)
    uint32_t force_new;  // no block is kept off the buckets for its share of oversized ones (BZH_INIT=msd)
    uint32_t fuse;       // chunk_finish also takes the first doubling step of the small groups (round_begin: r0_fused)
    uint32_t dbg;        // timing experiments only (BZH_MSD_DBG): 16 = cycles per phase of chunk_finish
};

__device__ __forceinline__ uint32_t *ms_seg_start_row(const Msd &m, uint32_t L, uint32_t b, uint32_t slot)
{
    return m.pool + (size_t)m.B * MS_BG_ROW + (((size_t)(L - 1) * m.B + b) * MS_SEG_SLOTS + slot) * MS_SEG_ROW;
}
__device__ __forceinline__ uint32_t *ms_seg_cur_row(const Msd &m, uint32_t L, uint32_t b, uint32_t slot)
{
    return m.segcur + (((size_t)(L - 1) * m.B + b) * MS_SEG_SLOTS + slot) * 256;
}
__device__ __forceinline__ uint32_t *ms_slot_counter(const Msd &m, uint32_t L, uint32_t b) { return m.cnt + MS_CNT_WORDS + (size_t)L * m.B + b; }
// unit record: x = block | buffer << 10 | uniform << 11 | table entries << 12, y = first position, z = end, w = table offset
// in the pool (uniform units: first position of the whole group)
__device__ __forceinline__ uint4 ms_unit(uint32_t b, uint32_t buf, uint32_t uniform, uint32_t nb, uint32_t s, uint32_t e, uint32_t tbl)
{
    return make_uint4(b | (buf << 10) | (uniform << 11) | (nb << 12), s, e, tbl);
}

// ---- one stable counting pass inside a tile held in registers ------------------------------------------------
// Item (wave w, step k, lane l) is element w * R * 64 + k * 64 + l of the tile's current order (R = steps per wave,
// the same for every wave).  Digit of an item: bits sh .. sh+7 of v[k].  Leaves in pos (16 bits per item) the
// item's slot in the new order.  The ranking is radix_scatter's: match-any by ballots, per-wave LDS counters.
// Counters: 16 bits are enough (a tile holds 8192 elements), two sets: a pass clears the set of the NEXT pass while it
// ranks, so a pass costs three barriers (the first pass of a tile needs `cur` cleared by the caller).
typedef uint16_t MsCnt[MS_NW][256];
__device__ __forceinline__ void ms_clear(MsCnt &c, uint32_t tid)
{
    reinterpret_cast<uint2 *>(&c[0][0])[tid] = make_uint2(0u, 0u); // 512 threads x 8 bytes = 4 KB
}
__device__ __forceinline__ void ms_clear(MsCnt &c) { ms_clear(c, threadIdx.x); }

template <int NBITS, typename T>
__device__ __forceinline__ void tile_rank(const T (&v)[MS_ITEMS], int sh, uint32_t actmask, int R, MsCnt &cur, MsCnt &nxt, uint32_t *ls,
                                          uint32_t (&pos)[MS_ITEMS / 2], uint32_t tid)
{
    const int wave = (int)(tid >> 6), lane = (int)(tid & 63u);
    constexpr uint32_t dmask = (1u << NBITS) - 1u;
    ms_clear(nxt, tid);
    uint32_t wr[MS_ITEMS / 2];
#pragma unroll
    for (int k = 0; k < MS_ITEMS / 2; k++) wr[k] = 0;
    // (tried: the match masks of four steps side by side, then their counter updates -- more independent work per
    // wavefront, but 7 % slower: the masks of four steps cost more registers than the schedule gains)
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++) {
        if (k < R) {
            const bool act = (actmask >> k) & 1u;
            const uint32_t d = (uint32_t)(v[k] >> sh) & dmask;
            const u64 m0 = __ballot(act);
            uint32_t mlo = (uint32_t)m0, mhi = (uint32_t)(m0 >> 32);
            // (tried: comparing only the bits that vary inside the tile -- text leaves two or three bits of a byte constant
            // inside a bucket -- with a loop over the set bits of a mask: the loop costs more than the skipped rounds save)
#pragma unroll
            for (int bit = 0; bit < NBITS; bit++) { // keep the lanes whose digit agrees with mine in this bit: m & ~(ballot ^ mine)
                const int om = __builtin_amdgcn_sbfe((int)d, bit, 1);
                const u64 bm = __builtin_amdgcn_ballot_w64(om != 0);
                mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bm, (uint32_t)om, 0x90);
                mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bm >> 32), (uint32_t)om, 0x90);
            }
            if (act) {
                const uint32_t before = cur[wave][d];
                const uint32_t off = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                if (off == 0) cur[wave][d] = (uint16_t)(before + (uint32_t)(__popc(mlo) + __popc(mhi)));
                wr[k >> 1] |= (before + off) << (16 * (k & 1));
            }
            asm volatile("" ::: "memory"); // (LDS operations of one wavefront execute in order; see radix_scatter)
        }
    }
    __syncthreads();
    // digit totals -> exclusive starts (threads 0..255 = digits: four wavefronts scan, their sums meet in LDS)
    uint32_t mytot = 0, inc = 0;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < MS_NW; w++) mytot += cur[w][tid];
        inc = wave_incl_add(mytot, lane);
        if (lane == 63) ls[wave] = inc;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t g = inc - mytot;
        for (int w = 0; w < wave; w++) g += ls[w];
#pragma unroll
        for (int w = 0; w < MS_NW; w++) {
            const uint32_t t = cur[w][tid];
            cur[w][tid] = (uint16_t)g;
            g += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MS_ITEMS / 2; k++) pos[k] = 0;
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++) {
        if (k < R && ((actmask >> k) & 1u)) pos[k >> 1] |= ((uint32_t)cur[wave][(uint32_t)(v[k] >> sh) & dmask] + ((wr[k >> 1] >> (16 * (k & 1))) & 0xFFFFu)) << (16 * (k & 1));
    }
}

// The FIRST counting pass of a tile has no order to keep (there is no lower digit yet): an element's slot among the
// elements with its digit may be any -- one LDS atomic per element instead of the ballot ranking (about 100 vector
// instructions a thread instead of 1,500).  `cnt`: 256 words.  Three barriers; the caller writes the stage behind it.
template <typename T>
__device__ __forceinline__ void tile_rank_unordered(const T (&v)[MS_ITEMS], int sh, uint32_t actmask, int R, uint32_t *cnt, uint32_t *ls,
                                                    uint32_t (&pos)[MS_ITEMS / 2])
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++)
        if (k < R && ((actmask >> k) & 1u)) atomicAdd(&cnt[(uint32_t)(v[k] >> sh) & 255u], 1u);
    __syncthreads();
    uint32_t c = 0, inc = 0;
    if (threadIdx.x < 256) {
        c = cnt[threadIdx.x];
        inc = wave_incl_add(c, lane);
        if (lane == 63) ls[wave] = inc;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t g = inc - c;
        for (int w = 0; w < wave; w++) g += ls[w];
        cnt[threadIdx.x] = g; // from here on: the digit's cursor
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MS_ITEMS / 2; k++) pos[k] = 0;
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++)
        if (k < R && ((actmask >> k) & 1u)) pos[k >> 1] |= atomicAdd(&cnt[(uint32_t)(v[k] >> sh) & 255u], 1u) << (16 * (k & 1));
}

// A run of equal keys inside one row of 64 sorted items claims its room with ONE atomic add on the key's cursor.
// `head`: this lane starts a run; `nact`: active lanes of the row (a prefix).  Returns the lane's destination.
__device__ __forceinline__ uint32_t row_claim(bool act, bool head, uint32_t *cursor, int lane)
{
    const u64 hm = __ballot(head), am = __ballot(act);
    const u64 upto = (2ull << lane) - 1ull; // bits 0..lane (lane 63: all)
    const u64 below = hm & upto, above = hm & ~upto;
    const int myhead = below ? 63 - __clzll((long long)below) : 0;
    const int nexth = above ? __ffsll((long long)above) - 1 : (int)__popcll(am);
    uint32_t base = 0;
    if (act && head) base = atomicAdd(cursor, (uint32_t)(nexth - lane));
    base = (uint32_t)__shfl((int)base, myhead, 64);
    return base + (uint32_t)(lane - myhead);
}

// ---- 2-byte histogram -------------------------------------------------------------------------------------------
constexpr int BGH_SEGS = 16; // a workgroup counts at most ceil(n / 16) + 16 < 65536 positions: 16-bit counters cannot overflow
__global__ void __launch_bounds__(1024) bigram_hist(Msd m)
{
    const uint32_t b = blockIdx.y, n = m.n[b];
    if (m.force_old || n == 0) return;
    const uint8_t *s = m.blk + (size_t)b * m.S;
    const uint32_t per = ((n + BGH_SEGS - 1) / BGH_SEGS + 15u) & ~15u;
    const uint32_t lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    if (lo >= hi) return;
    __shared__ uint32_t h[MS_BG / 2];
    for (int k = threadIdx.x * 4; k < (int)(MS_BG / 2); k += 4096) *reinterpret_cast<uint4 *>(&h[k]) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // 16 positions a thread and step: one aligned 16-byte load + the byte behind it; equal neighbouring bigrams
    // (a run of one byte) share one add
    for (uint32_t i = lo + threadIdx.x * 16; i < hi; i += 16384) {
        if (i + 17 <= hi || (i + 16 <= hi && i + 16 < n)) {
            const uint4 q = *reinterpret_cast<const uint4 *>(s + i); // (lo and the block base are 16-byte aligned)
            const uint32_t w[5] = {q.x, q.y, q.z, q.w, (uint32_t)s[i + 16]};
            uint32_t curk = 0xFFFFFFFFu, cnt = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t c0 = (w[k >> 2] >> ((k & 3) * 8)) & 255u, c1 = (w[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 255u;
                const uint32_t key = (c0 << 8) | c1;
                if (key != curk) {
                    if (cnt) atomicAdd(&h[curk >> 1], cnt << (16 * (curk & 1)));
                    curk = key;
                    cnt = 0;
                }
                cnt++;
            }
            atomicAdd(&h[curk >> 1], cnt << (16 * (curk & 1)));
        } else {
            for (uint32_t j = i; j < min(hi, i + 16); j++) {
                const uint32_t key = ((uint32_t)s[j] << 8) | s[j + 1 < n ? j + 1 : 0];
                atomicAdd(&h[key >> 1], 1u << (16 * (key & 1)));
            }
        }
    }
    __syncthreads();
    uint32_t *cur = m.bgcur + (size_t)b * MS_BG;
    for (int w = threadIdx.x; w < (int)(MS_BG / 2); w += 1024) {
        const uint32_t v = h[w];
        if (v & 0xFFFFu) atomicAdd(&cur[2 * w], v & 0xFFFFu);
        if (v >> 16) atomicAdd(&cur[2 * w + 1], v >> 16);
    }
}

// ---- plan of a block ----------------------------------------------------------------------------------------------
// Units are packed greedily over the non-empty buckets in order: a bucket joins the unit before it unless the unit
// would exceed MS_TILE suffixes (or, 2-byte level, 256 buckets: the bucket index is one more 8-bit digit); a bucket
// of more than MS_TILE suffixes is a unit of its own kind: an oversized bucket for the next level.
constexpr uint32_t MS_OVER_PCT = 35;
constexpr uint32_t MS_NE_MAX = 8192; // non-empty 2-byte buckets the greedy packing holds in LDS (text has ~1,000-5,000);
                                     // a block with more (random / binary data) keeps the 8-pass path

// Appends an oversized bucket [s, e) of block b to level L: slot of the block, cleared digit counters, its tiles.
// Called by ONE thread per bucket.
__device__ __forceinline__ void ms_push_seg(const Msd &m, uint32_t L, uint32_t b, uint32_t s, uint32_t e)
{
    const uint32_t slot = atomicAdd(ms_slot_counter(m, L, b), 1u);
    if (slot >= MS_SEG_SLOTS) { // more than n / MS_TILE oversized buckets in one level: impossible
        atomicOr(m.err, ERR_MSD);
        return;
    }
    const uint32_t g = atomicAdd(&m.cnt[MC_SEGS + L], 1u);
    m.segs[(size_t)L * m.B * MS_SEG_SLOTS + g] = make_uint4(b | (slot << 10), s, e, 0u);
    uint32_t *row = ms_seg_cur_row(m, L, b, slot);
    for (int d = 0; d < 256; d += 4) *reinterpret_cast<uint4 *>(row + d) = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t tiles = (e - s + MS_TILE - 1) / MS_TILE;
    const uint32_t i0 = atomicAdd(&m.cnt[MC_ITEMS + L], tiles);
    if (i0 + tiles > m.B * MS_ITEM_CAP) {
        atomicOr(m.err, ERR_MSD);
        return;
    }
    for (uint32_t t = 0; t < tiles; t++) m.items[(size_t)L * m.B * MS_ITEM_CAP + i0 + t] = g | (t << 20);
}

// The units of a block form a list of their own (row 0 of the per-block counters: its length, row MS_LEVELS + 1: the
// tickets chunk_finish hands out over it): the finishing kernel works through a block's units on ONE XCD.
__host__ __device__ __forceinline__ uint32_t *ms_unit_count(const Msd &m, uint32_t b) { return m.cnt + MS_CNT_WORDS + b; }
__device__ __forceinline__ uint32_t *ms_unit_ticket(const Msd &m, uint32_t b) { return m.cnt + MS_CNT_WORDS + (size_t)(MS_LEVELS + 1) * m.B + b; }
// (rows MS_LEVELS + 2 ..: bwt.hip, msc_row -- "the block holds a group that spans several units", i.e. more than MS_TILE
// rotations share 7 bytes; the tickets and tile counts of mid_sort; the run counters of the big lists)
__host__ __device__ __forceinline__ uint32_t *ms_spans(const Msd &m, uint32_t b) { return msc_row(m.cnt, m.B, MSR_SPANS) + b; }
__device__ __forceinline__ void ms_push_unit(const Msd &m, uint4 u)
{
    const uint32_t b = u.x & 1023u;
    const uint32_t i = atomicAdd(ms_unit_count(m, b), 1u);
    if (i >= MS_UNIT_CAP) {
        atomicOr(m.err, ERR_MSD);
        return;
    }
    atomicAdd(&m.cnt[MC_UNITS], 1u); // (trace only)
    m.units[(size_t)b * MS_UNIT_CAP + i] = u;
}

__global__ void __launch_bounds__(1024) bigram_plan(Msd m, uint32_t *hsum, uint32_t seq)
{
    const uint32_t b = blockIdx.x, n = m.n[b], tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ u64 tab[MS_NE_MAX];   // the sample test's hash set, then the non-empty buckets: [bucket:16 @40][size:20 @20][start:20]
    __shared__ uint2 heads[2048];    // unit heads: (start, bucket | oversized << 17)
    __shared__ uint32_t ls[20];
    __shared__ uint32_t s_distinct, s_nheads;
    bool np = !m.force_old && n > 0 && (n + BGH_SEGS - 1) / BGH_SEGS + 16u < 65536u;
    const uint8_t *s = m.blk + (size_t)b * m.S;
    if (np && n >= 32768u) {
        // Repetitive blocks (runs, a tile laid over and over) are better off with the 8-pass path and its SWEEP mode:
        // sample 4096 rotations; if fewer than half of their 8-byte prefixes are distinct (fewer than ~2,500 equally
        // frequent 8-byte strings in the whole block) the block stays there.
        for (int k = tid; k < (int)MS_NE_MAX; k += 1024) tab[k] = 0ull;
        if (tid == 0) s_distinct = 0;
        __syncthreads();
        const uint32_t stride = n / MS_SAMPLES;
        uint32_t mine = 0;
        u64 v[MS_SAMPLES / 1024];
#pragma unroll
        for (uint32_t j = 0; j < MS_SAMPLES / 1024; j++) {
            const uint32_t p = (tid * (MS_SAMPLES / 1024) + j) * stride;
            v[j] = 0;
            if (p + 8 <= n) {
                __builtin_memcpy(&v[j], s + p, 8);
            } else {
                for (uint32_t q = 0; q < 8; q++) v[j] |= (u64)s[(p + q) % n] << (8 * q);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < MS_SAMPLES / 1024; j++) {
            u64 x = v[j] ^ 0xA5A5A5A5A5A5A5A5ull;
            if (x == 0) x = 1;
            uint32_t slot = (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 51) & (MS_NE_MAX - 1);
            for (;;) {
                const u64 old = atomicCAS((unsigned long long *)&tab[slot], 0ull, (unsigned long long)x);
                if (old == 0ull) {
                    mine++;
                    break;
                }
                if (old == x) break;
                slot = (slot + 1) & (MS_NE_MAX - 1);
            }
        }
        mine = wave_reduce_add(mine);
        if (lane == 0 && mine) atomicAdd(&s_distinct, mine);
        __syncthreads();
        np = (uint64_t)s_distinct * 2u >= (uint64_t)MS_SAMPLES;
    }
    uint32_t NE = 0, exT = 0, exNE = 0;
    uint32_t *cur = m.bgcur + (size_t)b * MS_BG;
    const uint32_t k0 = tid * 64;
    if (np) {
        // my 64 buckets -- suffixes, non-empty buckets
        uint32_t tot = 0, ne = 0, big = 0;
        for (int q = 0; q < 64; q += 4) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(cur + k0 + q);
            tot += c4.x + c4.y + c4.z + c4.w;
            ne += (c4.x != 0u) + (c4.y != 0u) + (c4.z != 0u) + (c4.w != 0u);
            big += (c4.x > (uint32_t)MS_TILE ? c4.x : 0u) + (c4.y > (uint32_t)MS_TILE ? c4.y : 0u) + (c4.z > (uint32_t)MS_TILE ? c4.z : 0u) +
                   (c4.w > (uint32_t)MS_TILE ? c4.w : 0u);
        }
        uint32_t all, bigall;
        exT = block_excl_add(tot, ls, &all);
        exNE = block_excl_add(ne, ls, &NE);
        (void)block_excl_add(big, ls, &bigall);
        if (all != n && tid == 0) atomicOr(m.err, ERR_MSD);
        if (tid == 0) atomicAdd(&m.cnt[23], bigall >> 10); // (trace: suffixes in oversized 2-byte buckets, in units of 1024)
        // Text has a few thousand distinct 2-byte prefixes; a block with more than MS_NE_MAX of them (random or binary
        // data: up to all 65,536, tiny) gains nothing from buckets and keeps the 8-pass path as well.  So does a block
        // that would send more than MS_OVER_PCT per cent of its suffixes through the level-by-level split of oversized
        // buckets (a small alphabet: few, large buckets -- every level is one more pass over them; measured: 10 % and 13 % on
        // the headline and on real text, where the buckets win by 4-7 %, 26 % on 57-letter text, where they lose 1.4 %).
        np = NE <= MS_NE_MAX && ((uint64_t)bigall * 100u <= (uint64_t)n * MS_OVER_PCT || m.force_new);
    }
    if (tid == 0) {
        m.np[b] = np ? 1u : 0u;
        const uint32_t k = atomicAdd(&m.cnt[np ? MC_NEW : MC_OLD], 1u);
        (np ? m.act_new : m.act_old)[k] = b;
    }
    if (np) {
        uint32_t *st = m.pool + (size_t)b * MS_BG_ROW;
        // pass B: bucket starts (kept for the finishing kernel) = claim cursors of the partition; the non-empty buckets
        // in order into LDS (the hash set is no longer needed: every thread passed the barriers of the scans)
        {
            uint32_t run = exT, j = exNE;
            for (int q = 0; q < 64; q += 4) {
                const uint4 c4 = *reinterpret_cast<const uint4 *>(cur + k0 + q);
                const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
                uint32_t o[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    o[t] = run;
                    if (c[t]) tab[j++] = ((u64)(k0 + q + t) << 40) | ((u64)c[t] << 20) | run;
                    run += c[t];
                }
                *reinterpret_cast<uint4 *>(st + k0 + q) = make_uint4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<uint4 *>(cur + k0 + q) = make_uint4(o[0], o[1], o[2], o[3]);
            }
            if (tid == 1023) st[MS_BG] = run; // = n
        }
        if (tid == 0) s_nheads = 0;
        __syncthreads();
        if (wave == 0) { // one wavefront packs: 64 buckets a step, one ballot round per unit that starts among them
            uint32_t ubase = 0, uidx = 0, nh = 0;
            bool open = false, carry_over = false;
            for (uint32_t i0 = 0; i0 < NE; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool valid = i < NE;
                const u64 e = valid ? tab[i] : 0ull;
                const uint32_t stt = (uint32_t)e & 0xFFFFFu, c = (uint32_t)(e >> 20) & 0xFFFFFu;
                const bool over = valid && c > (uint32_t)MS_TILE;
                bool pover = __shfl_up((int)over, 1, 64) != 0;
                if (lane == 0) pover = carry_over;
                carry_over = __shfl((int)over, 63, 64) != 0;
                const uint32_t end = stt + c;
                u64 todo = __ballot(valid);
                while (todo) {
                    const bool brk = valid && (!open || over || pover || end - ubase > (uint32_t)MS_TILE || i - uidx >= 256u);
                    const u64 bm = __ballot(brk) & todo;
                    if (!bm) break;
                    const int f = __ffsll((long long)bm) - 1;
                    ubase = (uint32_t)__shfl((int)stt, f, 64);
                    uidx = i0 + (uint32_t)f;
                    open = true;
                    if (lane == f && nh < 2048u) heads[nh] = make_uint2(stt, (uint32_t)(e >> 40) | ((over ? 2u : 0u) << 17));
                    nh++;
                    todo &= ~((2ull << f) - 1ull);
                }
            }
            if (lane == 0) {
                s_nheads = nh;
                if (nh > 2048u) atomicOr(m.err, ERR_MSD); // (cannot happen: at most ~n / MS_TILE * 2 + NE / 256 units)
            }
        }
        __syncthreads();
        const uint32_t H = min(s_nheads, 2048u);
        for (uint32_t i = tid; i < H; i += 1024) {
            const uint2 hd = heads[i];
            const uint32_t us = hd.x, uk = hd.y & 0x1FFFFu, cls = hd.y >> 17;
            const uint32_t ue = i + 1 < H ? heads[i + 1].x : n;
            const uint32_t uk1 = i + 1 < H ? (heads[i + 1].y & 0x1FFFFu) : MS_BG;
            if (cls < 2u) {
                if (ue - us > (uint32_t)MS_TILE) atomicOr(m.err, ERR_MSD);
                ms_push_unit(m, ms_unit(b, 0u, 0u, uk1 - uk, us, ue, (uint32_t)((size_t)b * MS_BG_ROW + uk)));
            } else {
                ms_push_seg(m, 1u, b, us, ue);
            }
        }
    }
    // the last block to finish tells the host how many blocks keep the 8-pass path
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const uint32_t d = atomicAdd(&m.cnt[MC_PLAN_DONE], 1u);
        if (d == gridDim.x - 1) {
            hsum[0] = atomicAdd(&m.cnt[MC_OLD], 0u);
            hsum[1] = atomicAdd(&m.cnt[MC_NEW], 0u);
            __threadfence_system();
            __hip_atomic_store(hsum + SUMMARY_WORDS - 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- text -> 2-byte buckets in one pass ---------------------------------------------------------------------------
__global__ void __launch_bounds__(MS_THREADS, 4) bigram_scatter(Msd m, uint32_t T, Lst lst)
{
    uint32_t b, tile;
    if (!wg_map(T, lst, b, tile)) return;
    const uint32_t n = m.n[b];
    const uint32_t tile0 = tile * MS_TILE;
    if (tile0 >= n) return;
    const uint32_t valid = min((uint32_t)MS_TILE, n - tile0);
    const uint8_t *s = m.blk + (size_t)b * m.S;
    __shared__ uint32_t txt[MS_TILE / 4 + 8]; // text bytes tile0 .. tile0 + MS_TILE + 31 (cyclic)
    __shared__ uint32_t stage[MS_TILE];
    __shared__ MsCnt cur[2];
    __shared__ uint32_t ucnt[256];
    __shared__ uint32_t ls[MS_NW + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ms_clear(cur[1]);
    {
        const uint32_t p = tile0 + threadIdx.x * 16;
        uint4 q;
        if (p + 16 <= n) {
            q = *reinterpret_cast<const uint4 *>(s + p); // (block base and tile0 are multiples of 16)
        } else {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) {
                uint32_t x = p + j;
                if (x >= n) x %= n;
                w[j >> 2] |= (uint32_t)s[x] << (8 * (j & 3));
            }
            q = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(&txt[threadIdx.x * 4]) = q;
        if (threadIdx.x < 8) {
            uint32_t w = 0;
            for (uint32_t j = 0; j < 4; j++) {
                uint32_t x = tile0 + MS_TILE + threadIdx.x * 4 + j;
                if (x >= n) x %= n;
                w |= (uint32_t)s[x] << (8 * j);
            }
            txt[MS_TILE / 4 + threadIdx.x] = w;
        }
    }
    __syncthreads();
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(txt);
    // item = [bigram : 16 @13][offset in the tile : 13]
    uint32_t v[MS_ITEMS];
    uint32_t actmask = 0;
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++) {
        const uint32_t p = wave * (MS_ITEMS * 64) + k * 64 + lane;
        actmask |= (p < valid ? 1u : 0u) << k;
        v[k] = ((uint32_t)tb[p] << 21) | ((uint32_t)tb[p + 1] << 13) | p; // (branch-free: slots past `valid` are never ranked)
    }
    // (Round 6 tried two cheaper-looking forms, both bit-exact, both slower.  GROUPING the tile by bigram through an open-addressing
    // hash table in LDS instead of sorting it -- one compare-and-swap and one atomic add an element, a scan of the slots; the
    // claims below only need a bigram's elements adjacent --: 975 against 590 us.  And in seg_scatter the one-byte pass with LDS
    // atomics (tile_rank_unordered) instead of the ballot ranking: 93 against 79 us, 59 against 42 on real text.  The atomics
    // of a frequent key queue on one LDS word; the ballot ranking does not care how often a digit occurs.)
    uint32_t pos[MS_ITEMS / 2];
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const int sh = 13 + 8 * pass;
        if (pass == 0) // (by byte 1: no order to keep yet)
            tile_rank_unordered(v, sh, actmask, MS_ITEMS, ucnt, ls, pos);
        else
            tile_rank<8>(v, sh, actmask, MS_ITEMS, cur[1], cur[0], ls, pos, threadIdx.x);
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++)
            if ((actmask >> k) & 1u) stage[(pos[k >> 1] >> (16 * (k & 1))) & 0xFFFFu] = v[k];
        __syncthreads();
        if (pass == 0) {
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) v[k] = stage[wave * (MS_ITEMS * 64) + k * 64 + lane];
        }
    }
    uint32_t *cursor = m.bgcur + (size_t)b * MS_BG;
    u64 *dst = m.bufX + (size_t)b * m.S;
    // A global atomic returns after one to two microseconds: the claims of all 16 rows are issued first, consumed after
    // (a row's claim: one atomic add per run of equal bigrams, by the run's first lane).
    uint32_t its[MS_ITEMS], basev[MS_ITEMS];
    uint32_t hd[MS_ITEMS / 4]; // the lane of my run's head, 8 bits a row
#pragma unroll
    for (int k = 0; k < MS_ITEMS / 4; k++) hd[k] = 0;
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++) {
        const uint32_t p = wave * (MS_ITEMS * 64) + k * 64 + lane;
        const bool act = p < valid;
        its[k] = act ? stage[p] : 0u;
        basev[k] = 0;
        const uint32_t bg = its[k] >> 13;
        const uint32_t pb = (uint32_t)__shfl_up((int)bg, 1, 64);
        const bool head = act && (lane == 0 || bg != pb);
        const u64 hm = __ballot(head), am = __ballot(act);
        if (am == 0ull) continue; // (the valid positions are a prefix of the tile)
        const u64 upto = (2ull << lane) - 1ull; // bits 0..lane (lane 63: all)
        const u64 below = hm & upto, above = hm & ~upto;
        const int myhead = below ? 63 - __clzll((long long)below) : 0;
        const int nexth = above ? __ffsll((long long)above) - 1 : (int)__popcll(am);
        hd[k >> 2] |= (uint32_t)myhead << (8 * (k & 3));
        if (act && head) basev[k] = atomicAdd(cursor + bg, (uint32_t)(nexth - lane));
    }
#pragma unroll
    for (int k = 0; k < MS_ITEMS; k++) {
        const uint32_t p = wave * (MS_ITEMS * 64) + k * 64 + lane;
        const bool act = p < valid;
        const int myhead = (int)((hd[k >> 2] >> (8 * (k & 3))) & 255u);
        const uint32_t d = (uint32_t)__shfl((int)basev[k], myhead, 64) + (uint32_t)(lane - myhead);
        if (act) {
            const uint32_t off = its[k] & 8191u, a = off + 2u;
            const u64 w = ((u64)txt[(a >> 2) + 1] << 32) | txt[a >> 2];
            const u64 x = w >> (8u * (a & 3u)); // byte 2 of the rotation lowest
            const u64 key40 = __builtin_bswap64(x << 24) & 0xFFFFFFFFFFull;
            if (d < n)
                dst[d] = (key40 << 20) | (u64)(tile0 + off);
            else
                atomicOr(m.err, ERR_MSD);
        }
    }
}

// ---- oversized buckets, level by level ------------------------------------------------------------------------------
__device__ __forceinline__ const u64 *ms_level_src(const Msd &m, uint32_t L) { return ((L - 1) & 1u) ? m.bufY : m.bufX; }
__device__ __forceinline__ u64 *ms_level_dst(const Msd &m, uint32_t L) { return (L & 1u) ? m.bufY : m.bufX; }

__global__ void __launch_bounds__(MS_THREADS) seg_count(Msd m, uint32_t L)
{
    const uint32_t nitems = m.cnt[MC_ITEMS + L];
    const uint32_t shift = 20u + 8u * (MS_LEVELS - L); // level 1: byte 2 of the rotation = bits 52..59
    __shared__ uint32_t h[MS_NW][256];
    const int wave = threadIdx.x >> 6;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint32_t item = m.items[(size_t)L * m.B * MS_ITEM_CAP + it];
        const uint4 sg = m.segs[(size_t)L * m.B * MS_SEG_SLOTS + (item & 0xFFFFFu)];
        const uint32_t b = sg.x & 1023u, slot = sg.x >> 10, t = item >> 20;
        const uint32_t e0 = sg.y + t * MS_TILE, cntv = min((uint32_t)MS_TILE, sg.z - e0);
        const u64 *src = ms_level_src(m, L) + (size_t)b * m.S + e0;
        for (int k = threadIdx.x; k < MS_NW * 256; k += MS_THREADS) (&h[0][0])[k] = 0;
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t p = k * MS_THREADS + threadIdx.x;
            if (p < cntv) atomicAdd(&h[wave][(uint32_t)(src[p] >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t c = 0;
#pragma unroll
            for (int w = 0; w < MS_NW; w++) c += h[w][threadIdx.x];
            if (c) atomicAdd(ms_seg_cur_row(m, L, b, slot) + threadIdx.x, c);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) seg_plan(Msd m, uint32_t L, uint32_t *hrec = nullptr, uint32_t seq = 0)
{
    const uint32_t nsegs = m.cnt[MC_SEGS + L];
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t ls[8];
    __shared__ uint2 bk[256];   // (start, count) of every sub-bucket
    __shared__ uint2 heads[256];
    __shared__ uint32_t s_H;
    for (uint32_t g = blockIdx.x; g < nsegs; g += gridDim.x) {
        const uint4 sg = m.segs[(size_t)L * m.B * MS_SEG_SLOTS + g];
        const uint32_t b = sg.x & 1023u, slot = sg.x >> 10, s = sg.y, e = sg.z;
        uint32_t *row = ms_seg_cur_row(m, L, b, slot);
        uint32_t *strow = ms_seg_start_row(m, L, b, slot);
        const uint32_t c = row[tid];
        uint32_t tot;
        const uint32_t ex = block_excl_add(c, ls, &tot);
        if (tot != e - s && tid == 0) atomicOr(m.err, ERR_MSD);
        const uint32_t start = s + ex;
        strow[tid] = start;
        if (tid == 0) strow[256] = e;
        row[tid] = start; // from here on: the claim cursor of the sub-bucket
        bk[tid] = make_uint2(start, c);
        __syncthreads();
        if (tid == 0) { // greedy packing of the (at most 256) sub-buckets, in order
            uint32_t nh = 0, ubase = 0;
            bool open = false, pover = false;
            for (uint32_t d = 0; d < 256; d++) {
                const uint2 q = bk[d];
                if (!q.y) continue;
                const bool over = q.y > (uint32_t)MS_TILE;
                if (!open || over || pover || q.x + q.y - ubase > (uint32_t)MS_TILE) {
                    heads[nh++] = make_uint2(q.x, d | ((over ? 2u : 0u) << 17));
                    ubase = q.x;
                    open = true;
                }
                pover = over;
            }
            s_H = nh;
        }
        __syncthreads();
        const uint32_t H = s_H;
        if (tid < H) {
            const uint2 hd = heads[tid];
            const uint32_t us = hd.x, uk = hd.y & 0x1FFFFu, ucls = hd.y >> 17;
            const uint32_t ue = tid + 1 < H ? heads[tid + 1].x : e;
            const uint32_t uk1 = tid + 1 < H ? (heads[tid + 1].y & 0x1FFFFu) : 256u;
            if (ucls < 2u) {
                if (ue - us > (uint32_t)MS_TILE) atomicOr(m.err, ERR_MSD);
                ms_push_unit(m, ms_unit(b, L & 1u, 0u, uk1 - uk, us, ue, (uint32_t)(strow + uk - m.pool)));
            } else if (L < MS_LEVELS) {
                ms_push_seg(m, L + 1, b, us, ue);
            } else { // bytes 0..6 all equal: ONE group, in tiles
                *ms_spans(m, b) = 1u; // (mid_sort orders whole groups inside one unit's records: not this block's)
                for (uint32_t q = us; q < ue; q += MS_TILE) ms_push_unit(m, ms_unit(b, L & 1u, 1u, 0u, q, min(ue, q + (uint32_t)MS_TILE), us));
            }
        }
        __syncthreads();
    }
    // Level 1 tells the host how many buckets are still oversized: text mostly has none, and the host then leaves the twelve
    // launches of levels 2-5 out (it reads this while seg_scatter of level 1 runs: nothing waits).  Words 4 and 5 of the
    // plan's record; the last workgroup to finish writes them.
    if (hrec && tid == 0) {
        __threadfence();
        if (atomicAdd(&m.cnt[5], 1u) == gridDim.x - 1) {
            hrec[4] = atomicAdd(&m.cnt[MC_SEGS + L + 1], 0u);
            __threadfence_system();
            __hip_atomic_store(hrec + 5, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void __launch_bounds__(MS_THREADS, 4) seg_scatter(Msd m, uint32_t L)
{
    const uint32_t nitems = m.cnt[MC_ITEMS + L];
    const uint32_t shift = 20u + 8u * (MS_LEVELS - L);
    __shared__ u64 stage[MS_TILE];
    __shared__ MsCnt cur[2];
    __shared__ uint32_t ls[MS_NW + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int par = 0;
    ms_clear(cur[0]);
    __syncthreads();
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint32_t item = m.items[(size_t)L * m.B * MS_ITEM_CAP + it];
        const uint4 sg = m.segs[(size_t)L * m.B * MS_SEG_SLOTS + (item & 0xFFFFFu)];
        const uint32_t b = sg.x & 1023u, slot = sg.x >> 10, t = item >> 20;
        const uint32_t e0 = sg.y + t * MS_TILE, cntv = min((uint32_t)MS_TILE, sg.z - e0);
        const u64 *src = ms_level_src(m, L) + (size_t)b * m.S + e0;
        u64 *dst = ms_level_dst(m, L) + (size_t)b * m.S;
        uint32_t *cursor = ms_seg_cur_row(m, L, b, slot);
        u64 v[MS_ITEMS];
        uint32_t actmask = 0;
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t p = wave * (MS_ITEMS * 64) + k * 64 + lane;
            actmask |= (p < cntv ? 1u : 0u) << k;
            v[k] = src[p < cntv ? p : 0u]; // (branch-free; slots past the end are never ranked)
        }
        uint32_t pos[MS_ITEMS / 2];
        tile_rank<8>(v, (int)shift, actmask, MS_ITEMS, cur[par], cur[par ^ 1], ls, pos, threadIdx.x);
        par ^= 1;
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++)
            if ((actmask >> k) & 1u) stage[(pos[k >> 1] >> (16 * (k & 1))) & 0xFFFFu] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t p = wave * (MS_ITEMS * 64) + k * 64 + lane;
            const bool act = p < cntv;
            if (__ballot(act) == 0) continue;
            const u64 x = act ? stage[p] : 0ull;
            const uint32_t d = (uint32_t)(x >> shift) & 255u;
            const uint32_t pd = (uint32_t)__shfl_up((int)d, 1, 64);
            const bool head = act && (lane == 0 || d != pd);
            const uint32_t at = row_claim(act, head, cursor + d, lane);
            if (act) {
                if (at >= sg.y && at < sg.z)
                    dst[at] = x;
                else
                    atomicOr(m.err, ERR_MSD);
            }
        }
        __syncthreads(); // stage and cur are reused by the next tile
    }
}

// ---- one unit = whole buckets, at most a tile: sorted, ranked and routed inside one workgroup ----------------------------
// Sort: elements in registers, wave w holding rows of 64 consecutive slots ("striped": what the ballot ranking wants);
// LDS element = [bucket index in the unit : 8 @53][bytes 2..6 : 40 @13][slot the element was loaded at : 13] -- the suffix
// is NOT carried through the passes: the low halves of the unit's elements are read once more after the last pass
// (coalesced, the lines are in the L2) into a table by load slot, from which every sorted element fetches its suffix.
// After the sort everything stays striped (slot w = k * 512 + thread; row k * 8 + wave = 64 consecutive slots of the
// sorted order): group heads by ballots (one 64-bit mask per row), a group's extent from the masks of its row and the
// rows next to it, and then
//   * THE FIRST DOUBLING STEP OF THE SMALL GROUPS (m.fuse): a group of 2..64 rotations that share their first 7 bytes
//     is ordered by bytes 7..14 -- one 8-byte load from the text per member (what rank[i + 7] at depth 7 would say, and
//     one byte more), all pairs inside the group from keys in LDS, exactly as tail_round ranks a group from gathered
//     ranks.  The members leave with their rank at depth 15: the rank words of this step are the ones the initial
//     binning writes anyway, the small-group list holds what is STILL unresolved (about a third of what depth 7 left),
//     and the block's small groups sit round 0 out (round_begin) -- there is no tail_round over 43 M suffixes;
//   * (rank word, suffix) pairs binned by 4096-suffix window of the rank array (rank_apply turns them into whole lines),
//     small groups to the small-group list (a group's members adjacent), large groups to the big list (in order).
// Units are whole buckets, so no group crosses a unit: no carries, no look-back.
constexpr int MS_SLOTS = MS_TILE + MS_TILE / 16;
__device__ __forceinline__ uint32_t ms_slot(uint32_t e) { return e + (e >> 4); } // (+1 per 16: blocked 8-byte accesses stay conflict free)
constexpr u64 MS_REC_BIG = 1ull << 63; // in the staged list records: member of a large group

// bytes 7..14 of rotation i (big-endian: the smaller key is the smaller rotation among rotations that share 7 bytes)
__device__ __forceinline__ u64 ms_key8(const uint8_t *txt, uint32_t i, uint32_t n)
{
    if (n <= 7u) return (u64)(n - 1u - i); // equal for 7 >= n bytes: identical rotations, larger index first
    uint32_t p = i + 7u;
    if (p >= n) p -= n;
    if (p + 8u <= n) {
        u64 v;
        __builtin_memcpy(&v, txt + p, 8); // (gfx950 global loads need no alignment)
        return __builtin_bswap64(v);
    }
    u64 v = 0;
    for (int q = 0; q < 8; q++) { // cyclic wrap (also more than once: n may be as small as 8)
        v = (v << 8) | txt[p];
        if (++p == n) p = 0;
    }
    return v;
}

// (a fresh copy of a lane-dependent value the optimizer cannot relate to the others: what is derived from it -- shuffle
// addresses, lane compares of a scan -- is computed where it is used instead of once per unit and held in registers)
__device__ __forceinline__ int ms_opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__global__ void __launch_bounds__(MS_THREADS, 4) chunk_finish(Msd m)
{
    __shared__ u64 stage[MS_SLOTS];
    __shared__ MsCnt cur[2];         // counters of the passes; afterwards: the bins of the rank binning
    __shared__ u64 HM[128];          // bucket heads by load position, later group heads by sorted position (bit per slot)
    __shared__ uint32_t rowpre[128]; // bucket heads before a row; later 1 + position of the last group head before a row
    __shared__ uint32_t wc[128];     // per row of the sorted order: surviving small-group records | large-group records << 16
    __shared__ uint32_t ls[MS_NW + 2], lsv[MS_NW];
    __shared__ u64 s_wo[MS_NW], s_wa[MS_NW]; // per wavefront: OR / AND of its elements
    __shared__ uint32_t s_unit, s_offS, s_offB, s_totB;
    uint32_t *const bh = reinterpret_cast<uint32_t *>(&cur[0][0][0]); // 4 x 256 words: counts, local starts, cursors, offsets
    uint32_t *const bl = bh + 256, *const bcur = bh + 512, *const bgo = bh + 768;
    // (timing experiments, BZH_MSD_DBG & 16: cycles per phase, summed over all units, to cnt[32 ..]; the last reading lives in
    // LDS and the sums go straight to memory: no register of the product path is held for it)
    __shared__ long long s_tlast;
#define MS_T(k)                                                                    \
    if ((m.dbg & 16u) && tid == 0) {                                               \
        const long long t_now = clock64();                                         \
        atomicAdd(&m.cnt[32 + (k)], (uint32_t)((t_now - s_tlast) >> 4));           \
        s_tlast = t_now;                                                           \
    }
    // ---- which unit next.  Workgroups are dealt round-robin over the 8 XCDs (as wg_map assumes; used for speed only), and a
    // workgroup works through the blocks b = xcd, xcd + 8, ... in turn, a ticket per unit of the block: the block's text
    // (the key gathers of the first doubling step: 0.9 MB), its rank windows (partial lines from many units) and its list
    // tails then live in ONE L2 instead of in all eight.  When its XCD's blocks are used up a workgroup takes units of
    // whichever block has the most left (one look at all counters by wave 0, then a ticket).  The ticket of the NEXT unit,
    // the room claimed in the two lists and in the rank windows are requested as soon as their arguments exist and consumed
    // as late as possible: a global atomic returns after one to two microseconds.
    __shared__ uint32_t s_blk, s_next;
    __shared__ uint32_t s_ng, s_gbase, glist[MS_TILE / TAIL_G]; // ranks of the unit's large groups (they get numbers: one atomic add a unit)
    const uint32_t NOBLK = 0xFFFFFFFFu;
    uint32_t cur_blk = blockIdx.x & 7u; // (uniform; >= m.B: nothing of my own)
    bool own = true;                    // still inside my XCD's sequence of blocks
    uint32_t cur_cnt = 0; // (cur_cnt: units of cur_blk -- final, the plans ran in earlier launches)
    if (cur_blk >= m.B) {
        own = false;
        cur_blk = NOBLK;
    } else {
        cur_cnt = min(*ms_unit_count(m, cur_blk), MS_UNIT_CAP);
    }
    if (threadIdx.x == 0 && cur_blk != NOBLK) s_next = atomicAdd(ms_unit_ticket(m, cur_blk), 1u);
    uint32_t tid = threadIdx.x;
    for (;;) {
        // The thread's index is made opaque once per unit: otherwise every address and mask that depends on it only (a few
        // dozen values: slots, rows, lane masks of all 16 steps) is hoisted out of this loop and kept alive across it -- in
        // scratch memory (the compiler's resource report showed 80 spilled registers, all of this kind).
        asm volatile("" : "+v"(tid)); // (one register carried round the loop: threadIdx.x itself does not live on beside it)
        const int lane = (int)(tid & 63u), wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6)); // (the wavefront's number is uniform: scalar row arithmetic)
        if ((m.dbg & 16u) && tid == 0) s_tlast = clock64();
        if (wave == 0) {
            uint32_t bq = cur_blk, t = s_next; // (the ticket requested a unit ago, parked in LDS once it had arrived)
            for (uint32_t tries = 0;; tries++) {
                if (bq != NOBLK && t < cur_cnt) break;
                if (own && bq != NOBLK && bq + 8u < m.B) {
                    bq += 8u;
                } else { // my XCD's blocks are used up: the block with the most units left, anywhere
                    own = false;
                    uint32_t best = 0, bb = NOBLK;
                    for (uint32_t b0 = 0; b0 < m.B; b0 += 64) {
                        const uint32_t bx = b0 + (uint32_t)lane;
                        uint32_t left = 0;
                        if (bx < m.B) {
                            const uint32_t c = min(*ms_unit_count(m, bx), MS_UNIT_CAP);
                            const uint32_t k = __hip_atomic_load(ms_unit_ticket(m, bx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            left = c > k ? c - k : 0u;
                        }
                        const uint32_t mx = wave_all_max(left);
                        if (mx > best) {
                            best = mx;
                            const u64 who = __ballot(left == mx);
                            bb = b0 + (uint32_t)__ffsll((long long)who) - 1u;
                        }
                    }
                    bq = bb;
                    if (bq == NOBLK || tries > 4096u) { // nothing left anywhere (or a logic error: never spin for ever)
                        bq = NOBLK;
                        break;
                    }
                }
                uint32_t tt = 0;
                if (lane == 0) tt = atomicAdd(ms_unit_ticket(m, bq), 1u);
                cur_cnt = min(*ms_unit_count(m, bq), MS_UNIT_CAP); // (beside the ticket's round trip)
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tt);
            }
            cur_blk = bq; // (wave 0's copy; the others take it from LDS)
            if (lane == 0) {
                s_blk = bq;
                s_unit = t;
            }
        }
        __syncthreads();
        const uint32_t ub = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_blk), u = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_unit); // (uniform: scalar registers)
        if (ub == NOBLK) break;
        cur_blk = ub;
        uint32_t pend_ticket = 0;
        if (tid == 0) pend_ticket = atomicAdd(ms_unit_ticket(m, ub), 1u);
        const uint4 ud = m.units[(size_t)ub * MS_UNIT_CAP + u];
        const uint32_t b = ud.x & 1023u, buf = (ud.x >> 10) & 1u, uniform = (ud.x >> 11) & 1u, nb = ud.x >> 12;
        const uint32_t s = ud.y, e = ud.z, tbl = ud.w, len = e - s;
        const uint32_t n = m.n[b];
        if (len == 0 || len > (uint32_t)MS_TILE) {
            if (tid == 0) {
                atomicOr(m.err, ERR_MSD);
                s_next = pend_ticket;
            }
            __syncthreads();
            continue;
        }
        const u64 *src = (buf ? m.bufY : m.bufX) + (size_t)b * m.S + s;
        const int R = (int)((len + 511u) / 512u); // rows of 64 per wave
        const uint32_t Lw = (uint32_t)R * 64u;    // slots per wave
        // ---- load; bucket index of every slot from the bucket starts
        u64 x[MS_ITEMS];
        uint32_t actmask = 0;
        if (tid < 128) HM[tid] = 0ull;
        if (tid == 0) s_ng = 0;
        ms_clear(cur[0], tid);
        __syncthreads();
        const bool multi = !uniform && nb > 1u;
        if (multi) {
            const uint32_t *tb = m.pool + tbl;
            for (uint32_t k = tid; k < nb; k += MS_THREADS) {
                const uint32_t st0 = tb[k], st1 = tb[k + 1];
                if (st1 > st0) {
                    const uint32_t p = st0 - s;
                    if (p < len)
                        atomicOr((unsigned long long *)&HM[p >> 6], 1ull << (p & 63u));
                    else
                        atomicOr(m.err, ERR_MSD);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t p = wave * Lw + k * 64 + lane; // (at most 8191)
            const bool act = k < R && p < len;
            actmask |= (act ? 1u : 0u) << k;
            const u64 w = src[act ? p : 0u]; // (branch-free: slots past the end are never ranked, stored or counted)
            x[k] = ((w >> 20) << 13) | p;
        }
        __syncthreads();
        uint32_t nbk = 1; // non-empty buckets of the unit
        if (multi) {
            if (wave == 0) { // bucket heads before every row of 64 slots: one wavefront, two rows a lane
                const uint32_t p0 = (uint32_t)__popcll(HM[2 * lane]), p1 = (uint32_t)__popcll(HM[2 * lane + 1]);
                const uint32_t inc = wave_incl_add(p0 + p1, ms_opaque(lane));
                rowpre[2 * lane] = inc - p0 - p1;
                rowpre[2 * lane + 1] = inc - p1;
                if (lane == 63) ls[0] = inc;
            }
            __syncthreads();
            nbk = ls[0];
            if (nbk > 256u && tid == 0) atomicOr(m.err, ERR_MSD);
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                if ((actmask >> k) & 1u) {
                    const uint32_t p = wave * Lw + k * 64 + lane, row = p >> 6;
                    const uint32_t bi = rowpre[row] + (uint32_t)__popcll(HM[row] & ((2ull << (p & 63u)) - 1ull)) - 1u;
                    x[k] |= (u64)(bi & 255u) << 53;
                }
            }
        }
        // ---- which digits vary inside the unit?  (a level-L unit shares bytes 2 .. 1+L; a single bucket has one index)
        {
            u64 o = 0ull, a = ~0ull;
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                if ((actmask >> k) & 1u) {
                    o |= x[k];
                    a &= x[k];
                }
            }
            o = ((u64)wave_all_or((uint32_t)(o >> 32)) << 32) | wave_all_or((uint32_t)o);
            a = ((u64)wave_all_and((uint32_t)(a >> 32)) << 32) | wave_all_and((uint32_t)a);
            if (lane == 0) {
                s_wo[wave] = o;
                s_wa[wave] = a;
            }
        }
        if (tid == 0) s_next = pend_ticket; // (requested before the unit's elements were: it is there, and its register is free for the sort)
        __syncthreads();
        MS_T(0);
        u64 vary = 0ull;
        if (!uniform) {
            u64 o = 0ull, a = ~0ull;
#pragma unroll
            for (int w = 0; w < MS_NW; w++) {
                o |= s_wo[w];
                a &= s_wa[w];
            }
            vary = o & ~a;
        }
        {
            uint32_t pos[MS_ITEMS / 2];
            int par = 0;
            bool staged = false;
#pragma unroll 1
            for (int pass = 0; pass < 6; pass++) {
                const int sh = pass < 5 ? 13 + 8 * pass : 53;
                if (((vary >> sh) & 255ull) == 0ull) continue; // (the same for every thread)
                tile_rank<8>(x, sh, actmask, R, cur[par], cur[par ^ 1], ls, pos, tid);
                par ^= 1;
                MS_T(1);
#pragma unroll
                for (int k = 0; k < MS_ITEMS; k++)
                    if ((actmask >> k) & 1u) stage[ms_slot((pos[k >> 1] >> (16 * (k & 1))) & 0xFFFFu)] = x[k];
                __syncthreads();
                MS_T(2);
                staged = true;
                bool again = false; // is there another pass?  (then the striped registers are refilled)
                for (int q = pass + 1; q < 6; q++) again |= ((vary >> (q < 5 ? 13 + 8 * q : 53)) & 255ull) != 0ull;
                if (again) {
#pragma unroll
                    for (int k = 0; k < MS_ITEMS; k++) x[k] = stage[ms_slot((wave * Lw + k * 64 + lane) & 8191u)];
                }
            }
            if (!staged) { // nothing to sort (one key): the next phase still reads the elements from the stage
#pragma unroll
                for (int k = 0; k < MS_ITEMS; k++)
                    if ((actmask >> k) & 1u) stage[ms_slot(wave * Lw + k * 64 + lane)] = x[k];
                __syncthreads();
            }
        }
        MS_T(3);
        // ---- sorted order, striped: slot w = k * 512 + tid; group heads of every row of 64 slots by ballot
        // (the suffixes follow their elements through a table by load slot: the second read of the unit's low halves is
        // issued first and lands while the heads are found)
        uint32_t sfl[MS_ITEMS];
        const uint32_t pb2 = (uint32_t)ms_opaque((int)(wave * Lw + lane)); // (a fresh base: the 16 load slots are not kept from the load on)
        {
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                const uint32_t p = pb2 + k * 64;
                sfl[k] = s32[((actmask >> k) & 1u) ? 2u * p : 0u];
            }
        }
        uint32_t sl[MS_ITEMS / 2]; // the slots my elements were loaded at, 16 bits each; later: where their list records go
#pragma unroll
        for (int k = 0; k < MS_ITEMS / 2; k++) sl[k] = 0;
        const uint32_t tq1 = (uint32_t)ms_opaque((int)tid); // (its own copy of the index: the slot addresses of this loop are not kept from the loops before)
        #pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t w = k * MS_THREADS + tq1;
            const bool act = w < len;
            u64 key = ~0ull;
            if (act) {
                const u64 y = stage[ms_slot(w)];
                key = y >> 13;
                sl[k >> 1] |= ((uint32_t)y & 8191u) << (16 * (k & 1));
            }
            u64 pk = ((u64)(uint32_t)__shfl_up((int)(uint32_t)(key >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)key, 1, 64);
            if (lane == 0) pk = (act && w > 0u) ? stage[ms_slot(w - 1u)] >> 13 : ~0ull;
            const bool head = act && (w == 0u || (!uniform && key != pk));
            const u64 hm = __ballot(head);
            if (lane == 0) HM[k * MS_NW + wave] = hm;
        }
        __syncthreads(); // every thread holds its elements: the stage is free; the head masks are complete
        // table of the load slots' suffixes over the stage; wave 0 on the way: last group head before every row, groups of the unit
        {
            uint32_t *st32 = reinterpret_cast<uint32_t *>(stage);
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                const uint32_t p = pb2 + k * 64;
                if ((actmask >> k) & 1u) st32[p] = sfl[k] & (uint32_t)SUF_MASK;
            }
        }
        if (wave == 0) {
            const u64 h0 = HM[2 * lane], h1 = HM[2 * lane + 1];
            const int v0 = h0 ? (2 * lane) * 64 + 64 - __clzll((long long)h0) : 0;
            const int v1 = h1 ? (2 * lane + 1) * 64 + 64 - __clzll((long long)h1) : 0;
            const int inc = wave_incl_max(max(v0, v1), ms_opaque(lane));
            int ex = __shfl_up(inc, 1, 64);
            if (lane == 0) ex = 0;
            rowpre[2 * lane] = (uint32_t)ex;
            rowpre[2 * lane + 1] = (uint32_t)max(ex, v0);
            // (the number of groups decides the first mode of blocks on the 8 passes only: nobody counts them here)
        }
        if (tid < 256) bh[tid] = 0; // (the pass counters are free: the last pass ended behind barriers)
        __syncthreads();
        MS_T(4);
        // ---- suffix, group extent and class of every element; rank windows counted
        const bool fuse = m.fuse != 0u && !uniform;
        const uint8_t *txt = m.blk + (size_t)b * m.S;
        uint32_t sf[MS_ITEMS];
        uint32_t gi[MS_ITEMS]; // [class : 2 @30][members - 1 : 6 @24 (small groups)][first slot of the group : 13]
        {
            const uint32_t *st32 = reinterpret_cast<const uint32_t *>(stage);
            const uint32_t tq2 = (uint32_t)ms_opaque((int)tid);
            #pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                const uint32_t w = k * MS_THREADS + tq2;
                sf[k] = 0;
                gi[k] = 0;
                if (w < len) {
                    sf[k] = st32[(sl[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
                    const uint32_t row = k * MS_NW + wave;
                    const u64 own = HM[row];
                    const u64 upto = (2ull << lane) - 1ull; // bits 0..lane (lane 63: all)
                    const u64 below = own & upto, above = own & ~upto;
                    const uint32_t g = below ? row * 64u + 63u - (uint32_t)__clzll((long long)below) : rowpre[row] - 1u;
                    uint32_t ge;
                    if (above) {
                        ge = row * 64u + (uint32_t)__ffsll((long long)above) - 1u;
                    } else { // (exact if it is a head or the unit's end; otherwise the group has more than 64 members anyway)
                        const u64 nm = row + 1u < 128u ? HM[row + 1u] : 0ull;
                        ge = nm ? (row + 1u) * 64u + (uint32_t)__ffsll((long long)nm) - 1u : min(len, (row + 2u) * 64u);
                    }
                    const uint32_t size = ge - g;
                    uint32_t c = size == 1u ? CLS_SINGLE : (size <= (uint32_t)TAIL_G ? CLS_SMALL : CLS_BIG);
                    if (uniform) c = CLS_BIG; // one tile of a group that spans several units
                    gi[k] = (c << 30) | (((size - 1u) & 63u) << 24) | g;
                    if (c == CLS_BIG && g == w) glist[atomicAdd(&s_ng, 1u)] = uniform ? tbl : s + g; // (at most one large group per TAIL_G + 1 slots)
                    atomicAdd(&bh[sf[k] >> 12], 1u);
                }
            }
        }
        __syncthreads(); // the table has been read; the bin counts are complete
        uint32_t pend_g = 0; // numbers for the unit's large groups: requested now, used at the end
        if (tid == 0 && s_ng) pend_g = atomicAdd(&m.gout.gcount[b], s_ng);
        MS_T(5);
        // keys of the small groups' members: bytes 7..14 of their rotations, one 8-byte load each -- issued now, in LDS
        // behind the scan of the rank windows (whose barriers they do not need)
        constexpr int KH = 12; // keys in flight across the scan (the rest follows behind it: registers)
        u64 kk[KH];
        const uint32_t tq3 = (uint32_t)ms_opaque((int)tid);
        #pragma unroll
        for (int k = 0; k < KH; k++) {
            const uint32_t w = k * MS_THREADS + tq3;
            kk[k] = 0ull;
            if (fuse && w < len && (gi[k] >> 30) == CLS_SMALL) kk[k] = BZH_DBG(m.dbg & 32u) ? (u64)sf[k] * 0x9E3779B97F4A7C15ull : ms_key8(txt, sf[k], n);
        }
        // rank binning: local starts of the bins -- one wavefront, four bins a lane, while the keys are on their way (no
        // workgroup scan: three barriers less per unit); their room in the block's windows is claimed behind the next
        // barrier and consumed at the end
        if (wave == 0) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(&bh[4 * lane]);
            const uint32_t t4 = c4.x + c4.y + c4.z + c4.w;
            const uint32_t ex = wave_incl_add(t4, ms_opaque(lane)) - t4;
            const uint4 e4 = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);
            *reinterpret_cast<uint4 *>(&bl[4 * lane]) = e4;
            *reinterpret_cast<uint4 *>(&bcur[4 * lane]) = e4;
        }
        uint32_t pendG = 0; // (tid < 256: room of my bin in the block's window; the bin's count is re-read from bh where needed)
        if (fuse) {
#pragma unroll
            for (int k = KH; k < MS_ITEMS; k++) {
                const uint32_t w = k * MS_THREADS + tid;
                if (w < len && (gi[k] >> 30) == CLS_SMALL) stage[w] = BZH_DBG(m.dbg & 32u) ? (u64)sf[k] * 0x9E3779B97F4A7C15ull : ms_key8(txt, sf[k], n);
            }
            const uint32_t tq4 = (uint32_t)ms_opaque((int)tid);
            #pragma unroll
            for (int k = 0; k < KH; k++) {
                const uint32_t w = k * MS_THREADS + tq4;
                if (w < len && (gi[k] >> 30) == CLS_SMALL) stage[w] = kk[k];
            }
            __syncthreads();
            if (tid < 256 && bh[tid]) pendG = atomicAdd(&m.bincur[(size_t)b * 256 + tid], bh[tid]);
        }
        if (tid == 0) s_gbase = pend_g; // (it has arrived behind the key gathers: parked, its register is free for the ranking)
        MS_T(6);
        // ---- the first doubling step of the small groups: every member counts the members that sort before it
        // (gi[k] becomes [class : 2 @30][first position of the element's group in the block's order : 20])
        {
            uint32_t nsv = 0, nin = 0; // my records for the two lists: small groups | large groups << 16; my members of small groups
            const uint32_t tq5 = (uint32_t)ms_opaque((int)tid);
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                const uint32_t w = k * MS_THREADS + tq5;
                if (w < len) {
                    uint32_t c = gi[k] >> 30;
                    const uint32_t g = gi[k] & 8191u;
                    uint32_t hp = s + g, dest = w;
                    if (fuse && c == CLS_SMALL && !BZH_DBG(m.dbg & 64u)) {
                        nin++; // (a member of a small group ENTERS the first doubling step: counted in A, bwt.hip round_begin)
                        const uint32_t ge = g + ((gi[k] >> 24) & 63u) + 1u;
                        const u64 my = stage[w];
                        uint32_t less = 0, eq = 0, eqb = 0;
#pragma unroll 4
                        for (uint32_t f = g; f < ge; f++) { // bounds known up front: the LDS reads pipeline
                            const u64 kf = stage[f];
                            less += kf < my;
                            eq += kf == my;
                            eqb += kf == my && f < w;
                        }
                        hp += less; // a group of <= 64 members: the members with equal keys share the new head
                        dest = g + less + eqb; // the u-th smallest member takes the slot of the u-th member
                        if (eq == 1u) c = CLS_SINGLE;
                    }
                    if (uniform) hp = tbl;
                    gi[k] = (c << 30) | hp;
                    nsv += (c == CLS_SMALL ? 1u : 0u) + (c == CLS_BIG ? 0x10000u : 0u);
                    sl[k >> 1] = (sl[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | (dest << (16 * (k & 1)));
                }
            }
            nsv = wave_all_add(nsv);
            nin = wave_all_add(nin);
            if (lane == 0) {
                lsv[wave] = nsv;
                if (nin) atomicAdd(&m.cnt[6], nin);
            }
        }
        __syncthreads(); // the keys have been read: the stage is free
        if (!fuse && tid < 256 && bh[tid]) pendG = atomicAdd(&m.bincur[(size_t)b * 256 + tid], bh[tid]);
        MS_T(7);
        uint32_t pendS = 0, pendB = 0; // (pendB: first record | number of the run << 20)
        if (tid == 0) { // (a unit holds at most 8192 records: 16 bits each)
            uint32_t t2 = 0;
#pragma unroll
            for (int w = 0; w < MS_NW; w++) t2 += lsv[w];
            const uint32_t totS = t2 & 0xFFFFu, totB = t2 >> 16;
            pendS = totS ? atomicAdd(&m.c_small[b], totS) : 0u;
            pendB = totB ? atomicAdd(&m.runq[b], (1u << 20) | totB) : 0u; // (mid_plan leaves the list's length where round_begin looks for it)
            s_totB = totB;
        }
        // ---- (rank word, suffix) pairs in bin order in LDS, then out as runs
        if (tid < 256) {
            const uint32_t w0 = tid * 4096u, wcap = w0 < n ? min(4096u, n - w0) : 0u, binc = bh[tid];
            if (binc && pendG + binc > wcap) atomicOr(m.err, ERR_MSD);
            bgo[tid] = min(n, w0) + pendG;
        }
        const uint32_t tq6 = (uint32_t)ms_opaque((int)tid);
        #pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t w = k * MS_THREADS + tq6;
            if (w < len) {
                const uint32_t c = gi[k] >> 30, head = gi[k] & 0xFFFFFu;
                uint32_t word = head;
                if (c == CLS_SINGLE) {
                    // A rotation that is alone in its group has its final place: its byte of the last column leaves now (the
                    // text is in this XCD's L2, neighbours in the order are neighbours in the column), and the rank word says
                    // so -- tag 31, which no round writes with less = 0 -- for bwt_emit, which then scatters the rest only.
                    word = head | RANK_RESOLVED | RANK_EMITTED;
                    m.bwt[(size_t)b * m.S + head] = txt[sf[k] ? sf[k] - 1u : n - 1u];
                }
                stage[atomicAdd(&bcur[sf[k] >> 12], 1u)] = ((u64)word << 32) | sf[k];
            }
        }
        __syncthreads();
        {
            u64 *dst = m.binned + (size_t)b * m.S;
            for (uint32_t q = tid; q < len; q += MS_THREADS) {
                const u64 w = stage[q];
                const uint32_t d = ((uint32_t)w & (uint32_t)SUF_MASK) >> 12;
                dst[bgo[d] + (q - bl[d])] = w;
            }
        }
        __syncthreads();
        MS_T(8);
        // ---- list records through LDS, each at its place in the new order: a small group's members adjacent, equal keys together
        const uint32_t tq7 = (uint32_t)ms_opaque((int)tid);
        #pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t w = k * MS_THREADS + tq7;
            if (w < len) {
                const uint32_t c = gi[k] >> 30;
                const u64 rec = ((u64)(gi[k] & 0xFFFFFu) << 40) | sf[k];
                stage[(sl[k >> 1] >> (16 * (k & 1))) & 0xFFFFu] = c == CLS_SMALL ? rec : (c == CLS_BIG ? (rec | MS_REC_BIG) : LIST_INVALID);
            }
        }
        if (tid == 0) {
            s_offS = pendS;
            s_offB = pendB; // (first record | number of the run << 20)
        }
        __syncthreads();
        // where the unit's large-group records stand in the block's big list: mid_sort (bwt.hip, round 0) orders the list run
        // by run -- every group is whole inside its unit's run.  (Everything from LDS, by a thread of its own: no register of
        // the unit's long phases is held for it.)
        if (tid == MS_THREADS - 1 && s_totB) m.runs[(size_t)s_blk * MS_UNIT_CAP + (s_offB >> 20)] = make_uint2(s_offB & 0xFFFFFu, s_totB);
        if (tid < s_ng) gid_assign(m.gout, b, s_gbase + tid, glist[tid]);
        {
            u64 o[MS_ITEMS];
            const uint32_t tq8 = (uint32_t)ms_opaque((int)tid);
            #pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                const uint32_t w = k * MS_THREADS + tq8;
                o[k] = w < len ? stage[w] : LIST_INVALID;
                const u64 mS = __ballot(o[k] != LIST_INVALID && !(o[k] & MS_REC_BIG)), mB = __ballot(o[k] != LIST_INVALID && (o[k] & MS_REC_BIG));
                if (lane == 0) wc[k * MS_NW + wave] = (uint32_t)__popcll(mS) | ((uint32_t)__popcll(mB) << 16);
            }
            __syncthreads(); // the row counts are there (and the records in registers)
            // exclusive scan of the row counts in slot order (every wavefront for itself: 128 values, two a lane)
            const uint32_t c0 = wc[2 * lane], c1 = wc[2 * lane + 1];
            const uint32_t inc = wave_incl_add(c0 + c1, ms_opaque(lane));
            const uint32_t ex0 = inc - c0 - c1, ex1 = inc - c1;
            u64 *ts = m.tail + (size_t)b * m.S + s_offS;
            u64 *bs = m.big + (size_t)b * m.S + (s_offB & 0xFFFFFu);
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                const int row = k * MS_NW + wave;
                const uint32_t a0 = (uint32_t)__shfl((int)ex0, row >> 1, 64), a1 = (uint32_t)__shfl((int)ex1, row >> 1, 64);
                const uint32_t off = (row & 1) ? a1 : a0;
                const bool isS = o[k] != LIST_INVALID && !(o[k] & MS_REC_BIG), isB = o[k] != LIST_INVALID && (o[k] & MS_REC_BIG);
                const u64 mS = __ballot(isS), mB = __ballot(isB);
                if (isS) ts[(off & 0xFFFFu) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mS >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mS, 0u))] = o[k];
                if (isB) bs[(off >> 16) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mB, 0u))] = o[k] & ~MS_REC_BIG;
            }
        }
        MS_T(9);
        __syncthreads(); // the stage, the bins and s_unit are reused by the next unit
    }
#undef MS_T
}

// ---- round 0 of the large groups, unit by unit ---------------------------------------------------------------------------
// What chunk_finish leaves in a block's big list are runs of records, one run per unit, and a group is whole inside its
// unit's run (units are whole buckets; a block with a group that spans several units -- ms_spans -- is not touched here and
// takes the global passes).  Ordering a run by (group, key2 = rank[i + h]) is therefore all round 0's big-list path has to
// do for such a block before refine_one: this kernel does it in LDS -- one workgroup per tile of runs (mid_plan), the records gathered and
// keyed once, four counting passes of 7 bits (20 bits of key2, then the group's index inside the run: a large group has
// more than TAIL_G members, so a run holds fewer than 128), written back in place in the form refine_one expects -- instead of
// active_gen + four (five) global radix passes over every record (8 + 4 x 16 bytes a record through HBM, and the round's
// critical path: round 0 has no small-group kernel to hide behind once chunk_finish takes their first step).
// Tiles are taken block by block on the block's XCD, as chunk_finish takes its units (the rank gathers stay inside one L2).
// One workgroup per bucket-first block, before mid_sort: the runs of the block's big list (in list order, by construction of
// their claims) packed greedily into tiles of at most MS_TILE records -- a tile is a stretch of the list made of whole runs,
// i.e. of whole groups, and mid_sort orders a tile at a time (a unit's run alone averages 1,200 records on text: seven times
// the workgroup rounds, each with its chain of dependent loads).  `first`: behind chunk_finish -- it also leaves the list's
// length where round_begin looks for it (chunk_finish's one atomic per unit is the run claim; refine_one makes both).
__global__ void __launch_bounds__(256) mid_plan(Msd m, uint32_t first)
{
    const uint32_t b = blockIdx.x;
    if (b >= m.B || m.np[b] == 0u) return; // (a block on the 8 passes: refine_one<init> writes its lists)
    __shared__ uint2 R[MS_UNIT_CAP];
    const uint32_t q = m.runq[b], recs = q & 0xFFFFFu;
    uint32_t runs = q >> 20;
    if (recs && !runs) runs = MS_UNIT_CAP; // (the 4096th run wrapped the field)
    const uint2 *rt = m.runs + (size_t)b * MS_UNIT_CAP;
    for (uint32_t k = threadIdx.x; k < runs; k += 256) R[k] = rt[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        if (first) m.c_big[b] = recs;
        m.mticket[b] = 0u;
        uint2 *tt = m.tiles + (size_t)b * MS_UNIT_CAP;
        uint32_t nt = 0, ts = 0, tl = 0;
        bool bad = false;
        for (uint32_t k = 0; k < runs; k++) {
            const uint2 r = R[k];
            bad |= r.x != ts + tl || r.y == 0u || r.y > (uint32_t)MS_TILE; // (runs follow one another in the list)
            if (tl && tl + r.y > m.midcap) {
                tt[nt++] = make_uint2(ts, tl);
                ts = r.x;
                tl = 0;
            }
            tl += r.y;
        }
        if (tl) tt[nt++] = make_uint2(ts, tl);
        if (bad || ts + tl != recs) atomicOr(m.err, ERR_MSD);
        m.ntiles[b] = nt;
    }
}

struct MidArgs {
    u64 *list;             // [B][S] the big lists, ordered in place
    const uint32_t *rank;  // [B][S]
    const uint32_t *hb;    // [B] depth of the block's round
    const uint32_t *len;   // [B] records in the block's big list (gateA)
    uint32_t tag;          // this round's id in the rank words
    // of the initial sort's state (Msd): the tiles of the big lists and their number per block (mid_plan), this kernel's
    // tickets, "a group spans several units", "took the bucket-first sort", block sizes
    const uint2 *tiles;
    const uint32_t *ucount, *spans, *np, *n;
    uint32_t *ticket, *err;
    uint32_t S, B;
    uint32_t dbg;          // timing experiments only (BZH_MID_DBG: 1 = no counting passes, 2 = no rank gather, 4 = no stores; wrong results)
};

__global__ void __launch_bounds__(MS_THREADS, 4) mid_sort(MidArgs a)
{
    __shared__ u64 stage[MS_SLOTS];
    __shared__ MsCnt cur[2];
    __shared__ u64 HM[128];          // group heads by load position (bit per slot)
    __shared__ uint32_t rowpre[128]; // group heads before a row
    __shared__ uint32_t G[128];      // group index inside the tile -> the group's rank (the record leaves with it in front)
    __shared__ uint32_t ls[MS_NW + 2];
    __shared__ u64 s_wo[MS_NW], s_wa[MS_NW];
    __shared__ uint32_t s_blk, s_unit, s_next;
    const uint32_t NOBLK = 0xFFFFFFFFu;
    const uint32_t *spans = a.spans;
    uint32_t cur_blk = blockIdx.x & 7u;
    bool own = true;
    uint32_t cur_cnt = 0;
    if (cur_blk >= a.B) {
        own = false;
        cur_blk = NOBLK;
    } else {
        cur_cnt = ms_block_is_mid(a.np, spans, cur_blk) ? min(a.ucount[cur_blk], MS_UNIT_CAP) : 0u;
    }
    if (threadIdx.x == 0 && cur_blk != NOBLK) s_next = cur_cnt ? atomicAdd(a.ticket + cur_blk, 1u) : 0u;
    uint32_t tid = threadIdx.x;
    for (;;) {
        asm volatile("" : "+v"(tid)); // (opaque once per unit: what depends on it alone is not kept across the loop -- see chunk_finish)
        const int lane = (int)(tid & 63u), wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
        if (wave == 0) { // which unit next (chunk_finish's scheme: my XCD's blocks in turn, then whichever block has the most left)
            uint32_t bq = cur_blk, t = s_next;
            for (uint32_t tries = 0;; tries++) {
                if (bq != NOBLK && t < cur_cnt) break;
                if (own && bq != NOBLK && bq + 8u < a.B) {
                    bq += 8u;
                } else {
                    own = false;
                    uint32_t best = 0, bb = NOBLK;
                    for (uint32_t b0 = 0; b0 < a.B; b0 += 64) {
                        const uint32_t bx = b0 + (uint32_t)lane;
                        uint32_t left = 0;
                        if (bx < a.B && ms_block_is_mid(a.np, spans, bx)) {
                            const uint32_t c = min(a.ucount[bx], MS_UNIT_CAP);
                            const uint32_t k = __hip_atomic_load(a.ticket + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            left = c > k ? c - k : 0u;
                        }
                        const uint32_t mx = wave_all_max(left);
                        if (mx > best) {
                            best = mx;
                            const u64 who = __ballot(left == mx);
                            bb = b0 + (uint32_t)__ffsll((long long)who) - 1u;
                        }
                    }
                    bq = bb;
                    if (bq == NOBLK || tries > 4096u) {
                        bq = NOBLK;
                        break;
                    }
                }
                cur_cnt = ms_block_is_mid(a.np, spans, bq) ? min(a.ucount[bq], MS_UNIT_CAP) : 0u;
                uint32_t tt = 0;
                if (lane == 0 && cur_cnt) tt = atomicAdd(a.ticket + bq, 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tt);
            }
            cur_blk = bq;
            if (lane == 0) {
                s_blk = bq;
                s_unit = t;
            }
        }
        __syncthreads();
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_blk), u = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_unit);
        if (b == NOBLK) break;
        cur_blk = b;
        uint32_t pend_ticket = 0;
        if (tid == 0) pend_ticket = atomicAdd(a.ticket + b, 1u); // (consumed at the end of the unit)
        const uint2 ud = a.tiles[(size_t)b * MS_UNIT_CAP + u];
        const uint32_t start = ud.x, len = ud.y;
        const uint32_t blen = a.len[b];
        if (len > (uint32_t)MS_TILE || start + len > blen) { // (not what mid_plan writes)
            if (tid == 0) {
                atomicOr(a.err, ERR_MSD);
                s_next = pend_ticket;
            }
            __syncthreads();
            continue;
        }
        if (len == 0) { // (never written)
            if (tid == 0) s_next = pend_ticket;
            __syncthreads();
            continue;
        }
        const uint32_t n = a.n[b], h = a.hb[b];
        u64 *recs = a.list + (size_t)b * a.S + start;
        const uint32_t *rank = a.rank + (size_t)b * a.S;
        const int R = (int)((len + 511u) / 512u);
        const uint32_t Lw = (uint32_t)R * 64u;
        if (tid < 128) HM[tid] = 0ull;
        ms_clear(cur[0], tid);
        __syncthreads();
        // ---- load (slot p = wave * Lw + k * 64 + lane), key2, group heads
        u64 x[MS_ITEMS];
        uint32_t actmask = 0;
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            const uint32_t p = (uint32_t)wave * Lw + (uint32_t)k * 64u + (uint32_t)lane;
            const bool act = k < R && p < len;
            actmask |= (act ? 1u : 0u) << k;
            x[k] = recs[act ? p : 0u];
        }
        // (the record before my wavefront's first slot: the head test of that slot compares with it -- requested with the rest)
        u64 xprev = 0ull;
        if (lane == 0 && wave > 0 && (uint32_t)wave * Lw < len) xprev = recs[(uint32_t)wave * Lw - 1u];
        uint32_t k2[MS_ITEMS];
#pragma unroll
        for (int k = 0; k < MS_ITEMS; k++) {
            k2[k] = 0;
            if ((actmask >> k) & 1u) {
                const uint32_t i = (uint32_t)(x[k] & SUF_MASK);
                if (h < n) {
                    uint32_t i2 = i + h;
                    if (i2 >= n) i2 -= n;
                    k2[k] = BZH_DBG(a.dbg & 2u) ? i2 * 2654435761u >> 12 : rank[rslot(i2)];
                } else {
                    k2[k] = n - 1u - i; // identical rotations: larger index first -- a plain rank, no tag
                }
            }
        }
        {
            uint32_t last = (uint32_t)(xprev >> 40) & 0xFFFFFu; // (lane 0: the rank in the slot before the row)
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                if (k < R) {
                    const uint32_t p = (uint32_t)wave * Lw + (uint32_t)k * 64u + (uint32_t)lane;
                    const bool act = (actmask >> k) & 1u;
                    const uint32_t r = (uint32_t)(x[k] >> 40) & 0xFFFFFu;
                    uint32_t pr = (uint32_t)__shfl_up((int)r, 1, 64);
                    if (lane == 0) pr = last;
                    const bool head = act && (p == 0u || pr != r);
                    const u64 hm = __ballot(head);
                    if (lane == 0) HM[wave * R + k] = hm;
                    last = (uint32_t)__shfl((int)r, 63, 64);
                }
            }
        }
        __syncthreads();
        if (wave == 0) { // group heads before every row of 64 slots
            const uint32_t p0 = (uint32_t)__popcll(HM[2 * lane]), p1 = (uint32_t)__popcll(HM[2 * lane + 1]);
            const uint32_t inc = wave_incl_add(p0 + p1, lane);
            rowpre[2 * lane] = inc - p0 - p1;
            rowpre[2 * lane + 1] = inc - p1;
            if (lane == 63) ls[MS_NW + 1] = inc;
        }
        __syncthreads();
        if (ls[MS_NW + 1] > 128u && tid == 0) atomicOr(a.err, ERR_MSD); // (a large group has more than TAIL_G members)
        {
            u64 o = 0ull, an = ~0ull;
#pragma unroll
            for (int k = 0; k < MS_ITEMS; k++) {
                if ((actmask >> k) & 1u) {
                    const uint32_t p = (uint32_t)wave * Lw + (uint32_t)k * 64u + (uint32_t)lane, row = p >> 6;
                    const u64 hm = HM[row];
                    const uint32_t gi = (rowpre[row] + (uint32_t)__popcll(hm & ((2ull << (p & 63u)) - 1ull)) - 1u) & 127u;
                    const uint32_t r = (uint32_t)(x[k] >> 40) & 0xFFFFFu;
                    if ((hm >> (p & 63u)) & 1ull) G[gi] = r;
                    const uint32_t key = h < n ? rank_at(k2[k], a.tag) : k2[k];
                    x[k] = ((u64)gi << 41) | ((u64)key << 20) | (x[k] & SUF_MASK);
                    o |= x[k];
                    an &= x[k];
                }
            }
            o = ((u64)wave_all_or((uint32_t)(o >> 32)) << 32) | wave_all_or((uint32_t)o);
            an = ((u64)wave_all_and((uint32_t)(an >> 32)) << 32) | wave_all_and((uint32_t)an);
            if (lane == 0) {
                s_wo[wave] = o;
                s_wa[wave] = an;
            }
        }
        if (tid == 0) s_next = pend_ticket;
        __syncthreads();
        u64 vary;
        {
            u64 o = 0ull, an = ~0ull;
#pragma unroll
            for (int w = 0; w < MS_NW; w++) {
                o |= s_wo[w];
                an &= s_wa[w];
            }
            vary = o & ~an;
        }
        {
            uint32_t pos[MS_ITEMS / 2];
            int par = 0;
            bool staged = false;
#pragma unroll 1
            for (int pass = 0; pass < 4; pass++) {
                const int sh = 20 + 7 * pass; // key2: bits 20..39 (+ bit 40, always 0); group index: bits 41..47
                if (((vary >> sh) & 127ull) == 0ull || BZH_DBG(a.dbg & 1u)) continue;
                tile_rank<7>(x, sh, actmask, R, cur[par], cur[par ^ 1], ls, pos, tid);
                par ^= 1;
#pragma unroll
                for (int k = 0; k < MS_ITEMS; k++)
                    if ((actmask >> k) & 1u) stage[ms_slot((pos[k >> 1] >> (16 * (k & 1))) & 0xFFFFu)] = x[k];
                __syncthreads();
                staged = true;
                bool again = false;
                for (int q = pass + 1; q < 4; q++) again |= ((vary >> (20 + 7 * q)) & 127ull) != 0ull;
                if (again) {
#pragma unroll
                    for (int k = 0; k < MS_ITEMS; k++) x[k] = stage[ms_slot(((uint32_t)wave * Lw + (uint32_t)k * 64u + (uint32_t)lane) & 8191u)];
                }
            }
            if (!staged) {
#pragma unroll
                for (int k = 0; k < MS_ITEMS; k++)
                    if ((actmask >> k) & 1u) stage[ms_slot((uint32_t)wave * Lw + (uint32_t)k * 64u + (uint32_t)lane)] = x[k];
                __syncthreads();
            }
        }
        // ---- out, in place: [rank : 20 @40][key2 : 20 @20][suffix : 20] (refine_one reads a block of this kind that way whatever the round's other lists carry)
        for (uint32_t q = tid; q < len && !BZH_DBG(a.dbg & 4u); q += MS_THREADS) {
            const u64 y = stage[ms_slot(q)];
            recs[q] = ((u64)G[(uint32_t)(y >> 41) & 127u] << 40) | (y & 0xFFFFFFFFFFull);
        }
        __syncthreads(); // the stage, G and s_unit are reused by the next unit
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
// Queues the bucket-first initial sort for every block of the batch that qualifies (bigram_plan decides on the device)
// and returns, in *n_old, how many blocks keep the 8-pass path (their ids: bt.ms_old[0 .. n_old), count also on the
// device in bt.ms_cnt[MC_OLD]).  Outputs for the blocks it handles: (rank word, suffix) pairs binned into `binned`,
// small-group lists in `tail`, big lists in `big`, c_small / c_big / c_groups -- what refine_one<init> leaves.
// X / Y: the two list buffers the partition levels alternate between (X also receives the 2-byte partition).
// In two halves: msd_sort_begin queues everything up to level 1 of the oversized buckets and returns as soon as the plan has
// reported (the blocks that keep the 8 passes are then queued on the second stream, beside all this); msd_sort_finish
// waits for level 1's report -- which arrives while its scatter runs -- queues the deeper levels only if a bucket is
// still oversized (text: mostly none; twelve empty launches were 68 us of every step), then the finishing kernel.
// (seg_plan: one oversized bucket a workgroup -- a bucket is a serial chain of ~20 us, the headline has 922 at level 1)
constexpr uint32_t SEG_PLAN_WGS = 1024;
static int msd_sort_begin(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal, u64 *X, u64 *Y, u64 *big, u64 *tail, u64 *binned,
                          bool force_old, uint32_t fuse, volatile uint32_t *hrec, uint32_t seq, uint32_t *n_old, hipEvent_t ev_plan, Msd *keep)
{
    Batch &bt = ctx->bt;
    hipStream_t st = ctx->stream;
    Msd &m = *keep;
    m = Msd{};
    m.blk = bt.rle;
    m.n = bt.n;
    m.S = bt.S;
    m.B = B;
    m.bgcur = bt.ms_bgcur;
    m.pool = bt.ms_pool;
    m.segcur = bt.ms_segcur;
    m.units = bt.ms_units;
    m.segs = bt.ms_segs;
    m.items = bt.ms_items;
    m.cnt = bt.ms_cnt;
    m.np = bt.ms_np;
    m.act_old = bt.ms_old;
    m.act_new = bt.ms_new;
    m.bincur = bt.ms_bincur;
    m.bufX = X;
    m.bufY = Y;
    m.big = big;
    m.tail = tail;
    m.binned = binned;
    m.bwt = bt.bwt;
    m.c_big = bt.c_big;
    m.c_small = bt.c_small;
    m.c_groups = bt.c_groups;
    m.err = bt.errflag;
    m.gout = GidOut{bt.gidof, bt.grank, bt.gcount, bt.gwide, bt.S, bt.B, 0u};
    m.runq = msc_row(bt.ms_cnt, B, MSR_RUNQ); // (round 0 reads the half of parity 0)
    m.runs = msc_runs(bt.ms_cnt, B, 0);
    m.tiles = msc_tiles(bt.ms_cnt, B);
    m.ntiles = msc_row(bt.ms_cnt, B, MSR_MTILES);
    m.mticket = msc_row(bt.ms_cnt, B, MSR_MTICKET);
    m.midcap = (uint32_t)MS_TILE;
#ifdef BZH_EXPERIMENTS // (a cap below 8192 also breaks the 2 L / SORT_TILE + 1 tile bound refine_one's launch is sized with)
    if (getenv("BZH_MID_CAP")) m.midcap = std::min<uint32_t>(MS_TILE, std::max(1, atoi(getenv("BZH_MID_CAP"))));
#endif
    m.force_old = force_old ? 1u : 0u;
    m.fuse = fuse;
    {
        static const bool init_msd = []() {
            const char *e = getenv("BZH_INIT");
            return e && !strcmp(e, "msd");
        }();
        m.force_new = init_msd ? 1u : 0u;
    }
    {   // (16 = cycles per phase of chunk_finish, bit-exact; the bits that leave work out exist with -DBZH_EXPERIMENTS only)
        static const uint32_t msd_dbg = getenv("BZH_MSD_DBG") ? (uint32_t)atoi(getenv("BZH_MSD_DBG")) : 0u;
#ifdef BZH_EXPERIMENTS
        m.dbg = msd_dbg;
#else
        m.dbg = msd_dbg & 16u;
#endif
    }
    // (bt.ms_cnt, bt.ms_bincur and bt.ms_bgcur arrive cleared: bwt_run's one clearing launch)
    {
        KSpan ks(ctx, K_MSD_PLAN, force_old ? 0 : ntotal, 2);
        if (!force_old) {
            bigram_hist<<<dim3(BGH_SEGS, B), 1024, 0, st>>>(m);
        }
        bigram_plan<<<dim3(B), 1024, 0, st>>>(m, const_cast<uint32_t *>(hrec), seq);
        if (ev_plan) hipEventRecord(ev_plan, st); // (what the blocks that keep the 8 passes wait for, on the second stream)
    }
    if (!force_old) {
        const Lst nl{bt.ms_new, bt.ms_cnt + MC_NEW, B};
        const uint32_t tiles = (nmax + MS_TILE - 1) / MS_TILE;
        const uint32_t T = tiles | (few_blocks(B) ? WG_SPREAD : 0u);
        {
            KSpan ks(ctx, K_MSD_SCATTER, 9 * ntotal);
            bigram_scatter<<<dim3(xcd_grid(T, B)), MS_THREADS, 0, st>>>(m, T, nl);
        }
        {
            KSpan ks(ctx, K_MSD_LEVELS, 0, 3 * MS_LEVELS);
            seg_count<<<dim3(1024), MS_THREADS, 0, st>>>(m, 1);
            seg_plan<<<dim3(SEG_PLAN_WGS), 256, 0, st>>>(m, 1, const_cast<uint32_t *>(hrec), seq);
            seg_scatter<<<dim3(1024), MS_THREADS, 0, st>>>(m, 1);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    // how many blocks keep the 8-pass path: bigram_plan's record (it ran long ago: the host waits for the first kernels only)
    for (uint64_t it = 0, idle = 0; hrec[SUMMARY_WORDS - 1] != seq; it++) {
        if ((it & 0xFFFu) == 0xFFFu) {
            const hipError_t e = hipStreamQuery(st);
            if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(ctx, e);
            if (e == hipSuccess && ++idle > 64) {
                bzh_set_error(ctx, "BWT: the plan of the initial sort never reported (internal error)");
                return BZH_E_HIP;
            }
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *n_old = hrec[0];
    return BZH_OK;
}

static int msd_sort_finish(bzh_ctx *ctx, hipStream_t st, const Msd &m, uint64_t ntotal, volatile uint32_t *hrec, uint32_t seq, bool *deeper_out)
{
    *deeper_out = false;
    if (m.force_old) return BZH_OK;
    {
        KSpan ks(ctx, K_MSD_LEVELS, 0, 3 * (MS_LEVELS - 1));
        // what level 1 left oversized (it reports while its scatter runs)
        for (uint64_t it = 0, idle = 0; hrec[5] != seq; it++) {
            if ((it & 0xFFFu) == 0xFFFu) {
                const hipError_t e = hipStreamQuery(st);
                if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(ctx, e);
                if (e == hipSuccess && ++idle > 64) {
                    bzh_set_error(ctx, "BWT: level 1 of the initial sort never reported (internal error)");
                    return BZH_E_HIP;
                }
            }
            __builtin_ia32_pause();
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const uint32_t deeper = hrec[4];
        *deeper_out = deeper != 0u; // (only level 5 makes units out of a group that spans several: without the deeper levels no block has one)
        for (uint32_t L = 2; L <= MS_LEVELS && deeper; L++) {
            seg_count<<<dim3(1024), MS_THREADS, 0, st>>>(m, L);
            seg_plan<<<dim3(SEG_PLAN_WGS), 256, 0, st>>>(m, L);
            seg_scatter<<<dim3(1024), MS_THREADS, 0, st>>>(m, L);
        }
    }
    {
        KSpan ks(ctx, K_MSD_FINISH, 16 * ntotal);
        chunk_finish<<<dim3(512), MS_THREADS, 0, st>>>(m);
        mid_plan<<<dim3(m.B), 256, 0, st>>>(m, 1u);
    }
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
