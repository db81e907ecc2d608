// bwt.hip -- batched wrap-around Burrows-Wheeler transform on gfx950.
//
// Replaces bwt::bwt (reference lib/bwt.rs:526-756: SA-IS on the doubled block).  Output
// contract: bwt[k] = S[(sa_k - 1) mod n] for the rotations of S in order, ties
// between identical rotations broken by DESCENDING start index (lib/bwt.rs:564-573 sorts S||S);
// ptr = k with sa_k = 0; has_byte[c] = c in S.
//
// Method: cyclic prefix doubling.  All bzip2 blocks of a batch are sorted at once
// (blockIdx.y = bzip2 block).  An initial LSD radix sort on the 4-byte cyclic prefix gives
// SA_4 / rank_4 (rank = first SA position of the suffix's group).  A doubling round with depth h
// needs every unresolved group ordered by rank[i+h]; instead of sorting on that second key, the
// unresolved suffixes are ENUMERATED in SA order of suffix i+h (one coalesced sweep of SA plus a
// rank gather) and stably radix-sorted by their own group rank only (20 bits -> 3 passes of 7
// bits).  Stability keeps the enumeration order inside each group, which is exactly the order by
// rank[i+h].  Group boundaries are re-flagged, ranks refined, resolved suffixes drop out.
// When h >= n the survivors are identical rotations: one last round enumerates them by
// descending index (the reference's tie rule).
//
// Kernels (all integer, HBM/LDS bound, no MFMA):
//   radix_hist     per-tile digit histogram in LDS            -> hist[b][digit][tile]
//   radix_scan     per-block exclusive scan of hist (1 workgroup per bzip2 block)
//   radix_scatter  stable scatter: wave match-any ranking + per-wave LDS cursors
//   flag_tiles / flag_carry / refine   boundary flags, max-scan of group heads, rank + SA update
//   bwt_emit       last column, ptr, has_byte
#include "common.h"

enum GenMode : int {
    GEN_BYTES4 = 0, // element e is suffix e, key = 4-byte big-endian cyclic prefix
    GEN_ROUND = 1,  // doubling round: SA-order (h < n) or descending-index (h >= n) enumeration
    GEN_LIST = 2    // element e is src[e]
};

struct SortArgs {
    const uint8_t *blk;    // [B][S]
    const uint32_t *n;     // [B]
    const uint32_t *cnt;   // [B] elements enumerated this pass
    const uint32_t *gate;  // [B] skip block when 0
    const uint32_t *rank;  // [B][S]
    const uint32_t *sa;    // [B][S]
    const uint2 *src;      // [B][S]
    uint2 *dst;            // [B][S]
    uint32_t *hist;        // [B][NBMAX*TPB]
    uint32_t S, TPB, h, shift;
};

constexpr int NBMAX = 256;

template <int MODE>
__device__ __forceinline__ bool gen_elem(const SortArgs &a, uint32_t b, uint32_t e, uint32_t n, uint32_t &key,
                                         uint32_t &val)
{
    const size_t base = (size_t)b * a.S;
    if (MODE == GEN_BYTES4) {
        const uint8_t *s = a.blk + base;
        uint32_t i0 = e, i1 = e + 1, i2 = e + 2, i3 = e + 3;
        if (i3 >= n) { // cyclic wrap (n may be smaller than 4)
            i1 %= n;
            i2 %= n;
            i3 %= n;
        }
        key = ((uint32_t)s[i0] << 24) | ((uint32_t)s[i1] << 16) | ((uint32_t)s[i2] << 8) | (uint32_t)s[i3];
        val = e;
        return true;
    } else if (MODE == GEN_ROUND) {
        uint32_t i;
        if (a.h < n) { // suffix j = sa[e] is the e-th smallest; i = j - h sees it as its second half
            uint32_t j = a.sa[base + e];
            i = j >= a.h ? j - a.h : j + n - a.h;
        } else { // identical rotations: larger index first
            i = n - 1 - e;
        }
        uint32_t r = a.rank[base + i];
        key = r;
        val = i;
        return (r & RANK_RESOLVED) == 0;
    } else {
        uint2 kv = a.src[base + e];
        key = kv.x;
        val = kv.y;
        return true;
    }
}

template <int BITS, int MODE>
__global__ void __launch_bounds__(SORT_THREADS) radix_hist(SortArgs a)
{
    constexpr int NB = 1 << BITS;
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    if (a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b], n = a.n[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    __shared__ uint32_t h[NB];
    for (int k = threadIdx.x; k < NB; k += SORT_THREADS) h[k] = 0;
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t e = tile * SORT_TILE + k * SORT_THREADS + threadIdx.x;
        if (e < cnt) {
            uint32_t key, val;
            if (gen_elem<MODE>(a, b, e, n, key, val)) atomicAdd(&h[(key >> a.shift) & (NB - 1)], 1u);
        }
    }
    __syncthreads();
    uint32_t *out = a.hist + (size_t)b * NBMAX * a.TPB;
    for (int k = threadIdx.x; k < NB; k += SORT_THREADS) out[(size_t)k * a.TPB + tile] = h[k];
}

// One workgroup per bzip2 block: exclusive scan of hist over (digit major, tile minor).
template <int BITS>
__global__ void __launch_bounds__(1024) radix_scan(SortArgs a)
{
    constexpr int NB = 1 << BITS;
    const uint32_t b = blockIdx.x;
    if (a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    const uint32_t total = NB * ntile;
    uint32_t *hist = a.hist + (size_t)b * NBMAX * a.TPB;
    __shared__ uint32_t lds[20];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < total; base += 1024) {
        uint32_t e = base + threadIdx.x;
        uint32_t addr = 0, v = 0;
        if (e < total) {
            uint32_t bin = e / ntile, t = e - bin * ntile;
            addr = bin * a.TPB + t;
            v = hist[addr];
        }
        uint32_t tot;
        uint32_t ex = block_excl_add(v, lds, &tot);
        if (e < total) hist[addr] = carry + ex;
        carry += tot;
    }
}

template <int BITS, int MODE>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter(SortArgs a)
{
    constexpr int NB = 1 << BITS;
    constexpr int NW = SORT_THREADS / 64;
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    if (a.gate[b] == 0) return;
    const uint32_t cnt = a.cnt[b], n = a.n[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    __shared__ uint32_t cur[NW][NB]; // per-wave write cursors
    for (int k = threadIdx.x; k < NW * NB; k += SORT_THREADS) (&cur[0][0])[k] = 0;

    // wave w owns the contiguous run [w*ITEMS*64, (w+1)*ITEMS*64) of the tile, 64 elements a step
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
    uint32_t actmask = 0;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t e = tile * SORT_TILE + wave * (SORT_ITEMS * 64) + k * 64 + lane;
        key[k] = 0;
        val[k] = 0;
        if (e < cnt && gen_elem<MODE>(a, b, e, n, key[k], val[k])) actmask |= 1u << k;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++)
        if (actmask & (1u << k)) atomicAdd(&cur[wave][(key[k] >> a.shift) & (NB - 1)], 1u);
    __syncthreads();
    {
        const uint32_t *hist = a.hist + (size_t)b * NBMAX * a.TPB;
        for (int bin = threadIdx.x; bin < NB; bin += SORT_THREADS) {
            uint32_t g = hist[(size_t)bin * a.TPB + tile];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                uint32_t t = cur[w][bin];
                cur[w][bin] = g;
                g += t;
            }
        }
    }
    __syncthreads();
    uint2 *dst = a.dst + (size_t)b * a.S;
    volatile uint32_t *mycur = cur[wave];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const bool act = (actmask >> k) & 1u;
        const uint32_t d = (key[k] >> a.shift) & (NB - 1);
        uint64_t m = __ballot(act);
#pragma unroll
        for (int bit = 0; bit < BITS; bit++) {
            const bool one = (d >> bit) & 1u;
            const uint64_t bm = __ballot(act && one);
            m &= one ? bm : ~bm;
        }
        if (act) {
            const uint32_t basepos = mycur[d];
            const uint32_t off = __popcll(m & lt);
            if (off == 0) mycur[d] = basepos + __popcll(m); // lowest lane of the digit group advances
            dst[basepos + off] = make_uint2(key[k], val[k]);
        }
    }
}

// ---- group refinement ---------------------------------------------------------------------------
struct RefineArgs {
    const uint32_t *n;    // [B]
    const uint32_t *cnt;  // [B] list length (n for the init pass, active count in rounds)
    const uint2 *list;    // [B][S] sorted (group rank | key, suffix)
    uint32_t *rank;       // [B][S]
    uint32_t *sa;         // [B][S]
    uint8_t *flg;         // [B][S]
    int2 *tagg;           // [B][TPB]
    uint32_t *nact_next;  // [B]
    uint32_t S, TPB, h;
    int init;
};

// flag bit0: first element of its (old) group; bit1: first element of its refined group
__global__ void __launch_bounds__(SORT_THREADS) flag_tiles(RefineArgs a)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t cnt = a.cnt[b], n = a.n[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    const size_t base = (size_t)b * a.S;
    const uint2 *list = a.list + base;
    const uint32_t *rank = a.rank + base;
    const bool desc = !a.init && a.h >= n;
    const uint32_t q0 = tile * SORT_TILE + threadIdx.x * SORT_ITEMS;

    uint32_t packed[SORT_ITEMS / 4] = {0, 0, 0, 0};
    int lastgs = -1, lastbd = -1;
    if (q0 < cnt) {
        uint2 prev = make_uint2(0, 0);
        uint32_t prevk2 = 0;
        if (q0 > 0) {
            prev = list[q0 - 1];
            if (!a.init && !desc) {
                uint32_t i2 = prev.y + a.h;
                if (i2 >= n) i2 -= n;
                prevk2 = rank[i2] & ~RANK_RESOLVED;
            }
        }
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; k++) {
            const uint32_t q = q0 + k;
            if (q < cnt) {
                const uint2 cur = list[q];
                uint32_t k2 = 0;
                bool gs, bd;
                if (a.init) {
                    gs = (q == 0);
                    bd = gs || cur.x != prev.x;
                } else {
                    gs = (q == 0) || cur.x != prev.x;
                    if (desc) {
                        bd = true;
                    } else {
                        uint32_t i2 = cur.y + a.h;
                        if (i2 >= n) i2 -= n;
                        k2 = rank[i2] & ~RANK_RESOLVED;
                        bd = gs || k2 != prevk2;
                    }
                }
                if (gs) lastgs = (int)q;
                if (bd) lastbd = (int)q;
                packed[k >> 2] |= ((gs ? 1u : 0u) | (bd ? 2u : 0u)) << ((k & 3) * 8);
                prev = cur;
                prevk2 = k2;
            }
        }
        uint8_t *f = a.flg + base + q0; // q0 is a multiple of 16 and S a multiple of 4096
        *reinterpret_cast<uint4 *>(f) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
    // tile aggregate: max over threads
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
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (ntile == 0) return;
    int2 *t = a.tagg + (size_t)b * a.TPB;
    __shared__ int l0[16], l1[16];
    __shared__ int inc0[1024], inc1[1024];
    const uint32_t e = threadIdx.x;
    int2 v = e < ntile ? t[e] : make_int2(-1, -1);
    inc0[e] = block_incl_max(v.x, l0);
    inc1[e] = block_incl_max(v.y, l1);
    __syncthreads();
    if (e < ntile) t[e] = e == 0 ? make_int2(-1, -1) : make_int2(inc0[e - 1], inc1[e - 1]);
}

__global__ void __launch_bounds__(SORT_THREADS) refine(RefineArgs a)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t cnt = a.cnt[b];
    const uint32_t ntile = (cnt + SORT_TILE - 1) / SORT_TILE;
    if (tile >= ntile) return;
    const size_t base = (size_t)b * a.S;
    const uint2 *list = a.list + base;
    const uint32_t q0 = tile * SORT_TILE + threadIdx.x * SORT_ITEMS;

    uint32_t packed[4] = {0, 0, 0, 0};
    uint32_t nextflag = 2; // flag of element q0+16 (end of list counts as a boundary)
    if (q0 < cnt) {
        uint4 v = *reinterpret_cast<const uint4 *>(a.flg + base + q0);
        packed[0] = v.x;
        packed[1] = v.y;
        packed[2] = v.z;
        packed[3] = v.w;
        if (q0 + SORT_ITEMS < cnt) nextflag = a.flg[base + q0 + SORT_ITEMS];
    }
    // per-thread last flagged index, then workgroup inclusive max-scan -> carry for each thread
    int tg = -1, td = -1;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
        if (q0 + k < cnt) {
            if (f & 1u) tg = (int)(q0 + k);
            if (f & 2u) td = (int)(q0 + k);
        }
    }
    __shared__ int l0[SORT_THREADS / 64], l1[SORT_THREADS / 64];
    int ig = block_incl_max(tg, l0);
    int id = block_incl_max(td, l1);
    // exclusive carry = inclusive value of the previous thread (or the tile carry)
    __shared__ int ex0[SORT_THREADS], ex1[SORT_THREADS];
    ex0[threadIdx.x] = ig;
    ex1[threadIdx.x] = id;
    __syncthreads();
    const int2 tc = a.tagg[(size_t)b * a.TPB + tile];
    int cg = tc.x, cd = tc.y;
    if (threadIdx.x > 0) {
        cg = max(cg, ex0[threadIdx.x - 1]);
        cd = max(cd, ex1[threadIdx.x - 1]);
    }
    if (q0 >= cnt) return;

    uint32_t *rank = a.rank + base;
    uint32_t *sa = a.sa + base;
    uint32_t unresolved = 0;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const uint32_t q = q0 + k;
        if (q < cnt) {
            const uint32_t f = (packed[k >> 2] >> ((k & 3) * 8)) & 3u;
            const uint32_t fn = (k + 1 < SORT_ITEMS) ? ((packed[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 3u) : nextflag;
            if (f & 1u) cg = (int)q;
            if (f & 2u) cd = (int)q;
            const uint2 cur = list[q];
            const uint32_t gbase = a.init ? 0u : (cur.x - (uint32_t)cg); // SA position of list entry 0 of the group, minus its list index
            const uint32_t pos = gbase + q;
            const uint32_t head = gbase + (uint32_t)cd;
            const bool single = (f & 2u) && ((q + 1 == cnt) || (fn & 2u));
            rank[cur.y] = single ? (head | RANK_RESOLVED) : head;
            sa[pos] = cur.y;
            unresolved += single ? 0u : 1u;
        }
    }
    unresolved = wave_reduce_add(unresolved);
    if ((threadIdx.x & 63) == 0 && unresolved) atomicAdd(&a.nact_next[b], unresolved);
}

// ---- last column ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bwt_emit(Batch bt)
{
    const uint32_t b = blockIdx.y;
    const uint32_t n = bt.n[b];
    const size_t base = (size_t)b * bt.S;
    const uint8_t *s = bt.rle + base;
    const uint32_t *sa = bt.sa + base;
    uint8_t *out = bt.bwt + base;
    __shared__ uint32_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    // 4 consecutive positions per thread -> one 32-bit store
    for (uint32_t p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 < n; p0 += gridDim.x * 256 * 4) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t p = p0 + k;
            if (p < n) {
                uint32_t j = sa[p];
                if (j == 0) bt.ptr[b] = p;
                uint32_t c = s[j ? j - 1 : n - 1];
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

// ---- host driver -----------------------------------------------------------------------------------
template <int BITS, int MODE>
static void launch_pass(bzh_ctx *ctx, SortArgs &a, uint32_t B, uint32_t maxcnt, uint64_t elems)
{
    uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    radix_hist<BITS, MODE><<<dim3(tiles, B), SORT_THREADS, 0, ctx->stream>>>(a);
    radix_scan<BITS><<<dim3(B), 1024, 0, ctx->stream>>>(a);
    if (ctx->profiling) { // HIP events bracket the dominant kernel only
        e0 = bzh_event(ctx);
        hipEventRecord(e0, ctx->stream);
    }
    radix_scatter<BITS, MODE><<<dim3(tiles, B), SORT_THREADS, 0, ctx->stream>>>(a);
    if (ctx->profiling) {
        e1 = bzh_event(ctx);
        hipEventRecord(e1, ctx->stream);
        ctx->sort_spans.push_back({e0, e1});
        ctx->stats.bwt_sort_launches += 1;
        ctx->stats.bwt_sort_elems += elems; // (key, suffix) pairs this launch writes
    }
}

static void launch_refine(bzh_ctx *ctx, RefineArgs &r, uint32_t B, uint32_t maxcnt)
{
    uint32_t tiles = (maxcnt + SORT_TILE - 1) / SORT_TILE;
    if (tiles == 0) return;
    flag_tiles<<<dim3(tiles, B), SORT_THREADS, 0, ctx->stream>>>(r);
    flag_carry<<<dim3(B), 1024, 0, ctx->stream>>>(r);
    refine<<<dim3(tiles, B), SORT_THREADS, 0, ctx->stream>>>(r);
}

// Suffix-sort and emit the last column for blocks 0..B-1 of the batch (bt.rle / bt.n filled).
// nmax = largest block length in the batch, ntotal = sum of block lengths (statistics only).
int bwt_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal)
{
    Batch &bt = ctx->bt;
    if (B == 0) return BZH_OK;
    hipStream_t st = ctx->stream;

    SortArgs a{};
    a.blk = bt.rle;
    a.n = bt.n;
    a.rank = bt.rank;
    a.sa = bt.sa;
    a.hist = bt.hist;
    a.S = bt.S;
    a.TPB = bt.TPB;
    a.h = 0;

    // ---- initial sort on the 4-byte cyclic prefix: 4 passes of 8 bits -------------------------
    a.cnt = bt.n;
    a.gate = bt.n;
    a.shift = 0;
    a.src = nullptr;
    a.dst = bt.listA;
    launch_pass<8, GEN_BYTES4>(ctx, a, B, nmax, ntotal);
    uint2 *cur = bt.listA, *oth = bt.listB;
    for (int p = 1; p < 4; p++) {
        a.shift = 8 * p;
        a.src = cur;
        a.dst = oth;
        launch_pass<8, GEN_LIST>(ctx, a, B, nmax, ntotal);
        uint2 *t = cur;
        cur = oth;
        oth = t;
    }

    uint32_t *nact = bt.nactA, *nact_next = bt.nactB;
    HIP_TRY(ctx, hipMemsetAsync(nact_next, 0, B * sizeof(uint32_t), st));

    RefineArgs r{};
    r.n = bt.n;
    r.cnt = bt.n;
    r.list = cur;
    r.rank = bt.rank;
    r.sa = bt.sa;
    r.flg = bt.flg;
    r.tagg = bt.tagg;
    r.nact_next = nact_next;
    r.S = bt.S;
    r.TPB = bt.TPB;
    r.h = 0;
    r.init = 1;
    launch_refine(ctx, r, B, nmax);

    // ---- doubling rounds ---------------------------------------------------------------------------
    uint32_t h = 4;
    uint32_t *hact = ctx->h_pinned;
    for (int round = 0; round < 40; round++) {
        HIP_TRY(ctx, hipMemcpyAsync(hact, nact_next, B * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        uint32_t maxact = 0;
        uint64_t sum = 0;
        for (uint32_t b = 0; b < B; b++) {
            maxact = hact[b] > maxact ? hact[b] : maxact;
            sum += hact[b];
        }
        if (maxact == 0) break;
        ctx->stats.bwt_active_sum += sum;
        ctx->stats.bwt_rounds = (uint64_t)(round + 1) > ctx->stats.bwt_rounds ? (uint64_t)(round + 1) : ctx->stats.bwt_rounds;
        { // swap counters
            uint32_t *t = nact;
            nact = nact_next;
            nact_next = t;
        }
        HIP_TRY(ctx, hipMemsetAsync(nact_next, 0, B * sizeof(uint32_t), st));

        a.h = h;
        a.gate = nact;
        // pass 0: enumerate unresolved suffixes in order of their second half, bucket by rank bits 0..6
        a.cnt = bt.n;
        a.shift = 0;
        a.src = nullptr;
        a.dst = bt.listA;
        launch_pass<7, GEN_ROUND>(ctx, a, B, nmax, sum);
        // passes 1, 2 over the compact list
        a.cnt = nact;
        a.shift = 7;
        a.src = bt.listA;
        a.dst = bt.listB;
        launch_pass<7, GEN_LIST>(ctx, a, B, maxact, sum);
        a.shift = 14;
        a.src = bt.listB;
        a.dst = bt.listA;
        launch_pass<7, GEN_LIST>(ctx, a, B, maxact, sum);

        r.cnt = nact;
        r.list = bt.listA;
        r.nact_next = nact_next;
        r.h = h;
        r.init = 0;
        launch_refine(ctx, r, B, maxact);

        if (h < (1u << 30)) h <<= 1;
    }

    HIP_TRY(ctx, hipMemsetAsync(bt.hasbyte, 0, (size_t)B * 256, st));
    uint32_t gx = (nmax + 1023) / 1024;
    if (gx > 256) gx = 256;
    if (gx == 0) gx = 1;
    bwt_emit<<<dim3(gx, B), 256, 0, st>>>(bt);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
