// huffman.hip -- Huffman stage + bit packing for a batch of blocks.
//
// Replaces huffman::encode (reference lib/huffman.rs:313-575) and the framing writes of
// lib/lib.rs:24-64 / lib/out.rs.  The reference's behaviour is reproduced, not
// libbz2's:
//   * 2 tables if num_syms <= 199 else 3 (lib/huffman.rs:319-326)
//   * iteration 0 assigns each 50-symbol segment to the table whose initial range holds the
//     FEWEST of its symbols (length 15 inside the range, 0 outside; :364-372, :424-438)
//   * iterations 1-3 zero the length tables, so table 0 wins every segment and its frequency
//     list grows by the global histogram each time (:402-409, :441-443): final frequencies are
//     f0_0 + 3F for table 0 and f0_t otherwise, every selector is 0, all symbols use table 0
//   * code lengths from the reference's own binary heap, operation for operation (:165-298)
//
// Kernels: huff_init (ranges), huff_segments (segment classification + per-table histograms),
// huff_build (exact heap in LDS, one wavefront per table working in lock step), huff_header
// (block header, symbol map, coding tables as a bit string; canonical codes; bit totals),
// block_scan, pack_tilebits / pack_tilescan / pack_symbols (prefix-sum placed parallel pack through
// an LDS word buffer), pack_headers.
#include "common.h"

constexpr int HUF_SYMS = 258;
constexpr int HUF_MAXLEN = 17; // lib/huffman.rs:13
constexpr int SEG = 50;        // lib/huffman.rs:310
constexpr uint32_t HDR_A = 64; // bytes reserved for the part before the selectors

// ---- initial ranges (lib/huffman.rs:333-376) -----------------------------------------------------------
__global__ void __launch_bounds__(64) huff_init(Batch bt, uint32_t *ranges)
{
    const uint32_t b = blockIdx.x;
    uint32_t *tf = bt.tfreq + (size_t)b * 3 * HUF_SYMS;
    for (int k = threadIdx.x; k < 3 * HUF_SYMS; k += 64) tf[k] = 0;
    if (threadIdx.x != 0) return;
    const uint32_t nsyms = bt.nsyms[b];
    const uint32_t *F = bt.freqs + (size_t)b * HUF_SYMS;
    const uint32_t ntab = nsyms <= 199 ? 2 : 3;
    bt.ntab[b] = ntab;
    uint32_t remaining = bt.m[b], left = 0;
    uint32_t *r = ranges + (size_t)b * 8;
    for (uint32_t t = 0; t < ntab; t++) {
        const uint32_t target = remaining / (ntab - t);
        uint32_t acc = 0, right = left;
        for (;;) {
            acc += F[right];
            if (acc >= target || right + 1 == nsyms) break;
            right++;
        }
        if (right > left && t != 0 && t != ntab - 1 && (t & 1u)) {
            acc -= F[right];
            right--;
        }
        r[2 * t] = left;
        r[2 * t + 1] = right;
        left = right + 1;
        remaining -= acc;
    }
    if (ntab == 2) {
        r[4] = 1; // empty range
        r[5] = 0;
    }
}

// ---- iteration 0: classify segments, accumulate per-table histograms (lib/huffman.rs:411-454) -------
__global__ void __launch_bounds__(256) huff_segments(Batch bt, const uint32_t *ranges)
{
    const uint32_t b = blockIdx.y;
    const uint32_t m = bt.m[b];
    const uint32_t nseg = (m + SEG - 1) / SEG;
    const uint32_t seg0 = blockIdx.x * 256;
    if (seg0 >= nseg) return;
    __shared__ uint32_t h[3][HUF_SYMS];
    for (int k = threadIdx.x; k < 3 * HUF_SYMS; k += 256) (&h[0][0])[k] = 0;
    __syncthreads();
    const uint32_t ntab = bt.ntab[b];
    const uint32_t *r = ranges + (size_t)b * 8;
    const uint32_t l0 = r[0], r0 = r[1], l1 = r[2], r1 = r[3], l2 = r[4], r2 = r[5];
    const uint32_t seg = seg0 + threadIdx.x;
    if (seg < nseg) {
        const uint16_t *s = bt.syms + (size_t)b * (bt.S + 64) + (size_t)seg * SEG;
        const uint32_t len = m - seg * SEG < SEG ? m - seg * SEG : SEG;
        uint32_t c0 = 0, c1 = 0, c2 = 0;
        // 50 u16 = 100 bytes, 4-byte aligned: read as 25 words
        uint32_t w[SEG / 2];
#pragma unroll
        for (int k = 0; k < SEG / 2; k++) w[k] = reinterpret_cast<const uint32_t *>(s)[k];
#pragma unroll
        for (int k = 0; k < SEG; k++) {
            const uint32_t v = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if ((uint32_t)k < len) {
                c0 += (v >= l0 && v <= r0);
                c1 += (v >= l1 && v <= r1);
                c2 += (v >= l2 && v <= r2);
            }
        }
        uint32_t best = 0, cost = c0; // cost_t = 15 * count_t; first strict minimum wins
        if (c1 < cost) {
            best = 1;
            cost = c1;
        }
        if (ntab == 3 && c2 < cost) best = 2;
#pragma unroll
        for (int k = 0; k < SEG; k++) {
            const uint32_t v = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if ((uint32_t)k < len) atomicAdd(&h[best][v], 1u);
        }
    }
    __syncthreads();
    uint32_t *tf = bt.tfreq + (size_t)b * 3 * HUF_SYMS;
    for (int k = threadIdx.x; k < 3 * HUF_SYMS; k += 256) {
        uint32_t v = (&h[0][0])[k];
        if (v) atomicAdd(&tf[k], v);
    }
}

// ---- exact heap (lib/huffman.rs:161-298) ---------------------------------------------------------------
// The reference's binary heap decides, through the positions at which equal priorities happen to sit,
// which symbol gets which length, so it is replayed operation for operation -- but by a whole
// wavefront in lock step instead of one lane chasing pointers through LDS:
//   insert  (:196-222): every ancestor slot idx>>1, idx>>2, ... of the new leaf is known up front, so the
//           lanes fetch all of them at once, a ballot says how far the new priority rises, and the passed
//           ancestors drop one level in a single store -- one LDS round trip per insert;
//   extract (:225-267): 30 lanes fetch the four levels below the hole (2 + 4 + 8 + 16 slots) at once; each
//           compares with its sibling (right child only if strictly smaller), two ballots describe the
//           preferred children and where the moved element stops, the path is walked in scalar registers
//           and the path's slots rise one level in a single store -- one round trip per four levels.
// Control flow is uniform across the wavefront; the heap itself stays in LDS.
struct HeapMem {
    // one word per heap slot: ((weight << 8 | depth) << 10) | node id.  The reference orders by the
    // lexicographic Priority(usize, u8) alone -- ids never take part in a comparison (they are shifted
    // out).  weight <= 4 m + 258 < 2^22, so a priority fits 32 bits.
    uint64_t key[HUF_SYMS + 2];
    int16_t par[2 * HUF_SYMS]; // parent of every tree node (root = 0)
    uint32_t fr[HUF_SYMS];
};
constexpr int HEAP_ID_BITS = 10; // node ids < 2 * 258
#define HEAP_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ void heap_insert(HeapMem &h, uint32_t &len, uint32_t id, uint32_t pr, uint32_t lane) // :196-222
{
    const uint32_t idx = len + 1;
    len++;
    const uint32_t k = lane + 1;   // lane 0 looks at the parent, lane 1 at the grandparent, ...
    const uint32_t anc = k < 16 ? idx >> k : 0u; // 0: no such ancestor
    const uint64_t ak = anc ? h.key[anc] : 0ull;
    const bool lt = anc && pr < (uint32_t)(ak >> HEAP_ID_BITS);
    const uint64_t m = __ballot(lt);
    const uint32_t d = (uint32_t)__builtin_ctzll(~m); // levels the new entry rises (uniform)
    if (k <= d) h.key[idx >> (k - 1)] = ak;          // the passed ancestors drop one level
    if (lane == 0) h.key[idx >> d] = ((uint64_t)pr << HEAP_ID_BITS) | id;
    HEAP_ORDER();
}

// The lanes of a lane's ancestors inside the four levels heap_extract looks at (level 1 = lanes 0-1, 2 = 2-5,
// 3 = 6-13, 4 = 14-29): a lane is on the path of preferred children iff it is preferred and every ancestor is.
__device__ __forceinline__ uint64_t heap_ancestors(uint32_t lane)
{
    uint32_t l = lane < 2 ? 1u : lane < 6 ? 2u : lane < 14 ? 3u : 4u, o = lane - ((1u << l) - 2u);
    uint64_t anc = 0;
    while (l > 1u) {
        o >>= 1;
        l--;
        anc |= 1ull << (((1u << l) - 2u) + o);
    }
    return lane < 30 ? anc : ~0ull; // (lanes 30..63 look at nothing and are never on the path)
}

__device__ __forceinline__ void heap_extract(HeapMem &h, uint32_t &len, uint32_t &oid, uint32_t &opr, uint32_t lane, uint64_t anc) // :225-267
{
    const uint64_t lk = h.key[len];
    len--;
    if (len == 0) {
        oid = (uint32_t)lk & ((1u << HEAP_ID_BITS) - 1u);
        opr = (uint32_t)(lk >> HEAP_ID_BITS);
        return;
    }
    const uint64_t top = h.key[1];
    oid = (uint32_t)top & ((1u << HEAP_ID_BITS) - 1u);
    opr = (uint32_t)(top >> HEAP_ID_BITS);
    const uint32_t lpr = (uint32_t)(lk >> HEAP_ID_BITS);
    // lane -> slot of the subtree below the hole: level 1 = lanes 0-1, 2 = 2-5, 3 = 6-13, 4 = 14-29
    const uint32_t lvl = lane < 2 ? 1u : lane < 6 ? 2u : lane < 14 ? 3u : 4u;
    const uint32_t off = lane - ((1u << lvl) - 2u);
    uint32_t hole = 1;
    for (;;) {
        const uint32_t node = (hole << lvl) + off;
        const bool ex = lane < 30 && node <= len;
        const uint64_t k = ex ? h.key[node] : 0ull;
        const uint32_t pr = ex ? (uint32_t)(k >> HEAP_ID_BITS) : 0xFFFFFFFFu;
        const uint32_t spr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pr, 0xB1, 0xF, 0xF, false); // sibling: quad_perm [1,0,3,2]
        const bool chosen = (off & 1u) ? (pr < spr) : !(spr < pr);  // right child only if strictly smaller
        const bool cont = ex && !(lpr < pr);                        // the moved element passes this slot
        const uint64_t cm = __ballot(chosen);
        const bool onpath = lane < 30 && chosen && (cm & anc) == anc; // one lane per level
        const uint64_t pm = __ballot(onpath), gm = __ballot(onpath && cont);
        // the element sinks while the path's slots let it pass: d levels (the path's lanes: level 1 in bits 0-1,
        // 2 in 2-5, 3 in 6-13, 4 in 14-29)
        const uint32_t g32 = (uint32_t)gm;
        const uint32_t d = (g32 & 0x3u) ? ((g32 & 0x3Cu) ? ((g32 & 0x3FC0u) ? ((g32 & 0x3FFFC000u) ? 4u : 3u) : 2u) : 1u) : 0u;
        if (onpath && lvl <= d) h.key[node >> 1] = k; // the passed slots rise one level
        if (d) {
            const uint32_t lm = d == 1u ? 0x3u : d == 2u ? 0x3Cu : d == 3u ? 0x3FC0u : 0x3FFFC000u;
            const uint32_t pl = (uint32_t)__builtin_ctz((uint32_t)pm & lm); // the path's lane at level d
            hole = (hole << d) + (pl - ((1u << d) - 2u));
        }
        HEAP_ORDER();
        if (d < 4) break;
    }
    if (lane == 0) h.key[hole] = lk;
    HEAP_ORDER();
}

// One attempt of build_table_from_freqs (:271-298) with scaling = 1 << sh, by all 64 lanes of one wavefront.
// Returns the longest code (uniform); dep[q] = depth of symbol q * 64 + lane.
__device__ int build_attempt(HeapMem &h, uint32_t nsyms, uint32_t sh, uint32_t lane, uint32_t (&dep)[(HUF_SYMS + 63) / 64])
{
    uint32_t nnodes = nsyms + 1, len = 0;
    const uint64_t anc = heap_ancestors(lane);
    for (uint32_t s = 0; s < nsyms; s++) heap_insert(h, len, s + 1, ((h.fr[s] >> sh) + 1u) << 8, lane);
    for (;;) {
        uint32_t a, c, pa, pc;
        heap_extract(h, len, a, pa, lane, anc);
        heap_extract(h, len, c, pc, lane, anc);
        if (nnodes == 2 * nsyms - 1) { // Tree::tie :60-74 -- last tie hangs off the root (id 0)
            if (lane == 0) {
                h.par[a] = 0;
                h.par[c] = 0;
            }
            break;
        }
        const uint32_t parent = nnodes++;
        if (lane == 0) {
            h.par[a] = (int16_t)parent;
            h.par[c] = (int16_t)parent;
        }
        const uint32_t da = pa & 0xFFu, dc = pc & 0xFFu;
        const uint32_t pr = (((pa >> 8) + (pc >> 8)) << 8) | ((da > dc ? da : dc) + 1u); // :147-158
        heap_insert(h, len, parent, pr, lane);
    }
    HEAP_ORDER();
    // leaf depths (:78-102): every lane walks a few leaves up to the root
    int maxlen = 0;
#pragma unroll
    for (int q = 0; q < (HUF_SYMS + 63) / 64; q++) {
        const uint32_t s = q * 64 + lane;
        dep[q] = 0;
        if (s < nsyms) {
            uint32_t x = s + 1, dd = 0;
            do {
                x = (uint32_t)h.par[x];
                dd++;
            } while (x != 0);
            dep[q] = dd;
            maxlen = max(maxlen, (int)dd);
        }
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, s2, 64));
    HEAP_ORDER();
    return maxlen;
}

__device__ __forceinline__ void store_lengths(uint8_t *out, uint32_t nsyms, uint32_t lane, const uint32_t (&dep)[(HUF_SYMS + 63) / 64])
{
#pragma unroll
    for (int q = 0; q < (HUF_SYMS + 63) / 64; q++) {
        const uint32_t s = q * 64 + lane;
        if (s < nsyms) out[s] = (uint8_t)dep[q];
    }
}

// the reference's loop: double `scaling` until no code is longer than 17 bits (:293-296), from scaling 1 << sh0 on
__device__ void build_lengths(HeapMem &h, uint32_t nsyms, uint8_t *out, uint32_t lane, uint32_t sh0 = 0)
{
    uint32_t dep[(HUF_SYMS + 63) / 64];
    for (uint32_t sh = sh0;; sh++) {
        if (build_attempt(h, nsyms, sh, lane, dep) <= HUF_MAXLEN) {
            store_lengths(out, nsyms, lane, dep);
            return;
        }
    }
}

// One wavefront per (table, attempt): the attempts with scaling 1, 2, 4, ... are independent computations from the same
// frequencies, so they run side by side and the first that fits is kept -- exactly the table the reference's sequential
// loop ends with, in the time of ONE build instead of up to five.  Eight attempts a table for two tables, five for three,
// in TWO workgroups a block (the lower and the upper exponents): a build is a chain of dependent LDS round trips whose
// instructions the wavefronts of a SIMD issue in turn -- 12 wavefronts on a CU: 224 us a build, 16: 298 us, 8 or 9: below --
// and a batch has fewer blocks than half the CUs.  Each half leaves what it found (bt.lens2, bt.lfit); huff_header takes
// the lower half's table if it has one.
// (Round 4 ran four attempts a table and noted "should none fit, the loop carries on from scaling 16 -- never seen".  It
// was seen all along: every block of the headline but the first two needs scaling 16 -- a 161-symbol alphabet with byte
// values that occur once in 300,000 symbols --, so the kernel took two builds, 424 us, whatever else was tried on it; its
// time against the number of blocks in the batch showed the step.)
// Should nothing fit in the upper half either, its first wavefront of the table carries on behind the last attempt.
constexpr int HB_WAVES = 10; // (at most 2 tables x 4 or 3 tables x 3 attempts a half)
constexpr uint32_t HB_NONE = 0xFFFFFFFFu;
__global__ void __launch_bounds__(64 * HB_WAVES) huff_build(Batch bt)
{
    const uint32_t half = blockIdx.x, b = blockIdx.y;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ntab = bt.ntab[b], nsyms = bt.nsyms[b];
    const uint32_t t0 = ntab <= 2u ? 4u : 3u, t1 = ntab <= 2u ? 8u : 5u; // (ntab is 2 or 3)
    const uint32_t lo = half ? t0 : 0u, per = half ? t1 - t0 : t0;
    const uint32_t t = w / per, a = lo + w % per;
    __shared__ HeapMem hm[HB_WAVES];
    __shared__ int mx[3][4];
    uint32_t dep[(HUF_SYMS + 63) / 64];
    int maxlen = 0;
    if (t < ntab) {
        const uint32_t *tf = bt.tfreq + ((size_t)b * 3 + t) * HUF_SYMS;
        const uint32_t *F = bt.freqs + (size_t)b * HUF_SYMS;
        for (uint32_t s = lane; s < nsyms; s += 64) hm[w].fr[s] = tf[s] + (t == 0 ? 3u * F[s] : 0u);
        HEAP_ORDER();
        maxlen = build_attempt(hm[w], nsyms, a, lane, dep);
        if (lane == 0) mx[t][a - lo] = maxlen;
    }
    __syncthreads();
    if (t >= ntab) return;
    uint32_t first = HB_NONE; // smallest scaling of my half that fits
    for (int k = (int)per - 1; k >= 0; k--)
        if (mx[t][k] <= HUF_MAXLEN) first = lo + (uint32_t)k;
    const size_t slot = ((size_t)half * bt.B + b) * 3 + t;
    uint8_t *out = bt.lens2 + slot * HUF_SYMS;
    if (first == a) store_lengths(out, nsyms, lane, dep);
    if (first == HB_NONE && half && a == lo) { // (scaling 256 / 32 and beyond: the reference's loop, one attempt after the other)
        build_lengths(hm[w], nsyms, out, lane, t1);
        first = t1;
    }
    if (a == lo && lane == 0) bt.lfit[slot] = first;
}

// ---- header bit string + canonical codes + bit totals ----------------------------------------------------
struct BitW { // MSB-first byte writer into global memory
    uint8_t *p;
    uint32_t acc, nacc, bits;
    __device__ void put(uint32_t v, uint32_t n)
    {
        bits += n;
        while (n) {
            const uint32_t take = min(n, 8u - nacc);
            acc = (acc << take) | ((v >> (n - take)) & ((1u << take) - 1u));
            nacc += take;
            n -= take;
            if (nacc == 8) {
                *p++ = (uint8_t)acc;
                acc = 0;
                nacc = 0;
            }
        }
    }
    __device__ void flush() // pad with zeros to a whole 32-bit word (pack_headers reads words)
    {
        if (nacc) *p++ = (uint8_t)(acc << (8 - nacc));
        const uint32_t bytes = (bits + 7) >> 3;
        for (uint32_t k = bytes; k & 3u; k++) *p++ = 0;
    }
};

__device__ void or_bits(uint32_t *out, uint64_t pos, const uint8_t *src, uint32_t nbits, uint32_t lane); // below

// Three wavefronts share the work: 0 -- the delta-coded tables, one lane per table, each into its own LDS string;
// 1 -- block header, symbol map, counts (lib/lib.rs:24-64, lib/huffman.rs:467-471) and the payload size;
// 2 -- canonical codes of table 0 (:548-561) by ballots.  Then wavefront 0 joins the table strings bit by bit.
constexpr uint32_t TB_BYTES = 1152; // one delta-coded table: <= 5 + 258 * 35 bits
__global__ void __launch_bounds__(192) huff_header(Batch bt)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ntab = bt.ntab[b], nsyms = bt.nsyms[b], m = bt.m[b];
    // the table of every attempt-half that found one (huff_build): the lower exponents' if there is one -- to bt.lens for
    // the kernels behind this one, and into LDS for the serial loops below (one lane per table walks its 258 lengths, one
    // lane writes the symbol map: from global memory every step was a load of its own -- 50 us a launch, whatever the batch)
    __shared__ uint8_t ll[3][HUF_SYMS + 6];
    for (uint32_t t = 0; t < ntab; t++) {
        const uint32_t h = bt.lfit[((size_t)0 * bt.B + b) * 3 + t] != 0xFFFFFFFFu ? 0u : 1u;
        const uint8_t *src = bt.lens2 + (((size_t)h * bt.B + b) * 3 + t) * HUF_SYMS;
        uint8_t *dst = bt.lens + ((size_t)b * 3 + t) * HUF_SYMS;
        for (uint32_t s2 = threadIdx.x; s2 < nsyms; s2 += blockDim.x) {
            const uint8_t v = src[s2];
            dst[s2] = v;
            ll[t][s2] = v;
        }
    }
    __syncthreads();
    uint8_t *hdr = bt.hdr + (size_t)b * HDR_BYTES;
    __shared__ __align__(16) uint8_t tb[3][TB_BYTES];
    __shared__ uint32_t tbits[3], abits, paybits;
    const uint32_t nsel = (m + SEG - 1) / SEG;
    if (wave == 0) {
        // part B will be ORed together below: clear it first
        uint32_t *pb = reinterpret_cast<uint32_t *>(hdr + HDR_A);
        for (uint32_t k = lane; k < (HDR_BYTES - HDR_A) / 4; k += 64) pb[k] = 0;
        if (lane < ntab) { // delta-coded table (:509-545)
            const uint8_t *tl = ll[lane];
            BitW c{tb[lane], 0, 0, 0};
            c.put(tl[0], 5);
            uint32_t acc = tl[0];
            for (uint32_t s2 = 0; s2 < nsyms; s2++) {
                const uint32_t l = tl[s2];
                while (acc < l) {
                    c.put(2, 2);
                    acc++;
                }
                while (acc > l) {
                    c.put(3, 2);
                    acc--;
                }
                c.put(0, 1);
            }
            c.flush();
            tbits[lane] = c.bits;
        }
    } else if (wave == 1) {
        // payload bits = sum F[s] * len0[s]
        const uint32_t *F = bt.freqs + (size_t)b * HUF_SYMS;
        uint32_t pay = 0;
        for (uint32_t s2 = lane; s2 < nsyms; s2 += 64) pay += F[s2] * ll[0][s2];
        pay = wave_reduce_add(pay);
        // the symbol map's sixteen 16-bit sectors, most significant bit first (:39-64): four ballots over has_byte
        const uint8_t *hb = bt.hasbyte + (size_t)b * 256;
        uint64_t hm4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) hm4[q] = __ballot(hb[q * 64 + lane] != 0);
        if (lane == 0) {
            paybits = pay;
            // part A: block header (lib/lib.rs:24-36), symbol map (:39-64), table count, selector count
            BitW a{hdr, 0, 0, 0};
            a.put(0x314159, 24);
            a.put(0x265359, 24);
            const uint32_t crc = bt.pdesc[b].crc;
            a.put(crc >> 16, 16);
            a.put(crc & 0xFFFF, 16);
            a.put(0, 1);
            a.put(bt.ptr[b], 24);
            uint32_t sector_map = 0, sectors[16], ns = 0;
            for (uint32_t x = 0; x < 16; x++) {
                const uint32_t field = (uint32_t)(hm4[x >> 2] >> (16u * (x & 3u))) & 0xFFFFu; // bit y: byte 16 x + y occurs
                const uint32_t sec = __brev(field) >> 16;
                sector_map <<= 1;
                if (sec) {
                    sector_map |= 1;
                    sectors[ns++] = sec;
                }
            }
            a.put(sector_map, 16);
            for (uint32_t k = 0; k < ns; k++) a.put(sectors[k], 16);
            a.put(ntab, 3);  // lib/huffman.rs:467
            a.put(nsel, 15); // :470-471
            a.flush();
            abits = a.bits;
        }
    } else {
        // canonical codes of table 0: within a length, symbols in ascending order (ballot ranks); lengths ascending
        uint32_t *codes = bt.codes + (size_t)b * HUF_SYMS;
        uint32_t l5[(HUF_SYMS + 63) / 64];
        uint32_t minl = 255, maxl = 0;
#pragma unroll
        for (int q = 0; q < (HUF_SYMS + 63) / 64; q++) {
            const uint32_t s2 = q * 64 + lane;
            l5[q] = s2 < nsyms ? ll[0][s2] : 0xFFu;
            if (s2 < nsyms) {
                minl = min(minl, l5[q]);
                maxl = max(maxl, l5[q]);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            minl = min(minl, (uint32_t)__shfl_xor((int)minl, d, 64));
            maxl = max(maxl, (uint32_t)__shfl_xor((int)maxl, d, 64));
        }
        uint32_t word = 0;
        for (uint32_t l = minl; l <= maxl; l++) {
#pragma unroll
            for (int q = 0; q < (HUF_SYMS + 63) / 64; q++) {
                const bool has = l5[q] == l;
                const unsigned long long mk = __ballot(has);
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                if (has) codes[q * 64 + lane] = (l << 24) | (word + below);
                word += (uint32_t)__popcll(mk);
            }
            word <<= 1;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // selectors: nsel single 0 bits (every selector is table 0, :483-505) -- left as zeros in the output
    // part B: the tables one after the other
    uint32_t cbits = 0;
    for (uint32_t t = 0; t < ntab; t++) {
        or_bits(reinterpret_cast<uint32_t *>(hdr + HDR_A), cbits, tb[t], tbits[t], lane);
        cbits += tbits[t];
    }
    if (lane == 0) {
        uint32_t *hb32 = bt.hdrbits + (size_t)b * 4;
        hb32[0] = abits;
        hb32[1] = nsel;
        hb32[2] = cbits;
        hb32[3] = paybits;
        bt.bits[b] = (uint64_t)abits + nsel + cbits + paybits;
    }
}

// Exclusive scan of block bit totals; bitoff[B] = sum.  B <= 1024.
__global__ void __launch_bounds__(1024) block_scan(Batch bt, uint32_t B)
{
    __shared__ uint64_t v[1024];
    const uint32_t t = threadIdx.x;
    v[t] = t < B ? bt.bits[t] : 0;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t x = t >= d ? v[t - d] : 0;
        __syncthreads();
        v[t] += x;
        __syncthreads();
    }
    if (t < B) bt.bitoff[t] = t ? v[t - 1] : 0;
    if (t == 0) bt.bitoff[B] = v[B - 1];
}

// ---- symbol packing -----------------------------------------------------------------------------------------
constexpr int PACK_THREADS = 256;
constexpr int PACK_ITEMS = 16;
static_assert(PACK_THREADS * PACK_ITEMS == PACK_TILE, "pack tile");

// FX: the "fixed" mode -- every 50-symbol segment has its own table (bt.fx_sel), codes of up to 6 tables
template <bool FX>
__global__ void __launch_bounds__(PACK_THREADS) pack_tilebits(Batch bt, uint32_t PT, uint32_t selmax)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t m = bt.m[b];
    if (tile * PACK_TILE >= m) return;
    constexpr uint32_t NT = FX ? FX_TABLES : 1u;
    __shared__ uint32_t cl[NT * HUF_SYMS];
    const uint32_t *codes = FX ? bt.fx_codes + (size_t)b * NT * HUF_SYMS : bt.codes + (size_t)b * HUF_SYMS;
    const uint32_t nsyms = bt.nsyms[b];
    const uint32_t ntab = FX ? bt.ntab[b] : 1u;
    for (uint32_t k = threadIdx.x; k < ntab * HUF_SYMS; k += PACK_THREADS) cl[k] = (k % HUF_SYMS) < nsyms ? codes[k] >> 24 : 0u;
    __syncthreads();
    const uint16_t *s = bt.syms + (size_t)b * (bt.S + 64);
    const uint8_t *sel = bt.fx_sel + (size_t)b * selmax;
    const uint32_t q0 = tile * PACK_TILE + threadIdx.x * PACK_ITEMS;
    uint32_t bits = 0;
    if (q0 < m) {
        const uint4 w0 = *reinterpret_cast<const uint4 *>(s + q0), w1 = *reinterpret_cast<const uint4 *>(s + q0 + 8);
        const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int k = 0; k < PACK_ITEMS; k++)
            if (q0 + k < m) {
                const uint32_t tab = FX ? sel[(q0 + k) / SEG] : 0u;
                bits += cl[tab * HUF_SYMS + ((w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu)];
            }
    }
    __shared__ uint32_t ls[PACK_THREADS / 64 + 2];
    uint32_t tot;
    (void)block_excl_add(bits, ls, &tot);
    if (threadIdx.x == 0) bt.symbits[(size_t)b * PT + tile] = tot;
}

__global__ void __launch_bounds__(1024) pack_tilescan(Batch bt, uint32_t PT)
{
    const uint32_t b = blockIdx.x;
    const uint32_t m = bt.m[b];
    const uint32_t ntile = (m + PACK_TILE - 1) / PACK_TILE; // <= 1024
    uint32_t *sb = bt.symbits + (size_t)b * PT;
    __shared__ uint32_t ls[20];
    const uint32_t v = threadIdx.x < ntile ? sb[threadIdx.x] : 0;
    uint32_t tot;
    const uint32_t ex = block_excl_add(v, ls, &tot);
    if (threadIdx.x < ntile) sb[threadIdx.x] = ex;
}

__device__ __forceinline__ void or_be32(uint32_t *out, uint64_t word_idx, uint32_t v)
{
    if (v) atomicOr(out + word_idx, __builtin_bswap32(v));
}

template <bool FX>
__global__ void __launch_bounds__(PACK_THREADS) pack_symbols(Batch bt, uint32_t PT, uint32_t *out, uint64_t bit_base, uint32_t selmax, const uint32_t *gate)
{
    if (gate && *gate == 0u) return; // (pack_gate: the batch's bits do not fit the output)
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t m = bt.m[b];
    if (tile * PACK_TILE >= m) return;
    constexpr uint32_t NT = FX ? FX_TABLES : 1u;
    __shared__ uint32_t cw[NT * HUF_SYMS];
    __shared__ uint32_t buf[PACK_TILE * HUF_MAXLEN / 32 + 4];
    const uint32_t *codes = FX ? bt.fx_codes + (size_t)b * NT * HUF_SYMS : bt.codes + (size_t)b * HUF_SYMS;
    const uint32_t nsyms = bt.nsyms[b];
    const uint32_t ntab = FX ? bt.ntab[b] : 1u;
    const uint8_t *sel = bt.fx_sel + (size_t)b * selmax;
    for (uint32_t k = threadIdx.x; k < ntab * HUF_SYMS; k += PACK_THREADS) cw[k] = (k % HUF_SYMS) < nsyms ? codes[k] : 0u;
    for (uint32_t k = threadIdx.x; k < PACK_TILE * HUF_MAXLEN / 32 + 4; k += PACK_THREADS) buf[k] = 0;
    __syncthreads();
    const uint16_t *s = bt.syms + (size_t)b * (bt.S + 64);
    const uint32_t q0 = tile * PACK_TILE + threadIdx.x * PACK_ITEMS;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t bits = 0;
    if (q0 < m) {
        const uint4 w0 = *reinterpret_cast<const uint4 *>(s + q0), w1 = *reinterpret_cast<const uint4 *>(s + q0 + 8);
        w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
        w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
#pragma unroll
        for (int k = 0; k < PACK_ITEMS; k++)
            if (q0 + k < m) {
                const uint32_t tab = FX ? sel[(q0 + k) / SEG] : 0u;
                bits += cw[tab * HUF_SYMS + ((w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu)] >> 24;
            }
    }
    __shared__ uint32_t ls[PACK_THREADS / 64 + 2];
    uint32_t tot;
    const uint32_t ex = block_excl_add(bits, ls, &tot);

    const uint32_t *hb = bt.hdrbits + (size_t)b * 4;
    const uint64_t tile_bit = bit_base + bt.bitoff[b] + hb[0] + hb[1] + hb[2] + bt.symbits[(size_t)b * PT + tile];
    const uint64_t word0 = tile_bit >> 5;
    uint32_t lb = (uint32_t)(tile_bit & 31u) + ex; // bit offset inside buf
    if (q0 < m) {
        uint32_t wi = lb >> 5, fill = lb & 31u, nacc = 0;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < PACK_ITEMS; k++) {
            if (q0 + k < m) {
                const uint32_t tab = FX ? sel[(q0 + k) / SEG] : 0u;
                const uint32_t c = cw[tab * HUF_SYMS + ((w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu)];
                const uint32_t L = c >> 24;
                acc = (acc << L) | (c & 0xFFFFFFu);
                nacc += L;
                if (fill + nacc >= 32) {
                    const uint32_t t = 32 - fill;
                    const uint32_t word = (uint32_t)(acc >> (nacc - t)) & (t == 32 ? 0xFFFFFFFFu : ((1u << t) - 1u));
                    atomicOr(&buf[wi], word);
                    nacc -= t;
                    acc &= (1ull << nacc) - 1ull;
                    wi++;
                    fill = 0;
                }
            }
        }
        if (nacc) atomicOr(&buf[wi], (uint32_t)(acc << (32 - fill - nacc)));
    }
    __syncthreads();
    const uint32_t total_bits = (uint32_t)(tile_bit & 31u) + tot;
    const uint32_t nw = (total_bits + 31) >> 5;
    for (uint32_t k = threadIdx.x; k < nw; k += PACK_THREADS) {
        if (k == 0 || k == nw - 1)
            or_be32(out, word0 + k, buf[k]); // words shared with the neighbouring tile / header
        else
            out[word0 + k] = __builtin_bswap32(buf[k]);
    }
}

// Copies `nbits` bits from src (MSB-first bytes, zero padded to words) to bit position `pos` of out.
__device__ void or_bits(uint32_t *out, uint64_t pos, const uint8_t *src, uint32_t nbits, uint32_t lane)
{
    const uint32_t nw = (nbits + 31) >> 5;
    const uint32_t sh = (uint32_t)(pos & 31u);
    const uint64_t w0 = pos >> 5;
    for (uint32_t k = lane; k < nw; k += 64) {
        const uint32_t v = __builtin_bswap32(reinterpret_cast<const uint32_t *>(src)[k]);
        or_be32(out, w0 + k, v >> sh);
        if (sh) or_be32(out, w0 + k + 1, v << (32 - sh));
    }
}

__global__ void __launch_bounds__(64) pack_headers(Batch bt, uint32_t *out, uint64_t bit_base, const uint32_t *gate)
{
    if (gate && *gate == 0u) return;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint32_t *hb = bt.hdrbits + (size_t)b * 4;
    const uint8_t *hdr = bt.hdr + (size_t)b * HDR_BYTES;
    const uint64_t pos = bit_base + bt.bitoff[b];
    or_bits(out, pos, hdr, hb[0], lane);
    or_bits(out, pos + hb[0] + hb[1], hdr + HDR_A, hb[2], lane);
}

__global__ void __launch_bounds__(64) fx_pack_headers(Batch bt, uint32_t *out, uint64_t bit_base, uint32_t selbytes, const uint32_t *gate)
{
    if (gate && *gate == 0u) return;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint32_t *hb = bt.hdrbits + (size_t)b * 4;
    const uint8_t *hdr = bt.fx_hdr + (size_t)b * FX_HDR_BYTES;
    const uint64_t pos = bit_base + bt.bitoff[b];
    or_bits(out, pos, hdr, hb[0], lane);
    or_bits(out, pos + hb[0], bt.fx_selbits + (size_t)b * selbytes, hb[1], lane);
    or_bits(out, pos + hb[0] + hb[1], hdr + FX_HDR_A, hb[2], lane);
}

// ================================================================================================================
// "Fixed" Huffman mode (bzh_set_mode(ctx, BZH_MODE_FIXED); never the default: it is not
// bit-identical to the reference).  What the reference's huffman::encode was meant to do (lib/huffman.rs:313-575):
// 2..6 tables -- chosen from the NUMBER OF SYMBOLS as libbz2 does, where the reference matches on the alphabet size
// and so never gets beyond 3 (:318-326) --, the equal-frequency initial partition (:333-376), and four refinement
// iterations that keep the tables between iterations and restart the frequency lists, where the reference zeroes the
// tables (:402-409) and ends up with one effective table.  Code lengths come from the same exact heap as the default
// mode (any valid length assignment decodes; this one is already here).
// ================================================================================================================
__global__ void __launch_bounds__(64) fx_init(Batch bt)
{
    const uint32_t b = blockIdx.x;
    uint32_t *tf = bt.fx_tfreq + (size_t)b * FX_TABLES * HUF_SYMS;
    for (uint32_t k = threadIdx.x; k < FX_TABLES * HUF_SYMS; k += 64) tf[k] = 0;
    if (threadIdx.x != 0) return;
    const uint32_t nsyms = bt.nsyms[b], m = bt.m[b];
    const uint32_t *F = bt.freqs + (size_t)b * HUF_SYMS;
    const uint32_t ntab = m < 200 ? 2 : m < 600 ? 3 : m < 1200 ? 4 : m < 2400 ? 5 : 6;
    bt.ntab[b] = ntab;
    uint8_t *lens = bt.fx_lens + (size_t)b * FX_TABLES * HUF_SYMS;
    uint32_t remaining = m, left = 0;
    for (uint32_t t = 0; t < ntab; t++) {
        uint32_t right = left, acc = 0;
        bool empty = left >= nsyms; // fewer symbols than tables: the table starts without a range of its own
        if (!empty) {
            const uint32_t target = remaining / (ntab - t);
            for (;;) {
                acc += F[right];
                if (acc >= target || right + 1 == nsyms) break;
                right++;
            }
            if (right > left && t != 0 && t != ntab - 1 && (t & 1u)) {
                acc -= F[right];
                right--;
            }
        }
        for (uint32_t s = 0; s < nsyms; s++) lens[t * HUF_SYMS + s] = (!empty && s >= left && s <= right) ? 0 : 15;
        if (!empty) {
            left = right + 1;
            remaining -= acc;
        }
    }
}

// one refinement iteration, first half: every segment goes to the table that codes it in the fewest bits
// (first minimum), and adds its symbols to that table's frequency list
__global__ void __launch_bounds__(256) fx_segments(Batch bt, uint32_t selmax)
{
    const uint32_t b = blockIdx.y;
    const uint32_t m = bt.m[b];
    const uint32_t nseg = (m + SEG - 1) / SEG;
    const uint32_t seg0 = blockIdx.x * 256;
    if (seg0 >= nseg) return;
    __shared__ uint32_t h[FX_TABLES][HUF_SYMS];
    __shared__ uint64_t cost[HUF_SYMS]; // the six code lengths of a symbol, 10 bits each: one add per symbol sums all tables
    for (int k = threadIdx.x; k < (int)(FX_TABLES * HUF_SYMS); k += 256) (&h[0][0])[k] = 0;
    const uint32_t ntab = bt.ntab[b], nsyms = bt.nsyms[b];
    const uint8_t *lens = bt.fx_lens + (size_t)b * FX_TABLES * HUF_SYMS;
    for (uint32_t sidx = threadIdx.x; sidx < nsyms; sidx += 256) {
        uint64_t c = 0;
        for (uint32_t t = 0; t < ntab; t++) c |= (uint64_t)lens[t * HUF_SYMS + sidx] << (10 * t);
        cost[sidx] = c;
    }
    __syncthreads();
    const uint32_t seg = seg0 + threadIdx.x;
    if (seg < nseg) {
        const uint16_t *s = bt.syms + (size_t)b * (bt.S + 64) + (size_t)seg * SEG;
        const uint32_t len = m - seg * SEG < SEG ? m - seg * SEG : SEG;
        uint32_t w[SEG / 2];
#pragma unroll
        for (int k = 0; k < SEG / 2; k++) w[k] = reinterpret_cast<const uint32_t *>(s)[k];
        uint64_t sum = 0;
#pragma unroll
        for (int k = 0; k < SEG; k++) {
            const uint32_t v = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if ((uint32_t)k < len) sum += cost[v];
        }
        uint32_t best = 0, bc = (uint32_t)sum & 1023u;
        for (uint32_t t = 1; t < ntab; t++) {
            const uint32_t c = (uint32_t)(sum >> (10 * t)) & 1023u;
            if (c < bc) {
                bc = c;
                best = t;
            }
        }
        bt.fx_sel[(size_t)b * selmax + seg] = (uint8_t)best;
#pragma unroll
        for (int k = 0; k < SEG; k++) {
            const uint32_t v = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if ((uint32_t)k < len) atomicAdd(&h[best][v], 1u);
        }
    }
    __syncthreads();
    uint32_t *tf = bt.fx_tfreq + (size_t)b * FX_TABLES * HUF_SYMS;
    for (int k = threadIdx.x; k < (int)(FX_TABLES * HUF_SYMS); k += 256) {
        const uint32_t v = (&h[0][0])[k];
        if (v) atomicAdd(&tf[k], v);
    }
}

// second half: rebuild every table from its frequency list; the lists restart for the next iteration (not after
// the last one: the header needs them for the payload size)
__global__ void __launch_bounds__(64 * FX_TABLES) fx_build(Batch bt, int last)
{
    const uint32_t b = blockIdx.x;
    const uint32_t t = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ntab = bt.ntab[b], nsyms = bt.nsyms[b];
    __shared__ HeapMem hm[FX_TABLES];
    if (t >= ntab) return;
    uint32_t *tf = bt.fx_tfreq + ((size_t)b * FX_TABLES + t) * HUF_SYMS;
    for (uint32_t s = lane; s < nsyms; s += 64) {
        hm[t].fr[s] = tf[s];
        if (!last) tf[s] = 0;
    }
    HEAP_ORDER();
    build_lengths(hm[t], nsyms, bt.fx_lens + ((size_t)b * FX_TABLES + t) * HUF_SYMS, lane);
}

__global__ void __launch_bounds__(64) fx_header(Batch bt, uint32_t selmax, uint32_t selbytes)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint32_t ntab = bt.ntab[b], nsyms = bt.nsyms[b], m = bt.m[b];
    const uint8_t *lens = bt.fx_lens + (size_t)b * FX_TABLES * HUF_SYMS;
    const uint32_t *tf = bt.fx_tfreq + (size_t)b * FX_TABLES * HUF_SYMS; // of the last iteration: symbols per table
    uint8_t *hdr = bt.fx_hdr + (size_t)b * FX_HDR_BYTES;
    uint32_t pay = 0;
    for (uint32_t t = 0; t < ntab; t++)
        for (uint32_t s = lane; s < nsyms; s += 64) pay += tf[t * HUF_SYMS + s] * lens[t * HUF_SYMS + s];
    pay = wave_reduce_add(pay);
    // canonical codes, one table per lane (lib/huffman.rs:548-561)
    if (lane < ntab) {
        const uint8_t *tl = lens + (size_t)lane * HUF_SYMS;
        uint32_t *codes = bt.fx_codes + ((size_t)b * FX_TABLES + lane) * HUF_SYMS;
        uint32_t minl = 255, maxl = 0;
        for (uint32_t s = 0; s < nsyms; s++) {
            minl = min(minl, (uint32_t)tl[s]);
            maxl = max(maxl, (uint32_t)tl[s]);
        }
        uint32_t word = 0;
        for (uint32_t l = minl; l <= maxl; l++) {
            for (uint32_t s = 0; s < nsyms; s++)
                if (tl[s] == l) codes[s] = (l << 24) | word++;
            word <<= 1;
        }
    }
    const uint32_t nsel = (m + SEG - 1) / SEG;
    uint32_t selbits = 0;
    if (lane == 1) { // selectors: move-to-front over the table ids, position j as j ones and a zero (:473-505)
        BitW w{bt.fx_selbits + (size_t)b * selbytes, 0, 0, 0};
        const uint8_t *sel = bt.fx_sel + (size_t)b * selmax;
        uint32_t order = 0x543210u; // recency list, 4 bits per entry
        for (uint32_t k = 0; k < nsel; k++) {
            const uint32_t want = sel[k];
            uint32_t j = 0;
            while (((order >> (4 * j)) & 15u) != want) j++;
            w.put((1u << (j + 1)) - 2u, j + 1);
            const uint32_t lowmask = (1u << (4 * j)) - 1u;
            order = (order & ~((1u << (4 * (j + 1))) - 1u)) | ((order & lowmask) << 4) | want;
        }
        w.flush();
        selbits = w.bits;
    }
    selbits = (uint32_t)__shfl((int)selbits, 1, 64);
    if (lane != 0) return;
    BitW a{hdr, 0, 0, 0};
    a.put(0x314159, 24);
    a.put(0x265359, 24);
    const uint32_t crc = bt.pdesc[b].crc;
    a.put(crc >> 16, 16);
    a.put(crc & 0xFFFF, 16);
    a.put(0, 1);
    a.put(bt.ptr[b], 24);
    {
        const uint8_t *hb = bt.hasbyte + (size_t)b * 256;
        uint32_t sector_map = 0, sectors[16], ns = 0;
        for (uint32_t x = 0; x < 16; x++) {
            uint32_t sec = 0;
            for (uint32_t y = 0; y < 16; y++) sec = (sec << 1) | (hb[(x << 4) | y] ? 1u : 0u);
            sector_map <<= 1;
            if (sec) {
                sector_map |= 1;
                sectors[ns++] = sec;
            }
        }
        a.put(sector_map, 16);
        for (uint32_t k = 0; k < ns; k++) a.put(sectors[k], 16);
    }
    a.put(ntab, 3);
    a.put(nsel, 15);
    a.flush();
    BitW c{hdr + FX_HDR_A, 0, 0, 0};
    for (uint32_t t = 0; t < ntab; t++) {
        const uint8_t *tl = lens + (size_t)t * HUF_SYMS;
        c.put(tl[0], 5);
        uint32_t acc = tl[0];
        for (uint32_t s = 0; s < nsyms; s++) {
            const uint32_t l = tl[s];
            while (acc < l) {
                c.put(2, 2);
                acc++;
            }
            while (acc > l) {
                c.put(3, 2);
                acc--;
            }
            c.put(0, 1);
        }
    }
    c.flush();
    uint32_t *hb32 = bt.hdrbits + (size_t)b * 4;
    hb32[0] = a.bits;
    hb32[1] = selbits;
    hb32[2] = c.bits;
    hb32[3] = pay;
    bt.bits[b] = (uint64_t)a.bits + selbits + c.bits + pay;
}

// ---- host drivers --------------------------------------------------------------------------------------------
// Tables, header strings, per-block bit totals and bitoff[] for blocks 0..B-1 (needs bt.syms, bt.m,
// bt.freqs, bt.nsyms, bt.ptr, bt.hasbyte, bt.desc[].crc).
int huff_prepare(bzh_ctx *ctx, uint32_t B, uint32_t mmax)
{
    KSpan ks(ctx, K_HUFF, 4 * (uint64_t)ctx->k_cur_ntotal, 7); // symbols in twice (segments, bit counts)
    Batch &bt = ctx->bt;
    if (B == 0) return BZH_OK;
    hipStream_t st = ctx->stream;
    uint32_t *ranges = reinterpret_cast<uint32_t *>(bt.tagg); // B*8 words <= B*TPB*2
    const uint32_t nsegmax = (mmax + SEG - 1) / SEG;
    const uint32_t PT = (bt.S + 64 + PACK_TILE - 1) / PACK_TILE;
    const uint32_t ptiles = (mmax + PACK_TILE - 1) / PACK_TILE;
    const uint32_t selmax = (bt.S + 64 + 49) / 50 + 2;
    const uint32_t selbytes = (uint32_t)((((size_t)selmax * 6 + 7) / 8 + 8 + 63) / 64 * 64); // as laid out in api.hip
    if (ctx->mode == BZH_MODE_FIXED) {
        fx_init<<<dim3(B), 64, 0, st>>>(bt);
        for (int it = 0; it < 4; it++) {
            fx_segments<<<dim3((nsegmax + 255) / 256, B), 256, 0, st>>>(bt, selmax);
            fx_build<<<dim3(B), 64 * FX_TABLES, 0, st>>>(bt, it == 3);
        }
        fx_header<<<dim3(B), 64, 0, st>>>(bt, selmax, selbytes);
        block_scan<<<dim3(1), 1024, 0, st>>>(bt, B);
        pack_tilebits<true><<<dim3(ptiles, B), PACK_THREADS, 0, st>>>(bt, PT, selmax);
        pack_tilescan<<<dim3(B), 1024, 0, st>>>(bt, PT);
        HIP_TRY(ctx, hipGetLastError());
        return BZH_OK;
    }
    huff_init<<<dim3(B), 64, 0, st>>>(bt, ranges);
    huff_segments<<<dim3((nsegmax + 255) / 256, B), 256, 0, st>>>(bt, ranges);
    huff_build<<<dim3(2, B), 64 * HB_WAVES, 0, st>>>(bt);
    huff_header<<<dim3(B), 192, 0, st>>>(bt);
    block_scan<<<dim3(1), 1024, 0, st>>>(bt, B);
    pack_tilebits<false><<<dim3(ptiles, B), PACK_THREADS, 0, st>>>(bt, PT, selmax);
    pack_tilescan<<<dim3(B), 1024, 0, st>>>(bt, PT);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

// Writes blocks 0..B-1 at bit_base + bitoff[b] of d_out (zero-initialised, 4-byte aligned).
// What encode_range does on the host between two batches -- read the batch's bit total, check the capacity, zero the words
// the bits will be ORed into, seed the first word -- for a call of one batch, on the device: T = bitoff[B].
// (`tail_bits`: bits the caller writes behind the batch's -- the stream footer, 80, when the whole stream is framed on the
// device: they are zeroed and counted against the capacity here as well)
__global__ void __launch_bounds__(256) pack_gate(uint32_t *out, uint64_t bit_base, const uint64_t *T, uint64_t cap_words, uint32_t seed,
                                                 uint32_t has_seed, uint32_t *gate, uint64_t *hostrec, uint32_t tail_bits)
{
    const uint64_t t = *T, w0 = bit_base / 32, need = (bit_base + t + tail_bits + 31) / 32 + 1;
    const bool ok = need <= cap_words;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *gate = ok ? 1u : 0u;
        hostrec[0] = t;
        hostrec[1] = ok ? 1ull : 0ull;
    }
    if (!ok) return;
    for (uint64_t w = w0 + (uint64_t)blockIdx.x * 256 + threadIdx.x; w < need; w += (uint64_t)gridDim.x * 256)
        out[w] = (w == w0 && has_seed) ? seed : 0u;
}

int huff_pack_gate(bzh_ctx *ctx, uint32_t B, uint8_t *d_out, uint64_t bit_base, uint64_t cap_words, uint32_t seed, bool has_seed,
                   uint64_t *hostrec, uint32_t tail_bits)
{
    Batch &bt = ctx->bt;
    pack_gate<<<dim3(2048), 256, 0, ctx->stream>>>(reinterpret_cast<uint32_t *>(d_out), bit_base, bt.bitoff + B, cap_words, seed, has_seed ? 1u : 0u,
                                                   bt.packgate, hostrec, tail_bits);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

// The stream's frame, written on the device behind the pack of a one-batch stream: "BZh" + level in word 0 (lib/lib.rs:18-22),
// footer magic + stream CRC behind the body (lib/lib.rs:66-70) -- the CRC folded over the batch's block CRCs in block order
// (lib/lib.rs:107-108: crc = block ^ rotl(crc, 1)) where crc_finish left them.  The host used to do this between two waits
// (read the bit total, zero the footer's words, fold, launch): 50 us with the device idle at the end of every step.
__global__ void __launch_bounds__(256) frame_stream(uint32_t *out, int level, const uint64_t *T, const BlockDesc *desc, uint32_t B, const uint32_t *gate)
{
    __shared__ uint32_t crcs[1024]; // (a batch has at most 1,024 blocks: bzh_create) -- loaded by all threads, folded by one from LDS
    if (*gate == 0u) return;
    for (uint32_t b = threadIdx.x; b < B && b < 1024u; b += 256) crcs[b] = desc[b].crc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t crc = 0;
    for (uint32_t b = 0; b < B; b++) crc = (b < 1024u ? crcs[b] : desc[b].crc) ^ ((crc << 1) | (crc >> 31));
    atomicOr(out, __builtin_bswap32(0x425A6800u | (uint32_t)('0' + level)));
    const uint32_t words[3] = {0x17724538u, 0x50900000u | (crc >> 16), crc << 16}; // 80 bits
    const uint64_t pos = 32 + *T;
    const uint32_t sh = (uint32_t)(pos & 31u);
    const uint64_t w0 = pos >> 5;
    for (int k = 0; k < 3; k++) {
        const uint32_t hi = words[k] >> sh, lo = sh ? words[k] << (32 - sh) : 0u;
        if (hi) atomicOr(out + w0 + k, __builtin_bswap32(hi));
        if (lo) atomicOr(out + w0 + k + 1, __builtin_bswap32(lo));
    }
}

int huff_frame_stream(bzh_ctx *ctx, uint32_t B, uint8_t *d_out)
{
    Batch &bt = ctx->bt;
    frame_stream<<<dim3(1), 256, 0, ctx->stream>>>(reinterpret_cast<uint32_t *>(d_out), ctx->level, bt.bitoff + B, bt.pdesc, B, bt.packgate);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

int huff_pack(bzh_ctx *ctx, uint32_t B, uint32_t mmax, uint8_t *d_out, uint64_t bit_base, bool gated)
{
    KSpan ks(ctx, K_PACK, 3 * (uint64_t)ctx->k_cur_ntotal, 2); // symbols in, about a third of a byte out per symbol
    Batch &bt = ctx->bt;
    if (B == 0) return BZH_OK;
    hipStream_t st = ctx->stream;
    const uint32_t PT = (bt.S + 64 + PACK_TILE - 1) / PACK_TILE;
    const uint32_t ptiles = (mmax + PACK_TILE - 1) / PACK_TILE;
    uint32_t *out = reinterpret_cast<uint32_t *>(d_out);
    const uint32_t selmax = (bt.S + 64 + 49) / 50 + 2;
    const uint32_t *gate = gated ? bt.packgate : nullptr;
    if (ctx->mode == BZH_MODE_FIXED) {
        const uint32_t selbytes = (uint32_t)((((size_t)selmax * 6 + 7) / 8 + 8 + 63) / 64 * 64);
        pack_symbols<true><<<dim3(ptiles, B), PACK_THREADS, 0, st>>>(bt, PT, out, bit_base, selmax, gate);
        fx_pack_headers<<<dim3(B), 64, 0, st>>>(bt, out, bit_base, selbytes, gate);
    } else {
        pack_symbols<false><<<dim3(ptiles, B), PACK_THREADS, 0, st>>>(bt, PT, out, bit_base, selmax, gate);
        pack_headers<<<dim3(B), 64, 0, st>>>(bt, out, bit_base, gate);
    }
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
