// mtf.hip -- move-to-front + RLE2 (RUNA/RUNB) + symbol histogram for a batch of blocks.
//
// Replaces mtf::mtf_and_rle (reference lib/mtf.rs:14-121).  Same outputs: output: Vec<u16>
// (here syms[b][0..m)), num_syms = names + 2, freqs[258] with freqs[EOB] = 1.
//
// MTF without a list: the position of byte c in the recency list equals the number of present
// symbols whose "key" is larger than c's key, where key = time of last occurrence, and a symbol
// not seen yet has key -1-c (so unseen symbols keep ascending order behind all seen ones --
// the identity initial list of lib/mtf.rs:39-43).  Absent bytes get INT_MIN and never count.
// Keys at a tile entry are a prefix-max over tiles of per-tile last occurrences, so tiles are
// independent: one wavefront walks 2048 bytes with its 256 keys in 4 VGPRs (symbol c lives in
// lane c&63, register c>>6); one step = scalar readlane + 4 ballots/popcounts.  A byte equal to
// the current front symbol (zero run, the common case after a BWT) costs one scalar compare.
//
// RLE2 is then fully parallel over the position array: every non-zero position emits the
// bijective base-2 digits (lib/mtf.rs:46-65) of the zero run that ends just before it, then its
// own symbol pos+1; offsets by scan; the trailing run and EOB are written by the per-block kernel.
#include "common.h"
#include <cstdlib>
#include <cstring>

constexpr int RLE_THREADS = 256;
constexpr int RLE_ITEMS = 16;
constexpr int RLE_TILE = RLE_THREADS * RLE_ITEMS; // S is a multiple of it: S / RLE_TILE tiles per block

__device__ __forceinline__ uint32_t run_digits(uint32_t z) // symbols emitted for a zero run of length z
{
    return z ? (31u - __clz(z + 1u)) : 0u;
}

// ---- dense names ------------------------------------------------------------------------------------
// names[c] = rank of byte c among the present bytes (lib/mtf.rs:17-24).  MTF positions are the same
// over names as over byte values (the renaming keeps order), and text blocks with <= 128 distinct
// bytes then need 2 key registers instead of 4.  Every workgroup rebuilds the table from has_byte.
__device__ __forceinline__ uint32_t build_names(const uint8_t *hasbyte, uint8_t *names /*LDS[256]*/, uint32_t *ls)
{
    // callable by 64..256 threads: thread t handles bytes t, t+blockDim, ...
    const uint32_t per = 256 / blockDim.x, c0 = threadIdx.x * per;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < per; k++) cnt += hasbyte[c0 + k] ? 1u : 0u;
    uint32_t total;
    uint32_t idx = block_excl_add(cnt, ls, &total);
    for (uint32_t k = 0; k < per; k++) {
        names[c0 + k] = (uint8_t)idx;
        idx += hasbyte[c0 + k] ? 1u : 0u;
    }
    __syncthreads();
    return total;
}

// ---- per-tile last occurrence (indexed by name) -----------------------------------------------------
__global__ void __launch_bounds__(256) mtf_tile_last(Batch bt, int32_t *tlast, uint32_t MT, uint32_t TL)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t n = bt.n[b];
    if (tile * TL >= n) return;
    __shared__ int last[256];
    __shared__ uint8_t names[256];
    __shared__ uint32_t ls[8];
    last[threadIdx.x] = -1;
    (void)build_names(bt.hasbyte + (size_t)b * 256, names, ls);
    const uint8_t *s = bt.bwt + (size_t)b * bt.S;
    static_assert(MTF_TILE % 2048 == 0, "256 threads x 8 bytes per sweep");
#pragma unroll 1
    for (uint32_t sub = 0; sub < TL; sub += 2048) {
    const uint32_t p0 = tile * TL + sub + threadIdx.x * 8;
    if (sub + threadIdx.x * 8 < TL && p0 < n) { // (a tile may be shorter than one sweep of the workgroup: 512 bytes in tiny batches)
        uint2 w = *reinterpret_cast<const uint2 *>(s + p0);
        // only the last byte of a run inside my 8 bytes can be its symbol's last occurrence among them (after a
        // BWT most bytes repeat their neighbour, and equal symbols from one wavefront queue on one LDS word)
        const uint64_t w64 = ((uint64_t)w.y << 32) | w.x;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t p = p0 + k;
            if (p < n) {
                const uint32_t c = (uint32_t)(w64 >> (8 * k)) & 255u;
                const bool more = k < 7 && p + 1 < n && ((uint32_t)(w64 >> (8 * (k + 1))) & 255u) == c;
                if (!more) atomicMax(&last[names[c]], (int)p);
            }
        }
    }
    }
    __syncthreads();
    tlast[((size_t)b * MT + tile) * 256 + threadIdx.x] = last[threadIdx.x];
}

// One workgroup per block: turn per-tile last occurrences into keys at tile entry (exclusive
// running "latest occurrence"), seeded with the initial order; also num_syms.
__global__ void __launch_bounds__(256) mtf_prefix(Batch bt, int32_t *tlast, uint32_t MT, uint32_t TL)
{
    const uint32_t b = blockIdx.x;
    const uint32_t n = bt.n[b];
    const uint32_t ntile = (n + TL - 1) / TL;
    const uint32_t c = threadIdx.x;
    const bool present = bt.hasbyte[(size_t)b * 256 + c] != 0;
    uint32_t cnt = __popcll(__ballot(present));
    __shared__ uint32_t w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    const uint32_t num_names = w[0] + w[1] + w[2] + w[3];
    if (threadIdx.x == 0) bt.nsyms[b] = num_names + 2; // lib/mtf.rs:118
    // thread = name: a never-seen name j sits behind every seen one, in name order (lib/mtf.rs:39-43)
    int run = c < num_names ? -1 - (int)c : INT32_MIN;
    int32_t *t = tlast + (size_t)b * MT * 256 + c;
    uint32_t tile = 0;
    for (; tile + 8 <= ntile; tile += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = t[(size_t)(tile + k) * 256];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            t[(size_t)(tile + k) * 256] = run;
            if (v[k] >= 0) run = v[k];
        }
    }
    for (; tile < ntile; tile++) {
        int v = t[(size_t)tile * 256];
        t[(size_t)tile * 256] = run;
        if (v >= 0) run = v;
    }
}

// ---- the walk: one wavefront per tile ---------------------------------------------------------------
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// NR = key registers in use (names < 64*NR).  All control flow is wave-uniform (scalar).
template <int NR>
__device__ __forceinline__ void walk_tile(const uint8_t *s, uint8_t *o, const int32_t *keys, const uint8_t *names,
                                          int *lkeys, uint32_t base_p, uint32_t tile_len, int lane)
{
    int k0 = keys[lane], k1 = NR > 1 ? keys[64 + lane] : INT32_MIN, k2 = NR > 2 ? keys[128 + lane] : INT32_MIN,
        k3 = NR > 2 ? keys[192 + lane] : INT32_MIN;
    if (NR > 1) { // LDS copy of the keys (one wavefront per workgroup: program order is enough)
        lkeys[lane] = k0;
        lkeys[64 + lane] = k1;
        lkeys[128 + lane] = k2;
        lkeys[192 + lane] = k3;
    }
    int front; // name at the head of the recency list = arg max key
    {
        int best = k0, bsym = lane;
        if (NR > 1 && k1 > best) { best = k1; bsym = 64 + lane; }
        if (NR > 2 && k2 > best) { best = k2; bsym = 128 + lane; }
        if (NR > 2 && k3 > best) { best = k3; bsym = 192 + lane; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            int ob = __shfl_xor(best, d, 64), os = __shfl_xor(bsym, d, 64);
            if (ob > best) {
                best = ob;
                bsym = os;
            }
        }
        front = __builtin_amdgcn_readfirstlane(bsym); // keys are distinct, every lane agrees
    }
    // A byte equal to its predecessor is at the front of the list: position 0, nothing to update.
    // The lanes find those in parallel (change mask); the scalar walk below only visits the others.
    uint32_t carry_last = (uint32_t)front; // name of the byte before the chunk (list head at tile entry)
#pragma unroll 1
    for (uint32_t cbase = 0; cbase < tile_len; cbase += 1024) {
        // 16 bytes per lane (S is padded so the vector load stays inside the arena), renamed once
        const uint4 raw = *reinterpret_cast<const uint4 *>(s + cbase + lane * 16);
        uint32_t in[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t w = in[d];
            in[d] = (uint32_t)names[w & 255u] | ((uint32_t)names[(w >> 8) & 255u] << 8) |
                    ((uint32_t)names[(w >> 16) & 255u] << 16) | ((uint32_t)names[w >> 24] << 24);
        }
        const uint32_t clen = tile_len - cbase < 1024 ? tile_len - cbase : 1024;
        const uint32_t nl = (clen + 15) / 16;
        // change mask: bit k = (byte k of this lane != the byte before it)
        uint32_t pw = (uint32_t)__shfl_up((int)in[3], 1, 64);
        if (lane == 0) pw = carry_last << 24;
        uint32_t chg = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t x = in[d];
            const uint32_t df = x ^ ((x << 8) | (pw >> 24));
            chg |= ((df & 0xFFu) ? 1u : 0u) << (4 * d);
            chg |= ((df & 0xFF00u) ? 2u : 0u) << (4 * d);
            chg |= ((df & 0xFF0000u) ? 4u : 0u) << (4 * d);
            chg |= ((df & 0xFF000000u) ? 8u : 0u) << (4 * d);
            pw = x;
        }
        {
            const uint32_t lo = (uint32_t)lane * 16u;
            const uint32_t nv = clen > lo ? (clen - lo < 16u ? clen - lo : 16u) : 0u;
            chg &= (1u << nv) - 1u; // bytes past the end of the tile are never visited
        }
        { // name of the chunk's last byte, for the next chunk's first comparison
            const uint32_t e = clen - 1;
            const uint32_t w = (uint32_t)rdlane((int)(((e >> 2) & 3u) == 0 ? in[0] : ((e >> 2) & 3u) == 1 ? in[1] : ((e >> 2) & 3u) == 2 ? in[2] : in[3]),
                                                (int)(e >> 4));
            carry_last = (w >> (8 * (e & 3u))) & 255u;
        }
        int o0 = 0, o1 = 0, o2 = 0, o3 = 0;
#pragma unroll 1
        for (uint32_t li = 0; li < nl; li++) {
            const uint32_t m = (uint32_t)rdlane((int)chg, (int)li);
            if (m == 0) continue; // 16 bytes at the front of the list
            const bool me = (uint32_t)lane == li;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (((m >> (4 * d)) & 15u) == 0) continue;
                const uint32_t w = (uint32_t)rdlane((int)in[d], (int)li);
                uint32_t ow = 0;
#pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    if (m & (1u << (4 * d + kb))) {
                        const int c = (int)((w >> (8 * kb)) & 255u);
                        const int p = (int)(base_p + cbase + li * 16 + d * 4 + kb);
                        int prev;
                        if (NR == 1) { // names < 64: the key sits in lane c
                            prev = rdlane(k0, c);
                            k0 = lane == c ? p : k0;
                        } else {
                            // old key through the LDS copy (all lanes read / write the same word), new key
                            // into the one lane whose name index matches: no scalar register selection
                            prev = __builtin_amdgcn_readfirstlane(lkeys[c]);
                            lkeys[c] = p;
                            k0 = lane == c ? p : k0;
                            k1 = lane + 64 == c ? p : k1;
                            if (NR > 2) {
                                k2 = lane + 128 == c ? p : k2;
                                k3 = lane + 192 == c ? p : k3;
                            }
                        }
                        // c's own key is already p (> prev), every other key is unchanged:
                        // position = (keys above prev) - 1 for the symbol itself.
                        uint32_t cnt = (uint32_t)__popcll(__ballot(k0 > prev)) - 1u;
                        if (NR > 1) cnt += (uint32_t)__popcll(__ballot(k1 > prev));
                        if (NR > 2) cnt += (uint32_t)__popcll(__ballot(k2 > prev)) + (uint32_t)__popcll(__ballot(k3 > prev));
                        ow |= cnt << (8 * kb);
                    }
                }
                if (d == 0) o0 = me ? (int)ow : o0;
                else if (d == 1) o1 = me ? (int)ow : o1;
                else if (d == 2) o2 = me ? (int)ow : o2;
                else o3 = me ? (int)ow : o3;
            }
        }
        if ((uint32_t)lane < nl) *reinterpret_cast<uint4 *>(o + cbase + lane * 16) = make_uint4(o0, o1, o2, o3);
    }
}

__global__ void __launch_bounds__(64) mtf_walk(Batch bt, const int32_t *tlast, uint32_t MT, uint32_t TL)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t n = bt.n[b];
    const uint32_t base_p = tile * TL;
    if (base_p >= n) return;
    const int lane = threadIdx.x;
    __shared__ uint8_t names[256];
    __shared__ int lkeys[256];
    __shared__ uint32_t ls[4];
    const uint32_t num_names = build_names(bt.hasbyte + (size_t)b * 256, names, ls);
    const int32_t *keys = tlast + ((size_t)b * MT + tile) * 256;
    const uint8_t *s = bt.bwt + (size_t)b * bt.S + base_p;
    uint8_t *o = bt.mtfpos + (size_t)b * bt.S + base_p;
    const uint32_t remain = n - base_p;
    const uint32_t tile_len = remain < TL ? remain : TL;
    if (num_names <= 64)
        walk_tile<1>(s, o, keys, names, lkeys, base_p, tile_len, lane);
    else if (num_names <= 128)
        walk_tile<2>(s, o, keys, names, lkeys, base_p, tile_len, lane);
    else
        walk_tile<4>(s, o, keys, names, lkeys, base_p, tile_len, lane);
}

// ---- the walk, parallel form: 64 run heads at a time ------------------------------------------------------------
// mtf_walk above visits the changed bytes of a tile one after the other (a scalar chain per byte: with ~160 byte
// values and ~45 % changed bytes it is the second most expensive stage of a step).  Here the lanes of the wavefront
// ARE 64 consecutive run heads (bytes that differ from their predecessor; every other byte has position 0 and leaves
// the list alone).  With E[s] = position of symbol s in the recency list when the chunk begins:
//   * a head whose symbol occurred before in the chunk, last at lane p: its position is the number of distinct symbols
//     between p and itself = the lanes u in (p, l) whose own symbol does not occur in (p, u);
//   * a head whose symbol is new in the chunk: E[c] + the symbols that were behind c in the list and have been seen
//     since the chunk began = the lanes u < l that are first occurrences with E[c_u] > E[c].
//   Both are ONE count over the 64 lanes with per-lane bounds (64 steps of a few vector instructions, no dependent
//   chain); the previous / next occurrence of every lane's symbol come from one match-any (8 ballots).
//   * the list for the next chunk: a seen symbol's position = the distinct symbols whose last occurrence in the chunk
//     is later; an unseen symbol moves back by the seen symbols that were behind it.
// The list at tile entry is the rank of mtf_prefix's keys (lib/mtf.rs:39-43 for the first tile).
__global__ void __launch_bounds__(64) mtf_walk_par(Batch bt, const int32_t *tlast, uint32_t MT, uint32_t TL)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t n = bt.n[b];
    const uint32_t base_p = tile * TL;
    if (base_p >= n) return;
    const int lane = threadIdx.x;
    __shared__ uint8_t names[256];
    __shared__ uint32_t ls[4];
    __shared__ uint16_t Etab[256];         // list position of every name
    __shared__ __attribute__((aligned(16))) uint32_t MF[8]; // per chunk: the list places of its new symbols, one bit each
    // (the tile is walked in halves of 1024 bytes so that a wavefront needs under 5 KB of LDS: the walk is a chain of
    // dependent steps per wavefront, what hides its latency is the number of wavefronts a compute unit can hold)
    __shared__ uint8_t hsym[1024];     // names of the run heads of the half, in order
    __shared__ uint16_t hoff[1024];    // their offsets in the half
    __shared__ __attribute__((aligned(16))) uint8_t opos[1024]; // positions of the half's bytes
    const uint32_t num_names = build_names(bt.hasbyte + (size_t)b * 256, names, ls);
    const int32_t *keys = tlast + ((size_t)b * MT + tile) * 256;
    const uint8_t *s = bt.bwt + (size_t)b * bt.S + base_p;
    uint8_t *o = bt.mtfpos + (size_t)b * bt.S + base_p;
    const uint32_t remain = n - base_p;
    const uint32_t tile_len = remain < TL ? remain : TL;
    // ---- list at tile entry: E[name] = names with a larger key
    int k[4] = {keys[lane], keys[64 + lane], keys[128 + lane], keys[192 + lane]};
    int front;
    const uint32_t regs = (num_names + 63u) / 64u; // registers of 64 names in use (text: two or three of four)
    const uint32_t ebits = num_names > 1u ? 32u - (uint32_t)__clz(num_names - 1u) : 1u; // bits of a list place
    {
        uint32_t e[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if ((uint32_t)r < regs) {
                const int cnt = (int)min(64u, num_names - 64u * r);
                for (int j = 0; j < cnt; j++) {
                    const int kj = rdlane(k[r], j);
#pragma unroll
                    for (int q = 0; q < 4; q++) e[q] += kj > k[q] ? 1u : 0u;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) Etab[q * 64 + lane] = (uint16_t)e[q];
        // the symbol at the head of the list = the byte before the tile (or name 0 at the start of the block)
        int best = k[0], bsym = lane;
#pragma unroll
        for (int q = 1; q < 4; q++)
            if (k[q] > best) {
                best = k[q];
                bsym = q * 64 + lane;
            }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int ob = __shfl_xor(best, d, 64), os = __shfl_xor(bsym, d, 64);
            if (ob > best) {
                best = ob;
                bsym = os;
            }
        }
        front = __builtin_amdgcn_readfirstlane(bsym);
    }
    // ---- half by half: its run heads compacted into LDS (16 bytes a lane), then 64 run heads at a time
    uint32_t carry_last = (uint32_t)front;
#pragma unroll 1
    for (uint32_t cbase = 0; cbase < tile_len; cbase += 1024) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(s + cbase + lane * 16); // (S is padded: the load stays inside the arena)
        uint32_t in[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t w = in[d];
            in[d] = (uint32_t)names[w & 255u] | ((uint32_t)names[(w >> 8) & 255u] << 8) | ((uint32_t)names[(w >> 16) & 255u] << 16) |
                    ((uint32_t)names[w >> 24] << 24);
        }
        const uint32_t clen = tile_len - cbase < 1024 ? tile_len - cbase : 1024;
        uint32_t pw = (uint32_t)__shfl_up((int)in[3], 1, 64);
        if (lane == 0) pw = carry_last << 24;
        uint32_t chg = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t x = in[d];
            const uint32_t df = x ^ ((x << 8) | (pw >> 24));
            chg |= ((df & 0xFFu) ? 1u : 0u) << (4 * d);
            chg |= ((df & 0xFF00u) ? 2u : 0u) << (4 * d);
            chg |= ((df & 0xFF0000u) ? 4u : 0u) << (4 * d);
            chg |= ((df & 0xFF000000u) ? 8u : 0u) << (4 * d);
            pw = x;
        }
        {
            const uint32_t lo = (uint32_t)lane * 16u;
            const uint32_t nv = clen > lo ? (clen - lo < 16u ? clen - lo : 16u) : 0u;
            chg &= (1u << nv) - 1u;
        }
        { // name of the chunk's last byte, for the next half's first comparison
            const uint32_t e = clen - 1;
            const uint32_t w = (uint32_t)rdlane((int)(((e >> 2) & 3u) == 0 ? in[0] : ((e >> 2) & 3u) == 1 ? in[1] : ((e >> 2) & 3u) == 2 ? in[2] : in[3]),
                                                (int)(e >> 4));
            carry_last = (w >> (8 * (e & 3u))) & 255u;
        }
        const uint32_t mine = (uint32_t)__popc(chg);
        const uint32_t inc = wave_incl_add(mine, lane);
        uint32_t at = inc - mine;
#pragma unroll
        for (int kb = 0; kb < 16; kb++) {
            if (chg & (1u << kb)) {
                hsym[at] = (uint8_t)((in[kb >> 2] >> (8 * (kb & 3))) & 255u);
                hoff[at] = (uint16_t)(lane * 16 + kb);
                at++;
            }
        }
        const uint32_t H = (uint32_t)__shfl((int)inc, 63, 64);
        *reinterpret_cast<uint4 *>(&opos[lane * 16]) = make_uint4(0u, 0u, 0u, 0u);
        const bool last_half = cbase + 1024 >= tile_len;
        // 64 run heads at a time (one wavefront: program order is enough between the LDS phases)
#pragma unroll 1
    for (uint32_t hb = 0; hb < H; hb += 64) {
        const uint32_t idx = hb + (uint32_t)lane;
        const bool act = idx < H;
        const uint32_t c = act ? hsym[idx] : 0u;
        // lanes with my symbol (match-any over the 8 bits of the name)
        const unsigned long long am = __ballot(act);
        uint32_t mlo = (uint32_t)am, mhi = (uint32_t)(am >> 32);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const int om = ((int)(c << (31 - bit))) >> 31;
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(om != 0);
            mlo &= ~((uint32_t)bm ^ (uint32_t)om);
            mhi &= ~((uint32_t)(bm >> 32) ^ (uint32_t)om);
        }
        const unsigned long long same = ((unsigned long long)mhi << 32) | mlo;
        const unsigned long long lower = (1ull << lane) - 1ull, upto = (2ull << lane) - 1ull;
        const unsigned long long below = same & lower;
        const int p = below ? 63 - __clzll((long long)below) : -1; // previous occurrence in the chunk
        const bool islast = act && (same & ~upto) == 0ull;
        const int Eown = (int)Etab[c];
        // A = the lanes below me that are the LAST occurrence of their symbol before me: one lane per distinct symbol
        // seen so far.  They are all lower lanes except the predecessors q_v of the lanes v below me: a prefix OR of
        // one bit per lane, six shuffle steps on the two halves of the mask.
        uint32_t xlo = (act && p >= 0 && p < 32) ? 1u << p : 0u, xhi = (act && p >= 32) ? 1u << (p - 32) : 0u;
        // (a prefix OR by data-parallel primitives, then one lane down: exclusive -- the predecessors of the lanes strictly below me)
        xlo = wave_from_below(wave_incl_or(xlo));
        xhi = wave_from_below(wave_incl_or(xhi));
        const unsigned long long A = lower & ~(((unsigned long long)xhi << 32) | xlo);
        int pos;
        unsigned long long firsts = __ballot(act && p < 0); // one lane per distinct symbol of the chunk
        const bool more = hb + 64 < H || !last_half;
        // MF: the places (in the list at chunk entry) of the chunk's new symbols, a bit each -- places are distinct
        if (lane < 8) MF[lane] = 0u;
        if (act && p < 0) atomicOr(&MF[(uint32_t)Eown >> 5], 1u << ((uint32_t)Eown & 31u));
        // a symbol that is new in the chunk: its place in the list at chunk entry + the new symbols BEFORE it in the chunk
        // that were behind it in the list (one scalar step per distinct new symbol)
        // = the lanes u below me among the chunk's first occurrences whose place E_u is larger than mine: a comparator over
        // the ballots of the places' bits, most significant first (gt: lanes already known to be larger, eq: lanes that agree
        // with me so far) -- at most eight steps whatever the number of new symbols, instead of a scalar step per new symbol
        // (round 6; the kernel's time did not move, 522 against 514-527 us: the loop was not what it waits for)
        int cntB;
        {
            const bool isfirst = act && p < 0;
            unsigned long long gt = 0ull, eq = firsts;
#pragma unroll
            for (int bit = 7; bit >= 0; bit--) {
                if ((uint32_t)bit >= ebits) continue; // (uniform: places are below the number of names)
                const bool mine = ((uint32_t)Eown >> bit) & 1u;
                const unsigned long long mb = __ballot(isfirst && mine);
                gt |= mine ? 0ull : (eq & mb);
                eq &= mine ? mb : ~mb;
            }
            cntB = (int)__popcll(gt & lower);
        }
        // the list when the next chunk begins: an unseen symbol moves back by the new symbols that were behind it = the
        // bits of MF above its place (no loop over the new symbols: a shift and two population counts per name)
        int eo[4] = {0, 0, 0, 0}, add[4] = {0, 0, 0, 0};
        if (more) {
            const uint4 ma = *reinterpret_cast<const uint4 *>(&MF[0]), mb = *reinterpret_cast<const uint4 *>(&MF[4]);
            const unsigned long long W0 = ((unsigned long long)ma.y << 32) | ma.x, W1 = ((unsigned long long)ma.w << 32) | ma.z,
                                     W2 = ((unsigned long long)mb.y << 32) | mb.x, W3 = ((unsigned long long)mb.w << 32) | mb.z;
            const int T2 = (int)__popcll(W3), T1 = T2 + (int)__popcll(W2), T0 = T1 + (int)__popcll(W1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if ((uint32_t)q < regs) {
                    eo[q] = (int)Etab[q * 64 + lane];
                    const uint32_t e = (uint32_t)eo[q], sel = e >> 6;
                    const unsigned long long Ws = sel == 0u ? W0 : sel == 1u ? W1 : sel == 2u ? W2 : W3;
                    const int Ts = sel == 0u ? T0 : sel == 1u ? T1 : sel == 2u ? T2 : 0;
                    add[q] = (int)__popcll((Ws >> (e & 63u)) >> 1) + Ts;
                }
            }
        }
        // a symbol seen before in the chunk, last at lane p: the distinct symbols between p and me
        pos = p < 0 ? Eown + cntB : (int)__popcll(A & ~((2ull << p) - 1ull));
        if (act) opos[hoff[idx]] = (uint8_t)pos;
        if (more) { // the list when the next chunk begins
#pragma unroll
            for (int q = 0; q < 4; q++)
                if ((uint32_t)q < regs) Etab[q * 64 + lane] = (uint16_t)(eo[q] + add[q]);
            // (seen symbols are overwritten: position = distinct symbols whose last occurrence comes later)
            const unsigned long long lasts = __ballot(islast);
            if (islast) Etab[c] = (uint16_t)__popcll(lasts & ~upto);
        }
    }
        if (cbase + (uint32_t)lane * 16u < tile_len) *reinterpret_cast<uint4 *>(o + cbase + lane * 16) = *reinterpret_cast<const uint4 *>(&opos[lane * 16]);
    }
}

// ---- RLE2 --------------------------------------------------------------------------------------------
struct RleTile {
    int first_nz; // global position of the first non-zero MTF position in the tile, -1 if none
    int last_nz;  // last one, -1 if none; after rle_block: last non-zero position BEFORE the tile
    uint32_t cnt; // symbols emitted by the tile, not counting the zero-run digits of first_nz
    uint32_t off; // after rle_block: output offset of the tile
};

// Loads 16 positions of the tile into v[], returns the number valid.
__device__ __forceinline__ uint32_t rle_load(const uint8_t *r, uint32_t q0, uint32_t n, uint32_t v[4])
{
    if (q0 >= n) return 0;
    uint4 w = *reinterpret_cast<const uint4 *>(r + q0);
    v[0] = w.x;
    v[1] = w.y;
    v[2] = w.z;
    v[3] = w.w;
    return n - q0 < 16 ? n - q0 : 16;
}

__global__ void __launch_bounds__(RLE_THREADS) rle_tiles(Batch bt, RleTile *rt)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t n = bt.n[b];
    if (tile * RLE_TILE >= n) return;
    const uint8_t *r = bt.mtfpos + (size_t)b * bt.S;
    const uint32_t q0 = tile * RLE_TILE + threadIdx.x * RLE_ITEMS;
    uint32_t v[4];
    const uint32_t valid = rle_load(r, q0, n, v);
    int tfirst = INT32_MAX, tlastnz = -1;
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        if ((uint32_t)k < valid && ((v[k >> 2] >> ((k & 3) * 8)) & 255u)) {
            if (tfirst == INT32_MAX) tfirst = (int)(q0 + k);
            tlastnz = (int)(q0 + k);
        }
    }
    __shared__ int lm[RLE_THREADS / 64];
    __shared__ int ex[RLE_THREADS];
    int inc = block_incl_max(tlastnz, lm);
    ex[threadIdx.x] = inc;
    __syncthreads();
    int cur = threadIdx.x ? ex[threadIdx.x - 1] : -1;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        if ((uint32_t)k < valid && ((v[k >> 2] >> ((k & 3) * 8)) & 255u)) {
            const int p = (int)(q0 + k);
            cnt += 1 + (cur >= 0 ? run_digits((uint32_t)(p - 1 - cur)) : 0u);
            cur = p;
        }
    }
    // reductions: sum cnt, min first, max last
    __shared__ uint32_t ls[RLE_THREADS / 64 + 2];
    __shared__ int lf[RLE_THREADS / 64];
    uint32_t tot;
    (void)block_excl_add(cnt, ls, &tot);
    int f = tfirst;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) f = min(f, __shfl_xor(f, d, 64));
    if ((threadIdx.x & 63) == 0) lf[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RLE_THREADS / 64; w++) f = min(f, lf[w]);
        RleTile t;
        t.first_nz = f == INT32_MAX ? -1 : f;
        t.last_nz = ex[RLE_THREADS - 1];
        t.cnt = tot;
        t.off = 0;
        rt[(size_t)b * (bt.S / RLE_TILE) + tile] = t;
    }
}

// One workgroup per block: carries across tiles, output offsets, trailing run + EOB, m, freqs init.
__global__ void __launch_bounds__(1024) rle_block(Batch bt, RleTile *rt)
{
    const uint32_t b = blockIdx.x;
    const uint32_t n = bt.n[b];
    const uint32_t ntile = (n + RLE_TILE - 1) / RLE_TILE; // <= S / RLE_TILE <= 1024
    const uint32_t t = threadIdx.x;
    uint32_t *freqs = bt.freqs + (size_t)b * 258;
    for (uint32_t k = t; k < 258; k += 1024) freqs[k] = 0;
    RleTile me{-1, -1, 0, 0};
    if (t < ntile) me = rt[(size_t)b * (bt.S / RLE_TILE) + t];
    __shared__ int lm[16];
    __shared__ int incl[1024];
    incl[t] = block_incl_max(me.last_nz, lm);
    __syncthreads();
    const int carry = t ? incl[t - 1] : -1;
    uint32_t cnt = me.cnt;
    if (me.first_nz >= 0) cnt += run_digits((uint32_t)(me.first_nz - 1 - carry));
    __shared__ uint32_t ls[20];
    uint32_t total;
    const uint32_t off = block_excl_add(cnt, ls, &total);
    if (t < ntile) {
        me.last_nz = carry;
        me.off = off;
        rt[(size_t)b * (bt.S / RLE_TILE) + t] = me;
    }
    if (t == 0) {
        const int lastnz = incl[1023];
        const uint32_t z = (uint32_t)((int)n - 1 - lastnz);
        const uint32_t d = run_digits(z);
        uint16_t *out = bt.syms + (size_t)b * (bt.S + 64);
        uint32_t fa = 0, fb = 0;
        for (uint32_t k = 0; k < d; k++) {
            uint32_t bit = ((z + 1) >> k) & 1u;
            out[total + k] = (uint16_t)bit;
            fa += bit ^ 1u;
            fb += bit;
        }
        const uint32_t eob = bt.nsyms[b] - 1; // names + 1 (lib/mtf.rs:30)
        out[total + d] = (uint16_t)eob;
        bt.m[b] = total + d + 1;
        // freqs were zeroed above by other threads of this workgroup
        __threadfence_block();
        atomicAdd(&freqs[0], fa);
        atomicAdd(&freqs[1], fb);
        atomicAdd(&freqs[eob], 1u);
    }
}

__global__ void __launch_bounds__(RLE_THREADS) rle_emit(Batch bt, const RleTile *rt)
{
    const uint32_t b = blockIdx.y, tile = blockIdx.x;
    const uint32_t n = bt.n[b];
    if (tile * RLE_TILE >= n) return;
    const RleTile me = rt[(size_t)b * (bt.S / RLE_TILE) + tile];
    const uint8_t *r = bt.mtfpos + (size_t)b * bt.S;
    const uint32_t q0 = tile * RLE_TILE + threadIdx.x * RLE_ITEMS;
    uint32_t v[4];
    const uint32_t valid = rle_load(r, q0, n, v);
    int tlastnz = -1;
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++)
        if ((uint32_t)k < valid && ((v[k >> 2] >> ((k & 3) * 8)) & 255u)) tlastnz = (int)(q0 + k);

    __shared__ int lm[RLE_THREADS / 64];
    __shared__ int ex[RLE_THREADS];
    __shared__ uint32_t hist[258];
    for (int k = threadIdx.x; k < 258; k += RLE_THREADS) hist[k] = 0;
    int inc = block_incl_max(tlastnz, lm);
    ex[threadIdx.x] = inc;
    __syncthreads();
    const int cur0 = max(me.last_nz, threadIdx.x ? ex[threadIdx.x - 1] : -1);
    int cur = cur0;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        if ((uint32_t)k < valid && ((v[k >> 2] >> ((k & 3) * 8)) & 255u)) {
            const int p = (int)(q0 + k);
            cnt += 1 + run_digits((uint32_t)(p - 1 - cur));
            cur = p;
        }
    }
    __shared__ uint32_t ls[RLE_THREADS / 64 + 2];
    uint32_t tot;
    uint32_t off = me.off + block_excl_add(cnt, ls, &tot);
    uint16_t *out = bt.syms + (size_t)b * (bt.S + 64);
    cur = cur0;
    uint32_t fa = 0, fb = 0;
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        const uint32_t pos = (uint32_t)k < valid ? ((v[k >> 2] >> ((k & 3) * 8)) & 255u) : 0u;
        if (pos) {
            const int p = (int)(q0 + k);
            const uint32_t z = (uint32_t)(p - 1 - cur);
            const uint32_t d = run_digits(z);
            for (uint32_t j = 0; j < d; j++) {
                uint32_t bit = ((z + 1) >> j) & 1u;
                out[off++] = (uint16_t)bit;
                fa += bit ^ 1u;
                fb += bit;
            }
            out[off++] = (uint16_t)(pos + 1); // lib/mtf.rs:91-92
            atomicAdd(&hist[pos + 1], 1u);
            cur = p;
        }
    }
    fa = wave_reduce_add(fa);
    fb = wave_reduce_add(fb);
    if ((threadIdx.x & 63) == 0) {
        if (fa) atomicAdd(&hist[0], fa);
        if (fb) atomicAdd(&hist[1], fb);
    }
    __syncthreads();
    uint32_t *freqs = bt.freqs + (size_t)b * 258;
    for (int k = threadIdx.x; k < 258; k += RLE_THREADS)
        if (hist[k]) atomicAdd(&freqs[k], hist[k]);
}

// MTF + RLE2 for blocks 0..B-1 (bt.bwt / bt.n / bt.hasbyte filled).  tlast and RleTile scratch
// live in the sort lists, which are free once the BWT is emitted.
int mtf_run(bzh_ctx *ctx, uint32_t B, uint32_t nmax, uint64_t ntotal)
{
    Batch &bt = ctx->bt;
    if (B == 0) return BZH_OK;
    hipStream_t st = ctx->stream;
    // A wavefront walks TL bytes from the recency list at their start, which it first builds from the tile's keys (one step per
    // name: a fifth of the walk of a 2,048-byte tile of text).  Tiles of 4,096 bytes halve that share and still leave a large
    // batch three rounds of wavefronts (100 MB: 8.40 -> 8.34 ms); a small batch needs the wavefronts more (28 MB: 3.39 ->
    // 3.49 ms with 4,096) and keeps 2,048.  8,192: no gain anywhere.
    // (Tiles of 512 bytes for batches of one to four blocks -- more wavefronts for a batch that cannot fill the device -- measured
    // SLOWER: config 2 1.062 -> 1.099 ms, one text block 1.258 -> 1.304 ms: the list a tile's walk starts from costs more than
    // the extra wavefronts return.)
    const uint32_t TL = B >= 64u ? 2u * MTF_TILE : MTF_TILE;
    const uint32_t MT = (bt.S + TL - 1) / TL;
    int32_t *tlast = reinterpret_cast<int32_t *>(bt.listA);  // B*MT*256*4 <= B*S*8
    RleTile *rt = reinterpret_cast<RleTile *>(bt.listB);     // B*(S/RLE_TILE)*16 bytes
    const uint32_t mt = (nmax + TL - 1) / TL;
    const uint32_t rtiles = (nmax + RLE_TILE - 1) / RLE_TILE;
    {
        KSpan ks(ctx, K_MTF_LAST, ntotal, 2);
        mtf_tile_last<<<dim3(mt, B), 256, 0, st>>>(bt, tlast, MT, TL);
        mtf_prefix<<<dim3(B), 256, 0, st>>>(bt, tlast, MT, TL);
    }
    {
        KSpan ks(ctx, K_MTF_WALK, 2 * ntotal);
        static const bool serial_walk = []() {
            const char *e = getenv("BZH_MTF");
            return e && !strcmp(e, "serial");
        }();
        if (serial_walk)
            mtf_walk<<<dim3(mt, B), 64, 0, st>>>(bt, tlast, MT, TL);
        else
            mtf_walk_par<<<dim3(mt, B), 64, 0, st>>>(bt, tlast, MT, TL);
    }
    KSpan ks(ctx, K_RLE2, 4 * ntotal, 3); // positions in twice, symbols (<= n, 2 bytes) out
    rle_tiles<<<dim3(rtiles, B), RLE_THREADS, 0, st>>>(bt, rt);
    rle_block<<<dim3(B), 1024, 0, st>>>(bt, rt);
    rle_emit<<<dim3(rtiles, B), RLE_THREADS, 0, st>>>(bt, rt);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
