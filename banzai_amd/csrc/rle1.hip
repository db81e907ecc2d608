// rle1.hip -- RLE1, block splitting and block CRCs on gfx950.
//
// Replaces rle::rle_one (reference lib/rle.rs:102-253) and crc32::checksum (lib/crc32.rs:31-48).
//
// rle_one is a sequential state machine; its result has a closed form (checked against the
// literal restatement in oracle/ on randomized multi-block inputs, tests/test_rle1_model.py):
//   * the RLE1 stream of a block is the canonical one -- every maximal run cut into chunks of
//     255 input bytes, a chunk of l >= 4 bytes becoming 4 literals + count l-4, shorter chunks
//     staying literal (lib/rle.rs:171-234) -- with chunking restarted at the block's first byte;
//   * tokens are taken greedily while they fit in M = 100000*level-1 output bytes (:120-122),
//     except that the 4th literal of a chunk is only taken when its count byte fits too
//     (bound checks at :136-151, :179-182, :193-203); so a block may end at M-1 bytes.
//
// Plan (whole input, parallel, two sweeps of the input, no per-run tables): per 64-byte granule
//   rsg  = start of the run covering the granule's first byte   (prefix max of run starts)
//   nrsg = first run start at or after the granule              (suffix min of run starts)
//   cg   = canonical RLE1 bytes emitted before the granule, relative to its 4096-byte tile,
//          plus tc[tile] = canonical bytes before the tile      (prefix sums of emission counts)
// Splitting (one wavefront, sequential over blocks: block k+1 starts where k ended): 64-ary
// searches in tc / cg find the granule where the budget runs out, the 64 lanes evaluate that
// granule byte-parallel (run starts by max-scan, emission prefix by add-scan) and the closed form
// cuts inside the run.  CRC-32/BZIP2 per block: per-thread table CRC of 32-byte pieces, shifted
// by x^(8*bytes_after) with GF(2) multiplies, XOR-reduced (CRC is linear), init/xorout folded in.
// Emit (per batch): one thread per 16 input bytes, offsets by in-tile scans, output staged in LDS.
#include "common.h"

constexpr int RL_THREADS = 256;
constexpr int RL_ITEMS = 16;
constexpr uint32_t RL_TILE = RL_THREADS * RL_ITEMS; // 4096 input bytes per workgroup
constexpr uint32_t GRAN = 64;                       // bytes per granule = one wavefront of the splitter
constexpr uint32_t GRAN_PER_TILE = RL_TILE / GRAN;  // 64
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

struct BlockAux { // per planned block: what the emit kernel needs about the run the block starts in
    uint64_t Ce;      // canonical offset at the end of that run
    uint32_t A;       // RLE1 bytes of that run's remainder (chunking restarted at in_off)
    uint32_t e_first; // end of that run
    uint32_t open;    // 1 = the cut could move if more input followed (streaming: not final yet)
    uint32_t pad;
};

struct PlanArrays {
    const uint8_t *in;
    uint64_t n;
    uint32_t ntiles;
    uint32_t ngran;
    uint32_t M;
    uint32_t maxblocks;
    uint32_t start;    // input offset the split begins at (a block start; 0 unless a sharded rank continues a chain)
    uint32_t stop;     // the split ends with the first block that starts at or after this offset (a sharded rank's range end)
    uint32_t *lrs;     // [ntiles]   last run start inside the tile (NONE32 if none); then exclusive prefix max
    uint32_t *frs;     // [ntiles+1] first run start inside the tile; then suffix min (frs[ntiles] = n)
    uint32_t *csum;    // [ntiles]   canonical bytes emitted by the tile
    uint64_t *tc;      // [ntiles+1] exclusive scan of csum
    uint32_t *cg;      // [ngran]
    uint32_t *rsg;     // [ngran]
    uint32_t *nrsg;    // [ngran+1]  nrsg[ngran] = n
    BlockDesc *blocks; // [maxblocks]
    BlockAux *aux;     // [maxblocks]
    uint32_t *nblocks; // [1]
};

__device__ __forceinline__ uint32_t canon_len(uint32_t L) // RLE1 bytes of a run of L equal bytes
{
    const uint32_t q = L / 255u, r = L - q * 255u;
    return 5u * q + (r < 4u ? r : 5u);
}

__device__ __forceinline__ uint32_t emitted_before(uint32_t d) // canonical bytes of the first d bytes of a run
{
    const uint32_t c = d / 255u, k = d - c * 255u;
    return 5u * c + (k < 4u ? k : 4u);
}

// 16 input bytes at p0 (p0 and `in` 16-byte aligned); the ragged tail is read byte by byte so
// nothing past in[n-1] is touched.
__device__ __forceinline__ void load16(const uint8_t *in, uint64_t n, uint64_t p0, uint32_t v[4])
{
    if (p0 + 16 <= n) {
        const uint4 w = *reinterpret_cast<const uint4 *>(in + p0);
        v[0] = w.x;
        v[1] = w.y;
        v[2] = w.z;
        v[3] = w.w;
    } else {
        v[0] = v[1] = v[2] = v[3] = 0;
        for (uint32_t k = 0; p0 + k < n; k++) v[k >> 2] |= (uint32_t)in[p0 + k] << ((k & 3) * 8);
    }
}

// Returns a 16-bit mask of run starts among the 16 bytes at p0 (needs the byte before p0).
__device__ __forceinline__ uint32_t start_mask(const uint8_t *in, uint64_t n, uint64_t p0, uint32_t &valid,
                                               uint32_t v[4])
{
    valid = 0;
    if (p0 >= n) return 0;
    load16(in, n, p0, v);
    valid = n - p0 < 16 ? (uint32_t)(n - p0) : 16u;
    uint32_t prev = p0 ? in[p0 - 1] : 0x100u;
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t c = (v[k >> 2] >> ((k & 3) * 8)) & 255u;
        if ((uint32_t)k < valid && c != prev) mask |= 1u << k;
        prev = c;
    }
    return mask;
}

// Is one of the thread's `valid` bytes at p0 the 4th (or later) of equal bytes in a row?  `mask` = its run starts.
// Byte q is such a byte iff bytes q-2, q-1 and q all do NOT start a run (the two flags before the thread's 16 come
// from the 3 bytes before p0; p0 is a multiple of 16).  Text has few runs of four: a wavefront without one emits
// every byte as one literal -- no run offsets modulo 255, no count bytes.
__device__ __forceinline__ bool has_run_of_four(const uint8_t *in, uint64_t p0, uint32_t valid, uint32_t mask)
{
    uint32_t before = 0; // bit 1: byte p0-1 does not start a run, bit 0: byte p0-2 does not
    if (valid && p0 >= 4) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(in + p0 - 4);
        const uint32_t b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
        before = (b3 == b2 ? 2u : 0u) | (b2 == b1 ? 1u : 0u);
    }
    const uint32_t vm = valid >= 16 ? 0xFFFFu : ((1u << valid) - 1u);
    const uint32_t ext = ((~mask & vm) << 2) | before;
    return (ext & (ext >> 1) & (ext >> 2)) != 0u;
}

// ---- plan sweep 1: first / last run start of every tile ------------------------------------------------
__global__ void __launch_bounds__(RL_THREADS) plan_starts(PlanArrays pa)
{
    const uint32_t tile = blockIdx.x;
    const uint64_t p0 = (uint64_t)tile * RL_TILE + threadIdx.x * RL_ITEMS;
    uint32_t valid, v[4];
    const uint32_t mask = start_mask(pa.in, pa.n, p0, valid, v);
    int last = mask ? (int)((uint32_t)p0 + 31u - (uint32_t)__clz((int)mask)) : -1;
    uint32_t first = mask ? (uint32_t)p0 + (uint32_t)__ffs((int)mask) - 1u : NONE32;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        last = max(last, __shfl_xor(last, d, 64));
        first = min(first, (uint32_t)__shfl_xor((int)first, d, 64));
    }
    __shared__ int sl[RL_THREADS / 64];
    __shared__ uint32_t sf[RL_THREADS / 64];
    if ((threadIdx.x & 63) == 0) {
        sl[threadIdx.x >> 6] = last;
        sf[threadIdx.x >> 6] = first;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RL_THREADS / 64; w++) {
            last = max(last, sl[w]);
            first = min(first, sf[w]);
        }
        pa.lrs[tile] = last < 0 ? NONE32 : (uint32_t)last;
        pa.frs[tile] = first;
    }
}

// Single workgroup: lrs -> exclusive prefix max (run start covering each tile's first byte, unless
// that byte starts a run itself); frs -> suffix min (first run start at or after each tile).
// 16 tiles per thread per sweep.  Both are unsigned max-scans: a position p is encoded as p+1 for
// the prefix max and as n-p for the suffix min (taken as a prefix max walking backwards); 0 = none.
constexpr int PC_PER = 16;
constexpr uint32_t PC_CHUNK = 1024 * PC_PER;

__device__ __forceinline__ uint32_t block_incl_umax(uint32_t v, uint32_t *lds) // lds: blockDim/64 words
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v = max(v, x);
    }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    uint32_t c = 0;
    for (int w = 0; w < wave; w++) c = max(c, lds[w]);
    v = max(v, c);
    __syncthreads();
    return v;
}

// 16 consecutive table words of a thread as four 16-byte accesses (the tables are 256-byte aligned and a thread's first index
// is a multiple of 16): a scalar loop made every wavefront load touch 64 different lines for 4 bytes each
__device__ __forceinline__ void load16(const uint32_t *p, uint32_t e0, uint32_t NT, uint32_t fill, uint32_t (&v)[PC_PER])
{
    if (e0 + PC_PER <= NT) {
#pragma unroll
        for (int q = 0; q < PC_PER / 4; q++) {
            const uint4 w = *reinterpret_cast<const uint4 *>(p + e0 + 4 * q);
            v[4 * q] = w.x;
            v[4 * q + 1] = w.y;
            v[4 * q + 2] = w.z;
            v[4 * q + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PC_PER; k++) v[k] = e0 + k < NT ? p[e0 + k] : fill;
    }
}
__device__ __forceinline__ void store16(uint32_t *p, uint32_t e0, uint32_t NT, const uint32_t (&v)[PC_PER])
{
    if (e0 + PC_PER <= NT) {
#pragma unroll
        for (int q = 0; q < PC_PER / 4; q++) *reinterpret_cast<uint4 *>(p + e0 + 4 * q) = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < PC_PER; k++)
            if (e0 + k < NT) p[e0 + k] = v[k];
    }
}

__global__ void __launch_bounds__(1024) plan_carries(PlanArrays pa)
{
    __shared__ uint32_t lm[16];
    __shared__ uint32_t tailv[1024];
    const uint32_t t = threadIdx.x, NT = pa.ntiles, n32 = (uint32_t)pa.n;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < NT; base += PC_CHUNK) {
        const uint32_t e0 = base + t * PC_PER;
        uint32_t v[PC_PER], o[PC_PER], run = 0;
        load16(pa.lrs, e0, NT, NONE32, v);
#pragma unroll
        for (int k = 0; k < PC_PER; k++) {
            v[k] = v[k] == NONE32 ? 0u : v[k] + 1u;
            run = max(run, v[k]);
        }
        tailv[t] = block_incl_umax(run, lm);
        __syncthreads();
        uint32_t ex = max(carry, t ? tailv[t - 1] : 0u);
#pragma unroll
        for (int k = 0; k < PC_PER; k++) {
            o[k] = ex ? ex - 1u : NONE32; // exclusive
            ex = max(ex, v[k]);
        }
        store16(pa.lrs, e0, NT, o);
        carry = max(carry, tailv[1023]);
        __syncthreads();
    }
    if (t == 0) pa.frs[NT] = n32;
    carry = 0;
    for (uint32_t base = 0; base < NT; base += PC_CHUNK) { // base counts tiles from the END
        const uint32_t r0 = base + t * PC_PER;
        uint32_t v[PC_PER], run = 0;
#pragma unroll
        for (int k = 0; k < PC_PER; k++) {
            const uint32_t r = r0 + k;
            const uint32_t raw = r < NT ? pa.frs[NT - 1 - r] : NONE32;
            v[k] = raw == NONE32 ? 0u : n32 - raw; // >= 1, larger = earlier position
            run = max(run, v[k]);
        }
        tailv[t] = block_incl_umax(run, lm);
        __syncthreads();
        uint32_t ex = max(carry, t ? tailv[t - 1] : 0u);
#pragma unroll
        for (int k = 0; k < PC_PER; k++) {
            const uint32_t r = r0 + k;
            ex = max(ex, v[k]); // inclusive: first run start at or after the tile's first byte
            if (r < NT) pa.frs[NT - 1 - r] = ex ? n32 - ex : n32;
        }
        carry = max(carry, tailv[1023]);
        __syncthreads();
    }
}

// ---- plan sweep 2: per-granule tables and per-tile canonical byte counts -----------------------------------
__global__ void __launch_bounds__(RL_THREADS) plan_granules(PlanArrays pa)
{
    const uint32_t tile = blockIdx.x;
    const uint64_t tile0 = (uint64_t)tile * RL_TILE;
    const uint64_t p0 = tile0 + threadIdx.x * RL_ITEMS;
    uint32_t valid, v[4];
    const uint32_t mask = start_mask(pa.in, pa.n, p0, valid, v);
    uint32_t nextb = 0x100u;
    if (valid == RL_ITEMS && p0 + RL_ITEMS < pa.n) nextb = pa.in[p0 + RL_ITEMS];
    __shared__ uint32_t ls[RL_THREADS / 64 + 2];
    __shared__ int lm[RL_THREADS / 64];
    __shared__ int exm[RL_THREADS];
    __shared__ uint32_t sfirst[RL_THREADS];
    // run start covering each thread's first byte
    int tl = mask ? (int)(threadIdx.x * RL_ITEMS) + (31 - __clz((int)mask)) : -1;
    exm[threadIdx.x] = block_incl_max(tl, lm);
    sfirst[threadIdx.x] = mask ? (uint32_t)p0 + (uint32_t)__ffs((int)mask) - 1u : NONE32;
    __syncthreads();
    const uint32_t rst = pa.lrs[tile]; // exclusive prefix max (NONE32 only for tile 0, whose byte 0 starts a run)
    const int carry = threadIdx.x ? exm[threadIdx.x - 1] : -1;
    const uint32_t rs_in = carry >= 0 ? (uint32_t)tile0 + (uint32_t)carry : rst;
    uint32_t cur_rs = rs_in, tsum = valid; // (a wavefront without a run of four: one literal per byte)
    if (__ballot(has_run_of_four(pa.in, p0, valid, mask)) != 0ull) {
        tsum = 0;
#pragma unroll
        for (int k = 0; k < RL_ITEMS; k++) {
            if ((uint32_t)k < valid) {
                const uint32_t p = (uint32_t)p0 + k;
                if (mask & (1u << k)) cur_rs = p;
                const uint32_t byte = (v[k >> 2] >> ((k & 3) * 8)) & 255u;
                const uint32_t nb = ((uint32_t)k + 1 < valid) ? ((v[((k + 1) & 15) >> 2] >> (((k + 1) & 3) * 8)) & 255u) : nextb;
                const uint32_t kk = (p - cur_rs) % 255u;
                tsum += (kk < 4u ? 1u : 0u) + ((kk >= 3u && (kk == 254u || nb != byte)) ? 1u : 0u);
            }
        }
    }
    uint32_t tot;
    const uint32_t ps = block_excl_add(tsum, ls, &tot);
    // suffix min of first run starts over the threads of the tile: shuffles inside the wavefront,
    // then the later wavefronts' minima through LDS (one barrier)
    {
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t sf = sfirst[threadIdx.x];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = (uint32_t)__shfl_down((int)sf, d, 64);
            if (ln + d < 64) sf = min(sf, x);
        }
        __shared__ uint32_t wmin[RL_THREADS / 64];
        if (ln == 0) wmin[wv] = sf;
        __syncthreads();
        for (int w = wv + 1; w < RL_THREADS / 64; w++) sf = min(sf, wmin[w]);
        sfirst[threadIdx.x] = sf;
    }
    if ((threadIdx.x & 3u) == 0) {
        const size_t g = (size_t)tile * GRAN_PER_TILE + (threadIdx.x >> 2);
        if (p0 < pa.n) {
            pa.cg[g] = ps;
            pa.rsg[g] = (mask & 1u) ? (uint32_t)p0 : rs_in;
            pa.nrsg[g] = min(sfirst[threadIdx.x], pa.frs[tile + 1]);
        } else {
            pa.nrsg[g] = (uint32_t)pa.n; // granule past the end of the input
        }
    }
    if (threadIdx.x == 0) {
        pa.csum[tile] = tot;
        if (tile == 0) pa.nrsg[pa.ngran] = (uint32_t)pa.n;
    }
}

// Single workgroup: tc = exclusive scan of csum (64-bit), tc[ntiles] = total.  16 tiles per thread:
// a chunk of 16384 tiles emits < 2^27 bytes, so the in-chunk scan is 32-bit, the carry 64-bit.
__global__ void __launch_bounds__(1024) plan_tc(PlanArrays pa)
{
    __shared__ uint32_t ls[20];
    uint64_t carry = 0;
    const uint32_t t = threadIdx.x, NT = pa.ntiles;
    for (uint32_t base = 0; base < NT; base += PC_CHUNK) {
        const uint32_t e0 = base + t * PC_PER;
        uint32_t v[PC_PER], sum = 0;
        load16(pa.csum, e0, NT, 0u, v);
#pragma unroll
        for (int k = 0; k < PC_PER; k++) sum += v[k];
        uint32_t tot;
        uint32_t ex = block_excl_add(sum, ls, &tot);
#pragma unroll
        for (int k = 0; k < PC_PER; k++) {
            if (e0 + k < NT) pa.tc[e0 + k] = carry + ex;
            ex += v[k];
        }
        carry += tot;
    }
    if (t == 0) pa.tc[NT] = carry;
}

// ---- block splitting: one wavefront, sequential over blocks ----------------------------------------------
__device__ __forceinline__ void cut_in_run(uint32_t Lr, uint32_t R, uint32_t &k, uint32_t &t)
{
    const uint32_t nf = Lr / 255u;
    k = R / 5u < nf ? R / 5u : nf;
    const uint32_t rho = R - 5u * k;
    const uint32_t ell = k < nf ? 255u : Lr - 255u * nf;
    t = ell >= 4u ? (rho < 3u ? rho : 3u) : rho;
}

struct Gran { // one 64-byte granule seen by the wavefront, lane l = byte g*64+l
    uint32_t pos;  // byte position of this lane
    bool valid;    // pos < n
    bool start;    // this byte starts a run
    uint64_t cpos; // canonical RLE1 bytes emitted before this byte
};

__device__ __forceinline__ Gran gran_eval(const PlanArrays &pa, uint32_t g, uint32_t lane)
{
    Gran r;
    const uint32_t n = (uint32_t)pa.n;
    r.pos = g * GRAN + lane;
    r.valid = r.pos < n;
    const uint32_t byte = r.valid ? pa.in[r.pos] : 0x100u;
    uint32_t prevb = (uint32_t)__shfl_up((int)byte, 1, 64);
    if (lane == 0) prevb = r.pos ? pa.in[r.pos - 1] : 0x200u;
    uint32_t nextb = (uint32_t)__shfl_down((int)byte, 1, 64);
    if (lane == 63) nextb = r.pos + 1 < n ? pa.in[r.pos + 1] : 0x100u;
    r.start = r.valid && byte != prevb;
    const bool is_last = nextb != byte;
    const int own = r.start ? (int)lane : -1;
    const int ms = wave_incl_max(own, (int)lane);
    const uint32_t rs = ms >= 0 ? g * GRAN + (uint32_t)ms : pa.rsg[g];
    const uint32_t kk = (r.pos - rs) % 255u;
    const uint32_t em = r.valid ? ((kk < 4u ? 1u : 0u) + ((kk >= 3u && (kk == 254u || is_last)) ? 1u : 0u)) : 0u;
    const uint32_t inc = wave_incl_add(em, (int)lane);
    r.cpos = pa.tc[g / GRAN_PER_TILE] + pa.cg[g] + (inc - em);
    return r;
}

// First run start at a position > s (n if there is none), from an already evaluated granule g = s / GRAN.
__device__ __forceinline__ uint32_t next_start_in(const PlanArrays &pa, const Gran &gr, uint32_t g, uint32_t s)
{
    const unsigned long long m = __ballot(gr.valid && gr.pos > s && gr.start);
    if (m) return g * GRAN + (uint32_t)__ffsll((long long)m) - 1u;
    return pa.nrsg[g + 1 <= pa.ngran ? g + 1 : pa.ngran];
}

// First run start at a position > s (n if there is none).
__device__ __forceinline__ uint32_t next_start_after(const PlanArrays &pa, uint32_t s, uint32_t lane)
{
    const uint32_t g = s / GRAN;
    const uint32_t pos = g * GRAN + lane;
    const uint32_t n = (uint32_t)pa.n;
    const uint32_t byte = pos < n ? pa.in[pos] : 0x100u;
    uint32_t prevb = (uint32_t)__shfl_up((int)byte, 1, 64);
    if (lane == 0) prevb = pos ? pa.in[pos - 1] : 0x200u;
    const unsigned long long m = __ballot(pos < n && pos > s && byte != prevb);
    if (m) return g * GRAN + (uint32_t)__ffsll((long long)m) - 1u;
    return pa.nrsg[g + 1 <= pa.ngran ? g + 1 : pa.ngran];
}

// The split is one chain of dependent table reads per block, by one wavefront that nothing else on the device keeps
// company: the tables were written by other XCDs, so every read goes to memory.  A second wavefront runs ahead and
// touches what the split is about to read -- for 64 blocks at a time, a lane each: a block ends where the canonical
// RLE1 offset has grown by M (less the few bytes a cut gives away, which add up: the window reaches further back for
// later blocks), so the tile comes from tc by a few secant steps and the granule from the offset inside the tile.
// Nothing depends on what it reads; a wrong guess (long runs) just warms the wrong lines.
__device__ void plan_prefetch(const PlanArrays &pa, uint32_t lane)
{
    const uint32_t NT = pa.ntiles, NG = pa.ngran, n = (uint32_t)pa.n;
    if (NT == 0 || NG == 0 || pa.start >= n) return;
    const uint64_t total = pa.tc[NT];
    const uint32_t t0 = min(pa.start / RL_TILE, NT - 1u);
    const uint64_t C0 = pa.tc[t0] + pa.cg[min(pa.start / GRAN, NG - 1u)];
    uint32_t sink = 0;
    for (uint32_t k0 = 0; k0 < pa.maxblocks && k0 < 8192u; k0 += 64) {
        const uint32_t k = k0 + lane;
        const uint64_t target = C0 + (uint64_t)(k + 1u) * pa.M;
        if (__ballot(target < total) == 0ull) break;
        if (target >= total) continue;
        const long long slope = (long long)(total / NT) + 1; // canonical bytes per tile, on average
        long long g = (long long)t0 + (long long)(target - C0) / slope;
#pragma unroll 1
        for (int it = 0; it < 5; it++) { // (bounded: a guess that does not settle warms the wrong lines, nothing else)
            g = g < 0 ? 0 : (g > (long long)NT - 1 ? (long long)NT - 1 : g);
            const long long diff = (long long)target - (long long)pa.tc[g];
            g += diff >= 0 ? diff / slope : -((-diff + slope - 1) / slope);
        }
        g = g < 0 ? 0 : (g > (long long)NT - 1 ? (long long)NT - 1 : g);
        for (int it = 0; it < 3 && g > 0 && pa.tc[g] > target; it++) g--;
        const uint64_t base = pa.tc[g];
        sink ^= (uint32_t)pa.tc[g + 1];
        const long long gc = g * (long long)GRAN_PER_TILE + (long long)((target > base ? target - base : 0) / GRAN);
        long long lo = gc - 4 - (long long)(5u * (k + 1u) / GRAN + 2u), hi = gc + 3;
        lo = lo < 0 ? 0 : lo;
        hi = hi > (long long)NG - 1 ? (long long)NG - 1 : hi;
        for (long long q = lo; q <= hi; q += 16) sink ^= pa.cg[q] ^ pa.rsg[q] ^ pa.nrsg[q];
        sink ^= pa.cg[hi] ^ pa.rsg[hi] ^ pa.nrsg[hi];
        for (uint64_t a = (uint64_t)lo * GRAN; a < ((uint64_t)hi + 1u) * GRAN && a < n; a += 64) sink ^= pa.in[a];
    }
    if (sink == 0x9E3779B9u && lane == 77u) pa.nblocks[1] = sink; // (never: keeps the loads alive)
}

// ---- one cut, by one wavefront --------------------------------------------------------------------------------------
struct CutState { // the granule the wavefront evaluated last (usually needed again by the next step)
    Gran ev{};
    uint32_t idx = 0xFFFFFFFFu;
};
struct CutEnd { // where a block ends whose largest canonical offset that still fits is `lim` (lim < total)
    uint32_t end;  // input position of the cut
    uint32_t R;    // budget that was left for the run the cut falls into
    uint32_t kt;   // canonical bytes taken from that run (5 per full chunk + literals)
    uint32_t open; // the cut could move if more input followed
};

// The second half of a cut: last tile / granule / run start whose canonical offset is <= lim, then the closed form inside
// that run.  `lo_tile`: a tile known to start at or before lim.  Also what tells where a LATER block would start if
// every block before it ended exactly on its budget (plan_split's speculation).
__device__ __forceinline__ CutEnd end_from_lim(const PlanArrays &pa, uint32_t lane, uint64_t lim, uint32_t lo_tile, CutState &cs)
{
    const uint32_t N = (uint32_t)pa.n, NT = pa.ntiles;
    // last tile with tc <= lim: probe 64 tiles around the literal-text guess, else 64-ary search
    uint32_t lo = lo_tile, hi = NT - 1; // tc[lo] <= lim
    {
        long long guess = (long long)lo + (long long)((lim - pa.tc[lo]) / RL_TILE) - 40;
        if (guess < (long long)lo) guess = lo;
        if (guess > (long long)hi) guess = hi;
        const uint32_t w0 = (uint32_t)guess; // lo <= w0 <= hi
        const uint32_t x = w0 + lane;
        const bool ok = x <= hi && pa.tc[x] <= lim;
        const unsigned long long m = __ballot(ok); // tc is nondecreasing: a prefix of the window
        if (m & 1ull) {
            const uint32_t c = (uint32_t)__popcll(m);
            lo = w0 + c - 1;
            if (c < 64) hi = lo; // tc[lo+1] > lim or lo is the last tile
        } else {
            hi = w0 - 1; // tc[w0] > lim, and w0 > lo because tc[lo] <= lim
        }
    }
    while (lo < hi) {
        const uint32_t span = hi - lo;
        const uint32_t step = (span + 63u) / 64u;
        const uint64_t x = (uint64_t)lo + (uint64_t)(lane + 1) * step;
        const bool ok = x <= hi && pa.tc[x] <= lim;
        const uint32_t c = (uint32_t)__popcll(__ballot(ok));
        lo += c * step;
        const uint64_t nh = (uint64_t)lo + step - 1;
        if (nh < hi) hi = (uint32_t)nh;
    }
    const uint32_t tl = lo;
    // last granule of that tile whose start offset is <= lim
    const uint32_t gb = tl * GRAN_PER_TILE;
    const uint32_t gidx = gb + lane;
    const bool gok = (uint64_t)gidx * GRAN < pa.n && pa.tc[tl] + pa.cg[gidx] <= lim;
    const uint32_t gx = gb + (uint32_t)__popcll(__ballot(gok)) - 1u; // lane 0 always ok
    if (gx != cs.idx) {
        cs.ev = gran_eval(pa, gx, lane);
        cs.idx = gx;
    }
    const Gran &gr = cs.ev;
    // last run start in the granule that still fits, else the run covering the granule
    const unsigned long long fit = __ballot(gr.start && gr.cpos <= lim);
    uint32_t x;
    uint64_t Cx;
    if (fit) {
        const int l = 63 - __clzll((long long)fit);
        x = gx * GRAN + (uint32_t)l;
        Cx = __shfl(gr.cpos, l, 64);
    } else {
        x = pa.rsg[gx];
        Cx = __shfl(gr.cpos, 0, 64) - emitted_before(gx * GRAN - x);
    }
    CutEnd r;
    r.R = (uint32_t)(lim - Cx); // budget left for the run starting at x
    const uint32_t xe = (x / GRAN == cs.idx) ? next_start_in(pa, cs.ev, cs.idx, x) : next_start_after(pa, x, lane);
    const uint32_t Lx = xe - x;
    r.open = xe >= N && (uint64_t)Lx < 255ull * (r.R / 5u + 2u);
    uint32_t k, t;
    cut_in_run(Lx, r.R, k, t);
    r.end = x + 255u * k + t;
    r.kt = 5u * k + t;
    return r;
}

struct CutRec {
    BlockDesc d;
    BlockAux ax;
};

// One block cut from input position s (lib/rle.rs:102-253 in closed form, see the head of this file): descriptor + what
// the emit kernel needs.  Returns the bytes consumed; *origin = lim - M of this cut, the canonical offset the block's
// budget is counted from (valid unless the budget ran out inside the block's first run: *origin = ~0).
__device__ __forceinline__ uint32_t cut_block(const PlanArrays &pa, uint32_t lane, uint32_t s, uint64_t total, CutState &cs, CutRec &rec,
                                              uint64_t *origin)
{
    const uint32_t N = (uint32_t)pa.n, M = pa.M;
    const uint32_t e_first = (s / GRAN == cs.idx) ? next_start_in(pa, cs.ev, cs.idx, s) : next_start_after(pa, s, lane);
    const uint32_t Lr = e_first - s;
    const uint32_t A = canon_len(Lr);
    uint32_t consumed, out, open = 0;
    uint64_t Ce = 0;
    *origin = ~0ull;
    // A cut inside the input's last run is final only if that run is known to hold at least one
    // more full 255-byte chunk than the budget can take (then its true length cannot matter).
    if (A > M) { // the budget runs out inside the run the block starts in
        uint32_t k, t;
        cut_in_run(Lr, M, k, t);
        consumed = 255u * k + t;
        out = 5u * k + t;
        open = e_first >= N && (uint64_t)Lr < 255ull * (M / 5u + 2u);
    } else {
        if (e_first >= N) {
            Ce = total;
        } else {
            if (e_first / GRAN != cs.idx) {
                cs.ev = gran_eval(pa, e_first / GRAN, lane);
                cs.idx = e_first / GRAN;
            }
            Ce = __shfl(cs.ev.cpos, (int)(e_first % GRAN), 64);
        }
        const uint64_t lim = (uint64_t)(M - A) + Ce; // largest canonical offset that still fits
        *origin = lim - M;
        if (total <= lim) { // everything to the end of the input fits
            consumed = N - s;
            out = A + (uint32_t)(total - Ce);
            open = 1;
        } else {
            const CutEnd ce = end_from_lim(pa, lane, lim, e_first / RL_TILE, cs); // (tc[e_first's tile] <= Ce <= lim)
            consumed = ce.end - s;
            out = M - ce.R + ce.kt;
            open = ce.open;
        }
    }
    rec.d.in_off = s;
    rec.d.in_len = consumed;
    rec.d.rle_len = out;
    rec.d.crc = 0;
    rec.ax.Ce = Ce;
    rec.ax.A = A;
    rec.ax.e_first = e_first;
    rec.ax.open = open;
    rec.ax.pad = 0;
    return consumed;
}

// ---- the split: SP_W wavefronts cut SP_K blocks each, from starts that are right unless a cut gave bytes away ---------
// Block k+1 starts where block k ended, and a cut is a chain of dependent table reads (2 us): 112 blocks were 232 us of
// every step on ONE wavefront -- and in a sharded run every later rank waits for the splits of all ranks before it.
// But a block that ends exactly on its budget (every block of text: the budget is only missed next to a run of four or
// more) leaves the next block's budget counted from a canonical offset M further on, so the start of block j is known
// without cutting blocks 0 .. j-1: it is where a block would end whose budget reaches origin + j M.  Wavefront w takes
// that start for j = w SP_K (one end_from_lim), cuts its SP_K blocks, and the workgroup keeps the blocks of the
// wavefronts whose start turned out to be the end of the wavefront before (wavefront 0 starts from the known start):
// everything else is cut again in the next window, from the last good end -- a window per irregular cut instead of a
// chain link per block.  The result is the sequential split's, block for block: a wavefront's cuts depend on its start
// only, and a start is used only if it is the true one.
constexpr uint32_t SP_W = 15, SP_K = 8;

// One wavefront that runs ahead of the split and touches the table lines it is about to read -- a launch of its own on the
// context's second stream, beside plan_split (as a sixteenth wavefront of plan_split it left the workgroup before the
// barriers of the others: a barrier not reached by every thread).
__global__ void __launch_bounds__(64) plan_prefetch_kernel(PlanArrays pa) { plan_prefetch(pa, threadIdx.x); }

__global__ void __launch_bounds__(64 * SP_W) plan_split(PlanArrays pa)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t N = (uint32_t)pa.n, M = pa.M, NT = pa.ntiles;
    const uint64_t total = pa.tc[NT];
    __shared__ CutRec rec[SP_W][SP_K];
    __shared__ uint32_t w_start[SP_W], w_end[SP_W], w_cnt[SP_W], w_fin[SP_W]; // per wavefront: first start, last end, blocks cut, "the split ends here"
    __shared__ unsigned long long s_origin;
    __shared__ uint32_t s_s, s_nb, s_done;
    if (threadIdx.x == 0) {
        s_s = pa.start;
        s_nb = 0;
        s_done = 0;
    }
    CutState cs;
    __syncthreads();
    for (;;) {
        const uint32_t s0 = s_s, nb0 = s_nb;
        if (s_done || s0 >= N || nb0 >= pa.maxblocks) break;
        // -- wavefront 0 cuts the window's first block: its origin is what the other wavefronts' starts are counted from
        uint32_t cnt = 0, sw = N, fin = 0;
        uint64_t origin = ~0ull;
        if (wave == 0) {
            const uint32_t used = cut_block(pa, lane, s0, total, cs, rec[0][0], &origin);
            cnt = 1;
            fin = s0 >= pa.stop ? 1u : 0u; // the block that belongs to the next range is on record: its start is all that was wanted
            sw = s0 + used;
            if (lane == 0) {
                s_origin = origin;
                w_start[0] = s0;
            }
        }
        __syncthreads();
        origin = s_origin;
        if (wave > 0) {
            // my first block would be block wave * SP_K of the window: it starts where block wave * SP_K - 1 ends
            sw = N;
            const uint64_t lim = origin + (uint64_t)wave * SP_K * M;
            if (origin != ~0ull && lim < total) {
                const uint32_t lo = s0 / RL_TILE; // (tc of the window's first tile <= canonical offset at s0 <= lim)
                sw = end_from_lim(pa, lane, lim, lo, cs).end;
            }
            if (lane == 0) w_start[wave] = sw;
        }
        // -- every wavefront cuts its blocks (wavefront 0 has one already)
        while (cnt < SP_K && sw < N && !fin) {
            uint64_t o2;
            const uint32_t used = cut_block(pa, lane, sw, total, cs, rec[wave][cnt], &o2);
            fin = sw >= pa.stop ? 1u : 0u;
            cnt++;
            sw += used;
        }
        if (lane == 0) {
            w_end[wave] = sw;
            w_cnt[wave] = cnt;
            w_fin[wave] = fin;
        }
        __syncthreads();
        // -- keep the wavefronts whose start is the end of the one before; the next window begins behind them
        uint32_t V = 1, off = 0, mine = 0xFFFFFFFFu, end_all = w_end[0], fin_all = w_fin[0];
        if (wave == 0) mine = 0;
        off = w_cnt[0];
        for (uint32_t w = 1; w < SP_W; w++) {
            if (fin_all || end_all >= N || w_cnt[w] == 0u || w_start[w] != end_all) break;
            if (w == wave) mine = off;
            off += w_cnt[w];
            end_all = w_end[w];
            fin_all = w_fin[w];
            V++;
        }
        const bool overflow = nb0 + off > pa.maxblocks;
        if (mine != 0xFFFFFFFFu && !overflow) {
            // 56-byte records, written by the lanes: 6 + 8 dwords (BlockDesc is 24 bytes, BlockAux 32)
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t *sd = reinterpret_cast<const uint32_t *>(&rec[wave][j].d);
                const uint32_t *sa = reinterpret_cast<const uint32_t *>(&rec[wave][j].ax);
                uint32_t *dd = reinterpret_cast<uint32_t *>(pa.blocks + nb0 + mine + j);
                uint32_t *da = reinterpret_cast<uint32_t *>(pa.aux + nb0 + mine + j);
                if (lane < sizeof(BlockDesc) / 4) dd[lane] = sd[lane];
                if (lane < sizeof(BlockAux) / 4) da[lane] = sa[lane];
            }
        }
        __syncthreads(); // (the records and the per-wavefront words are reused by the next window)
        if (threadIdx.x == 0) {
            s_nb = nb0 + off;
            s_s = overflow ? 0u : (fin_all ? N : end_all);
            s_done = (overflow || fin_all || end_all >= N) ? 1u : 0u;
            if (overflow) s_nb = 0xFFFFFFFFu;
        }
        __syncthreads();
        (void)V;
    }
    if (threadIdx.x == 0) *pa.nblocks = (s_nb == 0xFFFFFFFFu || (!s_done && s_s < N)) ? 0xFFFFFFFFu : s_nb; // overflow marker
}

// ---- CRC-32/BZIP2 ----------------------------------------------------------------------------------------
constexpr uint32_t CRC_POLY = 0x04C11DB7u;
constexpr uint32_t CRC_PIECE = 32;                         // bytes per thread
constexpr uint32_t CRC_TILE = RL_THREADS * CRC_PIECE;      // 8192 bytes per workgroup
constexpr uint32_t CRC_WG_TILES = 8;                       // adjacent tiles a workgroup folds (at least)

struct CrcTables {
    uint32_t pow2[40];   // x^(2^k) mod P, k = 0..39 (bit exponents)
    uint32_t shift[256]; // x^(8*32*i) mod P: moves a 32-byte piece i pieces to the left
};

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) // a*b mod P, bit 31 = x^31
{
    uint32_t r = 0;
#pragma unroll 8
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r >> 31) ? CRC_POLY : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}

// x^e mod P for a wave-uniform exponent e (< 2^40): lanes take one bit each, product by butterfly.
__device__ __forceinline__ uint32_t gf_pow_x(const CrcTables &ct, uint64_t e, uint32_t lane)
{
    uint32_t f = 1u; // polynomial 1
    if (lane < 40 && ((e >> lane) & 1ull)) f = ct.pow2[lane];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) f = gf_mul(f, __shfl_xor(f, d, 64));
    return f;
}

// The CRC of the block's tiles [tA, tB) (adjacent 8 KiB pieces of its input), shifted to the block's end and XORed
// into *accb.  The tiles fold into one value (acc * x^(8 * tile bytes) + tile), so a workgroup pays one power of x and
// one atomic, not one per tile.  Slicing by four: tab[k][v] = CRC of byte v followed by k zero bytes, so four bytes
// cost one dependent step (four independent LDS lookups) instead of four.
__device__ __forceinline__ void crc_range(const uint8_t *in, const BlockDesc &d, uint32_t *accb, const CrcTables &ct, uint64_t tA,
                                          uint64_t tB)
{
    __shared__ uint32_t tab[4][256];
    __shared__ uint32_t wred[RL_THREADS / 64];
    {
        uint32_t c = threadIdx.x << 24;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c << 1) ^ ((c >> 31) ? CRC_POLY : 0u);
        tab[0][threadIdx.x] = c;
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 1; k < 4; k++) {
        const uint32_t c = tab[k - 1][threadIdx.x];
        tab[k][threadIdx.x] = (c << 8) ^ tab[0][c >> 24];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc_c = 0; // (wavefront 0)
    uint64_t range_end = 0;
    for (uint64_t t = tA; t < tB; t++) {
        const uint64_t t0 = t * CRC_TILE;
        const uint32_t tile_len = d.in_len - t0 < CRC_TILE ? (uint32_t)(d.in_len - t0) : CRC_TILE;
        range_end = t0 + tile_len;
        // pieces: a ragged first piece of r bytes (if any), then full 32-byte pieces, so that the bytes
        // after every piece are a multiple of 32
        const uint32_t r = tile_len % CRC_PIECE, np = tile_len / CRC_PIECE + (r ? 1u : 0u);
        uint32_t crc = 0;
        if (threadIdx.x < np) {
            uint32_t off, len;
            if (r) {
                off = threadIdx.x ? r + (threadIdx.x - 1) * CRC_PIECE : 0;
                len = threadIdx.x ? CRC_PIECE : r;
            } else {
                off = threadIdx.x * CRC_PIECE;
                len = CRC_PIECE;
            }
            const uint8_t *p = in + d.in_off + t0 + off;
            if (len == CRC_PIECE) { // (gfx950 global loads need no alignment: two 16-byte loads of the piece)
                uint32_t w[8];
                __builtin_memcpy(w, p, 32);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t x = crc ^ __builtin_bswap32(w[q]); // the stream's first byte is the polynomial's highest term
                    crc = tab[3][x >> 24] ^ tab[2][(x >> 16) & 255u] ^ tab[1][(x >> 8) & 255u] ^ tab[0][x & 255u];
                }
            } else {
                for (uint32_t k = 0; k < len; k++) crc = (crc << 8) ^ tab[0][(crc >> 24) ^ p[k]];
            }
            crc = gf_mul(crc, ct.shift[np - 1 - threadIdx.x]);
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) crc ^= __shfl_xor(crc, s, 64);
        if (lane == 0) wred[threadIdx.x >> 6] = crc;
        __syncthreads();
        if (threadIdx.x < 64) {
            uint32_t c = 0;
            for (int w = 0; w < RL_THREADS / 64; w++) c ^= wred[w];
            // x^(8 * tile_len): a constant for full tiles (8 * 8192 = 2^16), a power for the block's last, ragged one
            const uint32_t xp = tile_len == CRC_TILE ? ct.pow2[16] : gf_pow_x(ct, 8ull * tile_len, lane);
            acc_c = gf_mul(acc_c, xp) ^ c;
        }
        __syncthreads(); // wred is reused by the next tile
    }
    if (threadIdx.x < 64 && tB > tA) {
        const uint32_t pw = gf_pow_x(ct, 8ull * (d.in_len - range_end), lane);
        if (lane == 0) atomicXor(accb, gf_mul(acc_c, pw));
    }
}

// Workgroup x of block y takes the x-th range of the block's tiles: CRC_WG_TILES of them (64 KiB) for an ordinary block,
// more when the launch has fewer workgroups than that needs (the caller knows the longest block).
__global__ void __launch_bounds__(RL_THREADS) crc_tiles(const uint8_t *in, const BlockDesc *blocks, uint32_t *acc,
                                                         const CrcTables *ctp, const uint32_t *nbp)
{
    const uint32_t b = blockIdx.y;
    if (nbp && b >= *nbp) return;
    const BlockDesc d = blocks[b];
    const uint64_t ntile = (d.in_len + CRC_TILE - 1) / CRC_TILE;
    const uint64_t per = max((uint64_t)CRC_WG_TILES, (ntile + gridDim.x - 1) / gridDim.x);
    const uint64_t tA = (uint64_t)blockIdx.x * per;
    if (tA >= ntile) return;
    crc_range(in, d, &acc[b], *ctp, tA, tA + per < ntile ? tA + per : ntile);
}

// The same over a FLAT grid, for the plan, which queues the CRCs before the host knows the blocks: workgroup g takes
// the g-th range of CRC_WG_TILES tiles counted over all blocks (a block that is one enormous run of input holds
// thousands of tiles, an ordinary one 110: a grid of blocks x ranges would be nearly all empty workgroups).  The block
// of a range: 64 blocks at a time, a lane each, by a wave scan of their range counts.
__global__ void __launch_bounds__(RL_THREADS) crc_tiles_flat(const uint8_t *in, const BlockDesc *blocks, uint32_t *acc,
                                                              const CrcTables *ctp, const uint32_t *nbp)
{
    const uint32_t nb = *nbp;
    if (nb == 0xFFFFFFFFu) return; // (the split overflowed: the host reports it)
    const uint32_t lane = threadIdx.x & 63;
    __shared__ uint32_t s_b, s_first;
    if (threadIdx.x < 64) {
        uint32_t before = 0, found = 0xFFFFFFFFu, first = 0;
        for (uint32_t b0 = 0; b0 < nb && found == 0xFFFFFFFFu; b0 += 64) {
            const uint32_t b = b0 + lane;
            const uint64_t ntile = b < nb ? (blocks[b].in_len + CRC_TILE - 1) / CRC_TILE : 0;
            const uint32_t cnt = (uint32_t)((ntile + CRC_WG_TILES - 1) / CRC_WG_TILES);
            const uint32_t inc = wave_incl_add(cnt, (int)lane);
            const unsigned long long m = __ballot(before + inc > blockIdx.x); // (counts are non-negative: a suffix of the lanes)
            if (m) {
                const int l = __ffsll((long long)m) - 1;
                found = b0 + (uint32_t)l;
                first = before + (uint32_t)__shfl((int)(inc - cnt), l, 64);
            }
            before += (uint32_t)__shfl((int)inc, 63, 64);
        }
        if (lane == 0) {
            s_b = found;
            s_first = first;
        }
    }
    __syncthreads();
    const uint32_t b = s_b;
    if (b == 0xFFFFFFFFu) return; // beyond the last range
    const BlockDesc d = blocks[b];
    const uint64_t ntile = (d.in_len + CRC_TILE - 1) / CRC_TILE;
    const uint64_t tA = (uint64_t)(blockIdx.x - s_first) * CRC_WG_TILES;
    crc_range(in, d, &acc[b], *ctp, tA, tA + CRC_WG_TILES < ntile ? tA + CRC_WG_TILES : ntile);
}

__global__ void __launch_bounds__(64) crc_finish(BlockDesc *blocks, const uint32_t *acc, uint32_t nb,
                                                  const CrcTables *ctp, const uint32_t *nbp)
{
    const uint32_t b = blockIdx.x;
    if (b >= (nbp ? *nbp : nb)) return;
    const CrcTables &ct = *ctp;
    const uint32_t pw = gf_pow_x(ct, 8ull * blocks[b].in_len, threadIdx.x);
    if (threadIdx.x == 0) blocks[b].crc = acc[b] ^ gf_mul(0xFFFFFFFFu, pw) ^ 0xFFFFFFFFu;
}

// Device copy of the GF(2) tables, created on first use (one context = one GPU).
static int crc_tables(bzh_ctx *ctx, const CrcTables **out);

static CrcTables make_crc_tables()
{
    CrcTables ct;
    auto mul = [](uint32_t a, uint32_t b) {
        uint32_t r = 0;
        for (int i = 31; i >= 0; i--) {
            r = (r << 1) ^ ((r >> 31) ? CRC_POLY : 0u);
            if ((b >> i) & 1u) r ^= a;
        }
        return r;
    };
    uint32_t p = 2u; // x
    for (int k = 0; k < 40; k++) {
        ct.pow2[k] = p;
        p = mul(p, p);
    }
    const uint32_t step = ct.pow2[8]; // x^256 = one 32-byte piece
    uint32_t sft = 1u;
    for (int i = 0; i < 256; i++) {
        ct.shift[i] = sft;
        sft = mul(sft, step);
    }
    return ct;
}

static int crc_tables(bzh_ctx *ctx, const CrcTables **out)
{
    if (!ctx->d_crctab) {
        const CrcTables ct = make_crc_tables();
        if (hipMalloc(&ctx->d_crctab, sizeof ct) != hipSuccess) return BZH_E_NOMEM;
        HIP_TRY(ctx, hipMemcpy(ctx->d_crctab, &ct, sizeof ct, hipMemcpyHostToDevice));
    }
    *out = (const CrcTables *)ctx->d_crctab;
    return BZH_OK;
}

// ---- emit: RLE1 bytes of a batch of blocks -----------------------------------------------------------------
struct EmitArgs {
    const uint8_t *in;
    uint64_t n;
    const uint32_t *rsg;
    const uint64_t *tc;
    const BlockDesc *blocks; // plan blocks, already offset to the batch's first block
    const BlockAux *aux;
};

// One workgroup = one 4096-byte input tile of one block.  Nothing is looked up per run: the
// canonical output offset and the covering run's start at the tile's first byte come from the plan
// tables (one lookup per tile); inside the tile, run starts are a max-scan of the start flags and
// output offsets an add-scan of the per-byte emission counts (0, 1 or 2).  The block's first run
// (chunking restarted at in_off) uses the closed form instead.  The tile's output is one contiguous
// byte range (<= 5/4 of the tile), staged in LDS and written as aligned 32-bit words (ragged edges
// byte by byte: neighbouring tiles own the other bytes of those words).
__global__ void __launch_bounds__(RL_THREADS) rle1_emit_kernel(EmitArgs ea, Batch bt)
{
    const uint32_t b = blockIdx.y;
    const BlockDesc d = ea.blocks[b];
    const uint64_t in_end = d.in_off + d.in_len;
    const uint32_t tile = (uint32_t)(d.in_off / RL_TILE) + blockIdx.x;
    const uint64_t tile0 = (uint64_t)tile * RL_TILE;
    const uint64_t p0 = tile0 + threadIdx.x * RL_ITEMS;
    if (tile0 >= in_end) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bt.n[b] = d.rle_len;
        bt.desc[b] = d;
    }
    __shared__ uint32_t ls[RL_THREADS / 64 + 2];
    __shared__ int lm[RL_THREADS / 64];
    __shared__ int exm[RL_THREADS];
    __shared__ uint8_t stage[RL_TILE + RL_TILE / 4 + 16];
    __shared__ uint32_t oend_s, rst_s;
    __shared__ uint64_t tc_s;

    uint32_t valid, v[4];
    const uint32_t mask = start_mask(ea.in, ea.n, p0, valid, v);
    // byte after this thread's 16 (to tell whether its last byte ends a run)
    uint32_t nextb = 0x100u;
    if (valid == RL_ITEMS && p0 + RL_ITEMS < ea.n) nextb = ea.in[p0 + RL_ITEMS];

    // block constants and tile carries (one thread, a handful of table reads)
    const BlockAux ax = ea.aux[b];
    const uint32_t A = ax.A;   // RLE1 bytes of the run the block starts in, after the restart
    const uint64_t Ce = ax.Ce; // canonical offset at that run's end
    if (threadIdx.x == 0) {
        rst_s = ea.rsg[(size_t)tile * GRAN_PER_TILE]; // start of the run covering the tile's first byte
        tc_s = ea.tc[tile];                           // canonical RLE1 bytes before the tile
        oend_s = 0;
    }
    // run start of every byte: max-scan of start positions, carried in from the covering run
    int tl = mask ? (int)(threadIdx.x * RL_ITEMS) + (31 - __clz((int)mask)) : -1;
    exm[threadIdx.x] = block_incl_max(tl, lm);
    __syncthreads();
    const uint32_t rst = rst_s;
    const int carry = threadIdx.x ? exm[threadIdx.x - 1] : -1; // tile-relative start of the open run, -1 = rst
    // canonical emission count per byte (2 bits) + "ends its run" (1 bit), packed 4 bits per byte
    const uint32_t rs_in = carry >= 0 ? (uint32_t)tile0 + (uint32_t)carry : rst;
    uint32_t cur_rs = rs_in;
    unsigned long long em = 0;
    uint32_t tsum = 0;
    const bool wave_slow = __ballot(has_run_of_four(ea.in, p0, valid, mask)) != 0ull; // (else: literals only)
    if (wave_slow) {
#pragma unroll
        for (int k = 0; k < RL_ITEMS; k++) {
            if ((uint32_t)k < valid) {
                const uint32_t p = (uint32_t)p0 + k;
                if (mask & (1u << k)) cur_rs = p;
                const uint32_t byte = (v[k >> 2] >> ((k & 3) * 8)) & 255u;
                const uint32_t nb = ((uint32_t)k + 1 < valid) ? ((v[((k + 1) & 15) >> 2] >> (((k + 1) & 3) * 8)) & 255u) : nextb;
                const bool is_last = nb != byte;
                const uint32_t kk = (p - cur_rs) % 255u;
                const uint32_t can = (kk < 4u ? 1u : 0u) + ((kk >= 3u && (kk == 254u || is_last)) ? 1u : 0u);
                em |= (unsigned long long)(can | (is_last ? 4u : 0u)) << (4 * k);
                tsum += can;
            }
        }
    } else {
        tsum = valid;
    }
    uint32_t tot;
    uint32_t ps = block_excl_add(tsum, ls, &tot); // canonical bytes of the tile before this thread
    const uint64_t tc = tc_s;
    // offset of the first byte this tile can emit (uniform): its first in-block position
    const uint32_t in_off32 = (uint32_t)d.in_off;
    uint32_t obase;
    if (tile0 <= d.in_off)
        obase = 0; // the block starts in this tile
    else if (rst <= in_off32)
        obase = emitted_before((uint32_t)tile0 - in_off32); // still inside the block's first run
    else
        obase = A + (uint32_t)(tc - Ce);

    uint32_t lastend = 0;
    cur_rs = rs_in;
    if (!wave_slow) { // literals only: a byte's offset is its canonical offset, or its distance from the block's start inside the block's first run
#pragma unroll
        for (int k = 0; k < RL_ITEMS; k++) {
            if ((uint32_t)k < valid) {
                const uint32_t p = (uint32_t)p0 + k;
                if (mask & (1u << k)) cur_rs = p;
                if (p0 + k >= d.in_off && p0 + k < in_end) {
                    const uint32_t off = cur_rs <= in_off32 ? p - in_off32 : A + (uint32_t)(tc + ps - Ce);
                    stage[off - obase] = (uint8_t)((v[k >> 2] >> ((k & 3) * 8)) & 255u);
                    lastend = off + 1u > lastend ? off + 1u : lastend;
                }
                ps += 1u;
            }
        }
    } else
#pragma unroll
    for (int k = 0; k < RL_ITEMS; k++) {
        if ((uint32_t)k < valid) {
            const uint32_t p = (uint32_t)p0 + k;
            if (mask & (1u << k)) cur_rs = p;
            const uint32_t e4 = (uint32_t)(em >> (4 * k)) & 15u;
            const uint32_t can = e4 & 3u;
            const bool is_last = e4 & 4u;
            if (p0 + k >= d.in_off && p0 + k < in_end) {
                uint32_t off, lit, cnt, kk;
                if (cur_rs <= in_off32) { // the block's first run: chunking restarts at in_off
                    const uint32_t dd = p - in_off32;
                    kk = dd % 255u;
                    off = emitted_before(dd);
                    lit = kk < 4u;
                    cnt = kk >= 3u && (kk == 254u || is_last); // lib/rle.rs:212-226
                } else {
                    kk = (p - cur_rs) % 255u;
                    off = A + (uint32_t)(tc + ps - Ce);
                    lit = kk < 4u;
                    cnt = can - lit;
                }
                const uint32_t o = off - obase;
                if (lit) stage[o] = (uint8_t)((v[k >> 2] >> ((k & 3) * 8)) & 255u);
                if (cnt) stage[o + lit] = (uint8_t)(kk - 3u);
                const uint32_t e = off + lit + cnt;
                lastend = e > lastend ? e : lastend;
            }
            ps += can;
        }
    }
    atomicMax(&oend_s, lastend);
    __syncthreads();
    const uint32_t nout = oend_s > obase ? oend_s - obase : 0;
    uint8_t *out = bt.rle + (size_t)b * bt.S + obase;
    // head bytes up to the first 4-byte aligned address, aligned words, tail bytes
    const uint32_t mis = (uint32_t)((uintptr_t)out & 3u);
    const uint32_t head = mis ? (4u - mis < nout ? 4u - mis : nout) : 0u;
    const uint32_t nwords = (nout - head) / 4u;
    const uint32_t tail0 = head + nwords * 4u;
    if (threadIdx.x < head) out[threadIdx.x] = stage[threadIdx.x];
    for (uint32_t w = threadIdx.x; w < nwords; w += RL_THREADS) {
        const uint32_t o = head + w * 4u;
        const uint32_t x = (uint32_t)stage[o] | ((uint32_t)stage[o + 1] << 8) | ((uint32_t)stage[o + 2] << 16) |
                           ((uint32_t)stage[o + 3] << 24);
        *reinterpret_cast<uint32_t *>(out + o) = x;
    }
    if (threadIdx.x < nout - tail0) out[tail0 + threadIdx.x] = stage[tail0 + threadIdx.x];
}

// ---- host side ------------------------------------------------------------------------------------------------
struct PlanWs { // layout of ctx->plan_ws
    PlanArrays pa;
    uint32_t *crcacc;
    size_t bytes;
};

static size_t a256(size_t v) { return (v + 255) / 256 * 256; }

static PlanWs plan_layout(uint8_t *base, uint64_t n, uint32_t M)
{
    PlanWs w{};
    const uint64_t ntiles = (n + RL_TILE - 1) / RL_TILE;
    const uint64_t ngran = ntiles * GRAN_PER_TILE;
    const uint64_t maxblocks = n / ((uint64_t)(M - 1) * 4 / 5) + 4;
    uint8_t *p = base;
    auto take = [&](size_t bytes) {
        uint8_t *r = p;
        p += a256(bytes);
        return r;
    };
    w.pa.n = n;
    w.pa.M = M;
    w.pa.ntiles = (uint32_t)ntiles;
    w.pa.ngran = (uint32_t)ngran;
    w.pa.maxblocks = (uint32_t)maxblocks;
    w.pa.lrs = (uint32_t *)take((ntiles + 2) * 4);
    w.pa.frs = (uint32_t *)take((ntiles + 2) * 4);
    w.pa.csum = (uint32_t *)take((ntiles + 2) * 4);
    w.pa.tc = (uint64_t *)take((ntiles + 2) * 8);
    w.pa.cg = (uint32_t *)take((ngran + 2) * 4);
    w.pa.rsg = (uint32_t *)take((ngran + 2) * 4);
    w.pa.nrsg = (uint32_t *)take((ngran + 2) * 4);
    w.pa.blocks = (BlockDesc *)take(maxblocks * sizeof(BlockDesc));
    w.pa.aux = (BlockAux *)take(maxblocks * sizeof(BlockAux));
    w.pa.nblocks = (uint32_t *)take(256);
    w.crcacc = (uint32_t *)take(maxblocks * 4);
    w.bytes = (size_t)(p - base);
    return w;
}

// The split's tables (run starts and canonical RLE1 offsets per tile and per granule) over d_in[0..n): d_in must be
// 16-byte aligned.  They describe the runs of the input, not the blocks, so they hold for a split that begins at
// any block start inside the buffer (rle1_plan_split) -- a sharded rank builds them while its first block's start
// is still on its way from the rank before.
int rle1_plan_tables(bzh_ctx *ctx, const uint8_t *d_in, size_t n)
{
    hipStream_t st = ctx->stream;
    if (ctx->crc_pending) { // (a call that failed before it collected its CRCs: the second stream must be done with the workspace)
        (void)hipEventSynchronize(ctx->plan_ev[1]);
        ctx->crc_pending = false;
    }
    ctx->plan_blocks.clear();
    ctx->plan_open.clear();
    ctx->plan_crc_ok.clear();
    // (the buffer becomes the plan's input only once its tables are queued: a failure below must not leave a later
    // bzh_plan_split_device with a length it accepts and a workspace that is missing or too small)
    ctx->plan_in = nullptr;
    ctx->plan_n = 0;
    if (n == 0) {
        ctx->plan_in = d_in;
        return BZH_OK;
    }
    if (n > 0xFFFF0000ull) {
        bzh_set_error(ctx, "input of %zu bytes exceeds the 32-bit position range of one plan", n);
        return BZH_E_ARG;
    }
    PlanWs probe = plan_layout(nullptr, n, ctx->M);
    if (probe.bytes > ctx->plan_ws_size) {
        if (ctx->plan_ws) hipFree(ctx->plan_ws);
        ctx->plan_ws = nullptr;
        ctx->plan_ws_size = 0;
        if (hipMalloc(&ctx->plan_ws, probe.bytes) != hipSuccess) {
            bzh_set_error(ctx, "hipMalloc(%zu) for the plan failed", probe.bytes);
            return BZH_E_NOMEM;
        }
        ctx->plan_ws_size = probe.bytes;
    }
    PlanWs w = plan_layout((uint8_t *)ctx->plan_ws, n, ctx->M);
    PlanArrays &pa = w.pa;
    pa.in = d_in;
    KSpan ks(ctx, K_PLAN, 2 * n, 4); // two sweeps of the input
    plan_starts<<<dim3(pa.ntiles), RL_THREADS, 0, st>>>(pa);
    plan_carries<<<dim3(1), 1024, 0, st>>>(pa);
    plan_granules<<<dim3(pa.ntiles), RL_THREADS, 0, st>>>(pa);
    plan_tc<<<dim3(1), 1024, 0, st>>>(pa);
    HIP_TRY(ctx, hipGetLastError());
    ctx->plan_in = d_in;
    ctx->plan_n = n;
    return BZH_OK;
}

int rle1_plan(bzh_ctx *ctx, const uint8_t *d_in, size_t n, bool with_crc, bool crc_async)
{
    BZH_TRY(rle1_plan_tables(ctx, d_in, n));
    return rle1_plan_split(ctx, 0, with_crc, SIZE_MAX, crc_async);
}

// The sequential split over the tables of rle1_plan_tables, from input offset `start` (a block start) to the end of
// the buffer -- or until a block starts at or after `stop` (that block is the last one listed); block offsets are
// relative to the buffer.
int rle1_plan_split(bzh_ctx *ctx, size_t start, bool with_crc, size_t stop, bool crc_async)
{
    hipStream_t st = ctx->stream;
    const size_t n = ctx->plan_n;
    const uint8_t *d_in = ctx->plan_in;
    ctx->plan_blocks.clear();
    ctx->plan_open.clear();
    ctx->plan_crc_ok.clear();
    if (start > n) return BZH_E_ARG;
    if (n == 0 || start == n) return BZH_OK;
    PlanWs w = plan_layout((uint8_t *)ctx->plan_ws, n, ctx->M);
    PlanArrays &pa = w.pa;
    pa.in = d_in;
    pa.start = (uint32_t)start;
    pa.stop = stop >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)stop;
    // The second stream takes what nothing on the main stream waits for: the prefetching wavefront and -- crc_async, a
    // whole-path call -- the block CRCs, which nobody reads before the block headers are written (prepare_batch makes the
    // main stream wait for plan_ev[1] there; the host collects them with rle1_plan_crc_join).  64 us of CRC kernels
    // and two fills leave the step's critical path.  (Profiling keeps everything on one stream: spans must not overlap.)
    hipStream_t side = ctx->profiling ? nullptr : bzh_side_stream(ctx);
    ctx->crc_pending = false;
    {
        KSpan ks(ctx, K_PLAN, 0, 1);
        if (side) {
            hipEventRecord(ctx->plan_ev[0], st); // (the tables are complete)
            hipStreamWaitEvent(side, ctx->plan_ev[0], 0);
            plan_prefetch_kernel<<<dim3(1), 64, 0, side>>>(pa);
        }
        plan_split<<<dim3(1), 64 * SP_W, 0, st>>>(pa);
    }
    const size_t span = (size_t)((const uint8_t *)pa.nblocks - (const uint8_t *)pa.blocks) + 4;
    const bool beside = with_crc && crc_async && side != nullptr;
    // The block CRCs are queued right behind the split, over as many blocks as there could be (the kernels read the
    // count on the device), and everything the host needs -- count, descriptors with their CRCs, cut status -- comes
    // back in ONE copy: the plan costs the host one wait.
    const CrcTables *ct = nullptr;
    // blocks | aux | nblocks are consecutive in the workspace (plan_layout)
    ctx->plan_host.resize(span);
    if (with_crc) {
        BZH_TRY(crc_tables(ctx, &ct));
        hipStream_t sc = st;
        if (beside) {
            // (the descriptors come back first: the copy on the main stream must not see a CRC field half written -- it reads
            // them before the side stream may write, and the CRCs come back in a copy of their own)
            HIP_TRY(ctx, hipMemcpyAsync(ctx->plan_host.data(), pa.blocks, span, hipMemcpyDeviceToHost, st));
            hipEventRecord(ctx->plan_ev[0], st);
            hipStreamWaitEvent(side, ctx->plan_ev[0], 0);
            sc = side;
        }
        KSpan ks(ctx, K_CRC, n, 2);
        HIP_TRY(ctx, hipMemsetAsync(w.crcacc, 0, (size_t)pa.maxblocks * 4, sc));
        // (ranges of 64 KiB counted over all blocks: at most n / 64 KiB + one ragged range a block)
        const uint64_t ranges = n / ((uint64_t)CRC_TILE * CRC_WG_TILES) + pa.maxblocks + 1;
        crc_tiles_flat<<<dim3((uint32_t)ranges), RL_THREADS, 0, sc>>>(d_in, pa.blocks, w.crcacc, ct, pa.nblocks);
        crc_finish<<<dim3(pa.maxblocks), 64, 0, sc>>>(pa.blocks, w.crcacc, 0, ct, pa.nblocks);
        if (beside) {
            const size_t need = (size_t)pa.maxblocks * sizeof(BlockDesc);
            if (need > ctx->crc_host_cap) {
                if (ctx->crc_host) hipHostFree(ctx->crc_host);
                ctx->crc_host = nullptr;
                ctx->crc_host_cap = 0;
                if (hipHostMalloc((void **)&ctx->crc_host, need * 2, hipHostMallocDefault) != hipSuccess) {
                    bzh_set_error(ctx, "hipHostMalloc(%zu) for the block CRCs failed", need * 2);
                    return BZH_E_NOMEM;
                }
                ctx->crc_host_cap = need * 2;
            }
            ctx->crc_host_len = need;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->crc_host, pa.blocks, need, hipMemcpyDeviceToHost, side));
            hipEventRecord(ctx->plan_ev[1], side);
            ctx->crc_pending = true;
        }
    }
    if (!beside) HIP_TRY(ctx, hipMemcpyAsync(ctx->plan_host.data(), pa.blocks, span, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    HIP_TRY(ctx, hipGetLastError());
    uint32_t nb = 0;
    memcpy(&nb, ctx->plan_host.data() + span - 4, 4);
    if (nb == 0 || nb == 0xFFFFFFFFu || nb > pa.maxblocks) {
        bzh_set_error(ctx, "block split failed (nb=%u)", nb);
        return BZH_E_HIP;
    }
    const BlockDesc *hb = reinterpret_cast<const BlockDesc *>(ctx->plan_host.data());
    const BlockAux *hax = reinterpret_cast<const BlockAux *>(ctx->plan_host.data() + ((const uint8_t *)pa.aux - (const uint8_t *)pa.blocks));
    ctx->plan_open.resize(nb);
    ctx->plan_blocks.resize(nb);
    for (uint32_t b = 0; b < nb; b++) {
        ctx->plan_open[b] = (uint8_t)hax[b].open;
        ctx->plan_blocks[b].in_off = hb[b].in_off;
        ctx->plan_blocks[b].in_len = hb[b].in_len;
        ctx->plan_blocks[b].rle_len = hb[b].rle_len;
        ctx->plan_blocks[b].crc = (with_crc && !beside) ? hb[b].crc : 0;
    }
    ctx->plan_crc_ok.assign(nb, (with_crc && !beside) ? 1 : 0);
    return BZH_OK;
}

// The CRCs a whole-path plan left to the second stream: waits for them (they are long there: the call is at its end) and
// fills them into plan_blocks.  No-op when none are pending.
int rle1_plan_crc_join(bzh_ctx *ctx)
{
    if (!ctx->crc_pending) return BZH_OK;
    ctx->crc_pending = false;
    HIP_TRY(ctx, hipEventSynchronize(ctx->plan_ev[1]));
    const BlockDesc *hb = reinterpret_cast<const BlockDesc *>(ctx->crc_host);
    const size_t nb = std::min(ctx->plan_blocks.size(), ctx->crc_host_len / sizeof(BlockDesc));
    for (size_t b = 0; b < nb; b++) {
        ctx->plan_blocks[b].crc = hb[b].crc;
        ctx->plan_crc_ok[b] = 1;
    }
    return BZH_OK;
}

// Block CRCs (lib/crc32.rs:31-48 over the raw bytes each block consumed) of plan blocks [b0, b1): into
// the device descriptors (the block headers read them there) and into ctx->plan_blocks.
int rle1_plan_crc(bzh_ctx *ctx, size_t b0, size_t b1)
{
    BZH_TRY(rle1_plan_crc_join(ctx));
    if (b1 > ctx->plan_blocks.size() || b0 > b1) return BZH_E_STATE;
    while (b0 < b1 && ctx->plan_crc_ok[b0]) b0++;
    while (b1 > b0 && ctx->plan_crc_ok[b1 - 1]) b1--;
    if (b0 == b1) return BZH_OK;
    hipStream_t st = ctx->stream;
    PlanWs w = plan_layout((uint8_t *)ctx->plan_ws, ctx->plan_n, ctx->M);
    PlanArrays &pa = w.pa;
    const uint32_t nb = (uint32_t)(b1 - b0);
    uint64_t maxlen = 0;
    for (size_t b = b0; b < b1; b++) maxlen = std::max<uint64_t>(maxlen, ctx->plan_blocks[b].in_len);
    const CrcTables *ct = nullptr;
    BZH_TRY(crc_tables(ctx, &ct));
    HIP_TRY(ctx, hipMemsetAsync(w.crcacc + b0, 0, (size_t)nb * 4, st));
    const uint32_t ctiles = (uint32_t)((maxlen + CRC_TILE - 1) / CRC_TILE);
    for (uint32_t k0 = 0; k0 < nb; k0 += 32768) { // grid.y limit
        const uint32_t cnt = nb - k0 < 32768 ? nb - k0 : 32768;
        crc_tiles<<<dim3(std::min(1024u, (ctiles + CRC_WG_TILES - 1) / CRC_WG_TILES), cnt), RL_THREADS, 0, st>>>(ctx->plan_in, pa.blocks + b0 + k0, w.crcacc + b0 + k0, ct, nullptr);
    }
    crc_finish<<<dim3(nb), 64, 0, st>>>(pa.blocks + b0, w.crcacc + b0, nb, ct, nullptr);
    std::vector<BlockDesc> hb(nb);
    HIP_TRY(ctx, hipMemcpyAsync(hb.data(), pa.blocks + b0, nb * sizeof(BlockDesc), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    HIP_TRY(ctx, hipGetLastError());
    for (uint32_t k = 0; k < nb; k++) {
        ctx->plan_blocks[b0 + k].crc = hb[k].crc;
        ctx->plan_crc_ok[b0 + k] = 1;
    }
    return BZH_OK;
}

// Fills bt.rle / bt.n / bt.desc for plan blocks [b0, b0+B).
int rle1_emit(bzh_ctx *ctx, size_t b0, uint32_t B)
{
    if (B == 0) return BZH_OK;
    const bzh_ctx *pc = ctx->parent ? ctx->parent : ctx; // lanes read the owner's plan
    PlanWs w = plan_layout((uint8_t *)pc->plan_ws, pc->plan_n, pc->M);
    EmitArgs ea{};
    ea.in = pc->plan_in;
    ea.n = pc->plan_n;
    ea.rsg = w.pa.rsg;
    ea.tc = w.pa.tc;
    ea.blocks = w.pa.blocks + b0;
    ea.aux = w.pa.aux + b0;
    uint64_t maxspan = 0;
    for (uint32_t b = 0; b < B; b++) {
        const bzh_block &pb = pc->plan_blocks[b0 + b];
        const uint64_t t0 = pb.in_off / RL_TILE, t1 = (pb.in_off + pb.in_len - 1) / RL_TILE;
        maxspan = std::max<uint64_t>(maxspan, t1 - t0 + 1);
    }
    ctx->bt.pdesc = w.pa.blocks + b0; // (the block headers read the CRCs where crc_finish writes them)
    KSpan ks(ctx, K_RLE1_EMIT, 2 * (uint64_t)ctx->k_cur_ntotal);
    rle1_emit_kernel<<<dim3((uint32_t)maxspan, B), RL_THREADS, 0, ctx->stream>>>(ea, ctx->bt);
    HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

// CRC of d_in[0..n) as one block (seam for the parity tests).
int crc_device(bzh_ctx *ctx, const uint8_t *d_in, size_t n, uint32_t *crc_out)
{
    hipStream_t st = ctx->stream;
    const CrcTables *ct = nullptr;
    BZH_TRY(crc_tables(ctx, &ct));
    BlockDesc d{0, n, 0, 0};
    BlockDesc *dd = ctx->bt.desc;        // scratch
    uint32_t *acc = ctx->bt.c_big;       // scratch
    HIP_TRY(ctx, hipMemcpyAsync(dd, &d, sizeof d, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(acc, 0, 4, st));
    const uint32_t ctiles = (uint32_t)((n + CRC_TILE - 1) / CRC_TILE);
    if (ctiles) crc_tiles<<<dim3(std::min(4096u, (ctiles + CRC_WG_TILES - 1) / CRC_WG_TILES), 1), RL_THREADS, 0, st>>>(d_in, dd, acc, ct, nullptr);
    crc_finish<<<dim3(1), 64, 0, st>>>(dd, acc, 1, ct, nullptr);
    HIP_TRY(ctx, hipMemcpyAsync(&d, dd, sizeof d, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    *crc_out = d.crc;
    return BZH_OK;
}
