// api.hip -- the C ABI of libbzhip.so (include/bzhip.h): context, workspace, whole-path drivers
// and the host-pointer stage seams used by the parity tests.
//
// Whole path = the loop of banzai::encode (reference lib/lib.rs:84-132) turned inside out:
//   plan    : RLE1 run scan + every block cut + block CRCs            (rle1.hip)
//   batches : RLE1 emit -> BWT -> MTF/RLE2 -> Huffman tables + bit lengths (all blocks at once)
//   pack    : block headers + payload bits written straight at their final bit offset
//   assemble: "BZh9", stream CRC fold, footer
#include <stdarg.h>
#include <algorithm>
#include <atomic>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "common.h"

static void stream_join(bzh_ctx *ctx); // waits for a streaming pass in flight (defined with bzh_stream_*)

// The error text has two writers while a streaming pass is in flight (the caller's thread and the pass's
// worker thread), so it is guarded; readers get a private copy (bzh_last_error).
void bzh_set_error(bzh_ctx *ctx, const char *fmt, ...)
{
    if (!ctx) return;
    std::lock_guard<std::mutex> g(ctx->err_mu);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
}

// Nothing unwinds across the C ABI (include/bzhip.h): every entry point runs inside this guard.  The
// reference's convention is the same -- errors are values (io::Result, lib/lib.rs:84-92), panics never
// cross the boundary.
template <typename F>
static int bzh_guard(bzh_ctx *ctx, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        bzh_set_error(ctx, "host allocation failed");
        return BZH_E_NOMEM;
    } catch (const std::exception &e) {
        bzh_set_error(ctx, "internal error: %s", e.what());
        return BZH_E_STATE;
    } catch (...) {
        bzh_set_error(ctx, "internal error");
        return BZH_E_STATE;
    }
}

// Only gfx950 code objects are in the library: "gfx950", optionally followed by feature flags
// (hipDeviceProp_t::gcnArchName reads e.g. "gfx950:sramecc+:xnack-").
extern "C" int bzh_arch_supported(const char *gcn_arch_name)
{
    if (!gcn_arch_name) return 0;
    if (strncmp(gcn_arch_name, "gfx950", 6) != 0) return 0;
    return gcn_arch_name[6] == 0 || gcn_arch_name[6] == ':';
}

hipEvent_t bzh_event(bzh_ctx *ctx)
{
    if (ctx->evnext == ctx->evpool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        ctx->evpool.push_back(e);
    }
    return ctx->evpool[ctx->evnext++];
}

extern "C" const char *bzh_strerror(int status)
{
    switch (status) {
    case BZH_OK: return "ok";
    case BZH_E_ARG: return "invalid argument";
    case BZH_E_NOMEM: return "out of memory";
    case BZH_E_HIP: return "HIP runtime error or no usable gfx950 device";
    case BZH_E_CAP: return "output buffer too small";
    case BZH_E_STATE: return "call sequence error";
    default: return "unknown status";
    }
}

extern "C" const char *bzh_last_error(const bzh_ctx *cctx)
{
    if (!cctx) return "no context";
    bzh_ctx *ctx = const_cast<bzh_ctx *>(cctx);
    std::lock_guard<std::mutex> g(ctx->err_mu);
    memcpy(ctx->err_out, ctx->err, sizeof ctx->err_out);
    return ctx->err_out;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The context's second stream with its events (the suffix sort's big-list path, the plan's CRCs), created on first use.
hipStream_t bzh_side_stream(bzh_ctx *ctx)
{
    if (ctx->side_stream) return ctx->side_stream;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; k++) {
        if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < k; j++) hipEventDestroy(ev[j]);
            hipStreamDestroy(s);
            return nullptr;
        }
    }
    ctx->side_ev[0] = ev[0];
    ctx->side_ev[1] = ev[1];
    ctx->plan_ev[0] = ev[2];
    ctx->plan_ev[1] = ev[3];
    ctx->side_stream = s;
    return s;
}

template <typename T>
static void carve(uint8_t *&p, T *&dst, size_t count)
{
    dst = reinterpret_cast<T *>(p);
    p += align_up(count * sizeof(T), 256);
}

// Lays the batch arrays out in the arena; with base == nullptr only measures.
static size_t layout_batch(Batch &bt, uint8_t *base, uint32_t B, uint32_t M)
{
    bt.B = B;
    bt.M = M;
    bt.S = (uint32_t)align_up((size_t)M + 1, SORT_TILE);
    bt.TPB = bt.S / SORT_TILE;
    const size_t S = bt.S, NB = B;
    const size_t MT = (S + MTF_TILE - 1) / MTF_TILE;
    const size_t PT = (S + 64 + PACK_TILE - 1) / PACK_TILE;
    uint8_t *p = base;
    carve(p, bt.rle, NB * S);
    carve(p, bt.n, NB);
    carve(p, bt.bwt, NB * S);
    carve(p, bt.ptr, NB);
    carve(p, bt.hasbyte, NB * 256);
    carve(p, bt.rank, NB * S);
    carve(p, bt.sa, NB * S);
    carve(p, bt.headp, NB * S);
    carve(p, bt.binned, NB * S);
    carve(p, bt.listA, NB * S);
    carve(p, bt.listB, NB * S);
    carve(p, bt.listC, NB * S);
    carve(p, bt.listD, NB * S);
    carve(p, bt.hist, NB * 512 * bt.TPB);
    carve(p, bt.dbase, NB * DB_STRIDE);
    carve(p, bt.dtot, NB * DB_STRIDE);
    carve(p, bt.flg, NB * S);
    carve(p, bt.tagg, NB * bt.TPB);
    { // round state of the suffix sort: RS_ROWS words per block, contiguous (one memset clears it)
        uint32_t *rs = nullptr;
        const size_t oA = (RS_ROWS * NB + 8 + SUMMARY_WORDS + 1) & ~(size_t)1; // 64-bit counter: even word index
        carve(p, rs, oA + 4);
        uint32_t **f[RS_ROWS] = {&bt.st_mode, &bt.st_h, &bt.st_nbig, &bt.st_ntail, &bt.c_big, &bt.c_small, &bt.c_tail,
                                 &bt.c_prog, &bt.gateS, &bt.gateA, &bt.gateR, &bt.gateT, &bt.actS, &bt.actA, &bt.actR,
                                 &bt.actT, &bt.actQ, nullptr, &bt.c_nolist, &bt.c_groups, &bt.scratch, &bt.st_tdst}; // row 17: sixth list (bwt.hip)
        for (int k = 0; k < RS_ROWS; k++)
            if (f[k]) *f[k] = rs ? rs + (size_t)k * NB : nullptr;
        bt.nlist = rs ? rs + RS_ROWS * NB : nullptr;
        bt.summary = rs ? rs + RS_ROWS * NB + 8 : nullptr;
        bt.stat_A = rs ? reinterpret_cast<unsigned long long *>(rs + oA) : nullptr;
    }
    carve(p, bt.chain, NB * 4);
    carve(p, bt.pshrink, NB * 4);
    carve(p, bt.errflag, 64);
    carve(p, bt.gidof, NB * S);
    carve(p, bt.grank, 2 * NB * GID_MAX);
    carve(p, bt.gcount, NB);
    carve(p, bt.gwide, 64);
    { // bucket-first initial sort (bwt_msd.h): only levels whose blocks can reach MS_MIN_N bytes ever use it
        const size_t MB = M >= MS_MIN_N ? NB : 0;
        carve(p, bt.ms_bgcur, MB * 65536);
        carve(p, bt.ms_pool, MB * MS_BG_ROW + (size_t)MS_LEVELS * MB * MS_SEG_SLOTS * MS_SEG_ROW);
        carve(p, bt.ms_segcur, (size_t)MS_LEVELS * MB * MS_SEG_SLOTS * 256);
        carve(p, bt.ms_units, MB * MS_UNIT_CAP);
        carve(p, bt.ms_segs, (size_t)(MS_LEVELS + 1) * MB * MS_SEG_SLOTS);
        carve(p, bt.ms_items, (size_t)(MS_LEVELS + 1) * MB * MS_ITEM_CAP);
        carve(p, bt.ms_cnt, MS_CNT_WORDS + (size_t)(MS_LEVELS + 7) * NB + 2 + 6 * NB * MS_UNIT_CAP);
        carve(p, bt.ms_np, NB);
        carve(p, bt.ms_old, NB);
        carve(p, bt.ms_new, NB);
        carve(p, bt.ms_bincur, NB * 256);
    }
    carve(p, bt.mtfpos, NB * S);
    carve(p, bt.tilelist, NB * MT * 256);
    carve(p, bt.tinfo, NB * MT * 4);
    carve(p, bt.syms, NB * (S + 64));
    carve(p, bt.m, NB);
    carve(p, bt.freqs, NB * 258);
    carve(p, bt.nsyms, NB);
    carve(p, bt.tfreq, NB * 3 * 258);
    carve(p, bt.lens, NB * 3 * 258);
    carve(p, bt.lens2, 2 * NB * 3 * 258);
    carve(p, bt.lfit, 2 * NB * 3);
    carve(p, bt.ntab, NB);
    carve(p, bt.codes, NB * 258);
    carve(p, bt.hdr, NB * HDR_BYTES);
    carve(p, bt.hdrbits, NB * 4);
    carve(p, bt.bits, NB);
    carve(p, bt.bitoff, NB + 1);
    carve(p, bt.packgate, 4);
    carve(p, bt.symbits, NB * PT);
    carve(p, bt.desc, NB);
    bt.pdesc = bt.desc; // (rle1_emit points it at the plan's descriptors of the batch)
    { // "fixed" Huffman mode (optional)
        const size_t selmax = (S + 64 + 49) / 50 + 2;
        carve(p, bt.fx_tfreq, NB * FX_TABLES * 258);
        carve(p, bt.fx_lens, NB * FX_TABLES * 258);
        carve(p, bt.fx_codes, NB * FX_TABLES * 258);
        carve(p, bt.fx_sel, NB * selmax);
        carve(p, bt.fx_selbits, NB * align_up((selmax * 6 + 7) / 8 + 8, 64));
        carve(p, bt.fx_hdr, NB * FX_HDR_BYTES);
    }
    return (size_t)(p - base);
}

// Makes the arena hold batches of `blocks` blocks (at most max_batch).  It only ever grows: to the size asked for,
// rounded up so that a stream of growing batches does not reallocate every time.  Nothing is in flight on the
// arena when this is called (every entry point waits for its own work before it returns).
static int ensure_arena(bzh_ctx *ctx, uint32_t blocks, size_t min_bytes = 0)
{
    blocks = std::max<uint32_t>(1, std::min<uint32_t>(blocks, ctx->max_batch));
    if (ctx->arena && blocks <= ctx->arena_blocks && min_bytes <= ctx->arena_size) return BZH_OK;
    uint32_t want = std::min<uint32_t>(ctx->max_batch, std::max<uint32_t>(blocks, 8));
    if (want > 8) want = std::min<uint32_t>(ctx->max_batch, (want + 15u) & ~15u);
    Batch probe{};
    const size_t bytes = std::max(layout_batch(probe, nullptr, want, ctx->M), min_bytes);
    if (ctx->arena) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(ctx->arena);
        ctx->arena = nullptr;
        ctx->arena_blocks = 0;
    }
    if (hipMalloc((void **)&ctx->arena, bytes) != hipSuccess) {
        bzh_set_error(ctx, "hipMalloc(%zu) for a %u-block workspace failed", bytes, want);
        return BZH_E_NOMEM;
    }
    ctx->arena_size = bytes;
    ctx->arena_blocks = want;
    layout_batch(ctx->bt, ctx->arena, want, ctx->M);
    return BZH_OK;
}

extern "C" int bzh_create(bzh_ctx **out, int device, int level, int max_batch)
{
    return bzh_guard(nullptr, [&]() -> int {
    if (!out || level < 1 || level > 9 || max_batch < 0) return BZH_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BZH_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return BZH_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || !bzh_arch_supported(prop.gcnArchName)) return BZH_E_HIP;
    bzh_ctx *ctx = new (std::nothrow) bzh_ctx();
    if (!ctx) return BZH_E_NOMEM;
    ctx->device = device;
    ctx->level = level;
    ctx->M = 100000u * (uint32_t)level - 1u; // lib/rle.rs:121
    // default: a batch covers about 520 MB of RLE1 output at level 9 (576 blocks; 1,024 blocks from level 5 down).  A pass
    // costs about half a millisecond of latency chains whatever its size (the plan, the late doubling rounds, the Huffman
    // heaps), so a long stream is cheaper in few, large batches -- 1 GB on one MI355X: 87.8 ms in batches of 128 blocks,
    // 81.6 ms in 256s, 78.2 ms in 576s (profiles/r06_multibatch.txt).  The arena is sized for the batch actually planned
    // (ensure_arena: 45 MB a block), so only an input that fills such a batch pays for it: 26 GB of 288 GB.
    ctx->max_batch = max_batch ? (uint32_t)max_batch : std::min<uint32_t>(1024u, 576u * 9u / (uint32_t)level);
    // a streaming pass is worth launching once a full batch of input is pending
    ctx->strm.min_feed = std::min<size_t>((size_t)128 << 20, (size_t)ctx->max_batch * (ctx->M + 1));
    Batch probe{};
    (void)layout_batch(probe, nullptr, 1, ctx->M);
    if (probe.TPB > 1024 || ctx->max_batch > 1024) {
        delete ctx;
        return BZH_E_ARG;
    }
    // The workspace arena (about 45 MB per block of a batch, 5.8 GB for the 112 blocks of a 100 MB input) is NOT allocated here:
    // ensure_arena sizes it for the batches actually planned, so a 1 MB file does not pay for a 128-block arena.
    ctx->S = probe.S;
    ctx->bt.S = probe.S;
    ctx->bt.TPB = probe.TPB;
    ctx->bt.M = ctx->M;
    if (hipHostMalloc((void **)&ctx->h_pinned, sizeof(uint32_t) * (ctx->max_batch * 8 + 64 + (MAX_ROUNDS + 1) * SUMMARY_WORDS), hipHostMallocCoherent) != hipSuccess) {
        delete ctx;
        return BZH_E_NOMEM;
    }
    memset(ctx->h_pinned, 0, sizeof(uint32_t) * (ctx->max_batch * 8 + 64 + (MAX_ROUNDS + 1) * SUMMARY_WORDS)); // (no stale sequence words)
    *out = ctx;
    return BZH_OK;
    });
}

extern "C" void bzh_destroy(bzh_ctx *ctx)
{
    if (!ctx) return;
    // a streaming pass in flight keeps launching kernels on the arena and the staging buffers: it ends
    // first, then the device drains, and only then is anything freed
    try {
        if (ctx->strm.worker.joinable()) ctx->strm.worker.join();
    } catch (...) {
    }
    hipSetDevice(ctx->device);
    // only this context's work has to end (other contexts of the device keep running): its stream, the lanes'
    // streams, the streaming copy stream
    hipStreamSynchronize(ctx->stream);
    for (bzh_ctx *l : ctx->lanes)
        if (l->stream) hipStreamSynchronize(l->stream);
    if (ctx->strm.copy_stream) hipStreamSynchronize(ctx->strm.copy_stream);
    auto drop_side = [](bzh_ctx *c) {
        if (!c->side_stream) return;
        hipStreamSynchronize(c->side_stream);
        hipEventDestroy(c->side_ev[0]);
        hipEventDestroy(c->side_ev[1]);
        hipEventDestroy(c->plan_ev[0]);
        hipEventDestroy(c->plan_ev[1]);
        hipStreamDestroy(c->side_stream);
        c->side_stream = nullptr;
        if (c->side2_stream) {
            hipStreamSynchronize(c->side2_stream);
            hipEventDestroy(c->side_ev[2]);
            hipStreamDestroy(c->side2_stream);
            c->side2_stream = nullptr;
        }
    };
    drop_side(ctx);
    for (bzh_ctx *l : ctx->lanes) drop_side(l);
    for (hipEvent_t e : ctx->evpool) hipEventDestroy(e);
    for (bzh_ctx *l : ctx->lanes) {
        for (hipEvent_t e : l->evpool) hipEventDestroy(e);
        if (l->stream) hipStreamDestroy(l->stream);
        if (l->h_pinned) hipHostFree(l->h_pinned);
        delete l;
    }
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->plan_ws) hipFree(ctx->plan_ws);
    if (ctx->d_stage_in) hipFree(ctx->d_stage_in);
    if (ctx->d_stage_out) hipFree(ctx->d_stage_out);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    if (ctx->crc_host) hipHostFree(ctx->crc_host);
    if (ctx->d_crctab) hipFree(ctx->d_crctab);
    for (int k = 0; k < 2; k++)
        if (ctx->strm.d_buf[k]) hipFree(ctx->strm.d_buf[k]);
    if (ctx->strm.h_out) hipHostFree(ctx->strm.h_out);
    for (int k = 0; k < 2; k++)
        if (ctx->strm.d_out[k]) hipFree(ctx->strm.d_out[k]);
    if (ctx->strm.copy_stream) hipStreamDestroy(ctx->strm.copy_stream);
    delete ctx;
}

extern "C" int bzh_set_stream(bzh_ctx *ctx, void *hip_stream)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx) return BZH_E_ARG;
    ctx->stream = (hipStream_t)hip_stream;
    return BZH_OK;
    });
}

extern "C" int bzh_set_lanes(bzh_ctx *ctx, int lanes)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (lanes != 1 && lanes != 2)) return BZH_E_ARG;
    ctx->nlanes = lanes;
    return BZH_OK;
    });
}

extern "C" int bzh_set_mode(bzh_ctx *ctx, int mode)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (mode != BZH_MODE_REFERENCE && mode != BZH_MODE_FIXED)) return BZH_E_ARG;
    ctx->mode = mode;
    for (bzh_ctx *l : ctx->lanes) l->mode = mode;
    return BZH_OK;
    });
}

extern "C" int bzh_set_profiling(bzh_ctx *ctx, int enabled)
{
    return bzh_guard(ctx, [&]() -> int {
    if (!ctx) return BZH_E_ARG;
    ctx->profiling = enabled ? 1 : 0;
    return BZH_OK;
    });
}

extern "C" int bzh_debug_fault(bzh_ctx *ctx, int kind)
{
    return bzh_guard(ctx, [&]() -> int {
    if (!ctx || kind < 0 || kind > 2) return BZH_E_ARG;
    ctx->debug_fault = (uint32_t)kind;
    return BZH_OK;
    });
}

extern "C" int bzh_get_stats(const bzh_ctx *ctx, bzh_stats *out)
{
    return bzh_guard(const_cast<bzh_ctx *>(ctx), [&]() -> int {
    if (!ctx || !out) return BZH_E_ARG;
    *out = ctx->stats;
    return BZH_OK;
    });
}

static void kstats_reset(bzh_ctx *ctx);
static void stats_begin(bzh_ctx *ctx)
{
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->evnext = 0;
    ctx->sort_spans.clear();
    kstats_reset(ctx);
}

static void kstats_reset(bzh_ctx *ctx)
{
    ctx->kspans.clear();
    for (int k = 0; k < K_COUNT; k++) {
        ctx->k_ms[k] = 0;
        ctx->k_bytes[k] = 0;
        ctx->k_launch[k] = 0;
    }
}

// Event pairs -> milliseconds per kernel class (the stream has been waited for).
static void kstats_collect(bzh_ctx *ctx)
{
    for (auto &r : ctx->kspans) {
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ctx->k_ms[r.cls] += t;
    }
    ctx->kspans.clear();
}

extern "C" int bzh_get_kernel_stats(const bzh_ctx *ctx, bzh_kstat *out, size_t max, size_t *count)
{
    return bzh_guard(const_cast<bzh_ctx *>(ctx), [&]() -> int {
    if (!ctx || !count || (max && !out)) return BZH_E_ARG;
    *count = K_COUNT;
    if (max < (size_t)K_COUNT) return BZH_E_CAP;
    for (int k = 0; k < K_COUNT; k++) {
        memset(&out[k], 0, sizeof out[k]);
        strncpy(out[k].name, KCLASS_NAME[k], sizeof out[k].name - 1);
        out[k].ms = ctx->k_ms[k];
        out[k].launches = ctx->k_launch[k];
        out[k].alg_bytes = ctx->k_bytes[k];
    }
    return BZH_OK;
    });
}

static void stats_collect_sort(bzh_ctx *ctx)
{
    kstats_collect(ctx);
    double ms = 0;
    for (auto &sp : ctx->sort_spans) {
        float t = 0;
        if (hipEventElapsedTime(&t, sp.first, sp.second) == hipSuccess) ms += t;
    }
    ctx->stats.ms_bwt_sort = ms;
}

static int ensure_stage(bzh_ctx *ctx, uint8_t *&buf, size_t &cur, size_t need)
{
    if (need <= cur) return BZH_OK;
    if (buf) hipFree(buf);
    buf = nullptr;
    cur = 0;
    size_t want = align_up(need + need / 8 + 4096, 4096);
    if (hipMalloc((void **)&buf, want) != hipSuccess) {
        bzh_set_error(ctx, "hipMalloc(%zu) failed", want);
        return BZH_E_NOMEM;
    }
    cur = want;
    return BZH_OK;
}

// ---- stage seam: BWT ---------------------------------------------------------------------------------
extern "C" int bzh_bwt_batch(bzh_ctx *ctx, const uint8_t *in, const uint64_t *offs, const uint32_t *lens,
                             size_t nblk, uint8_t *bwt_out, uint32_t *ptr, uint8_t *has_byte)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !in || !offs || !lens || !bwt_out || !ptr || !has_byte) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stats_begin(ctx);
    BZH_TRY(ensure_arena(ctx, (uint32_t)std::min<size_t>(nblk, ctx->max_batch)));
    Batch &bt = ctx->bt;
    for (size_t k0 = 0; k0 < nblk; k0 += ctx->max_batch) {
        uint32_t B = (uint32_t)std::min<size_t>(ctx->max_batch, nblk - k0);
        uint32_t nmax = 0;
        uint64_t ntotal = 0;
        for (uint32_t b = 0; b < B; b++) {
            uint32_t n = lens[k0 + b];
            ntotal += n;
            if (n == 0 || n > ctx->M) {
                bzh_set_error(ctx, "block %zu length %u outside 1..%u", k0 + b, n, ctx->M);
                return BZH_E_ARG;
            }
            nmax = std::max(nmax, n);
            HIP_TRY(ctx, hipMemcpyAsync(bt.rle + (size_t)b * bt.S, in + offs[k0 + b], n, hipMemcpyHostToDevice,
                                        ctx->stream));
        }
        HIP_TRY(ctx, hipMemcpyAsync(bt.n, lens + k0, B * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        BZH_TRY(bwt_run(ctx, B, nmax, ntotal));
        for (uint32_t b = 0; b < B; b++)
            HIP_TRY(ctx, hipMemcpyAsync(bwt_out + offs[k0 + b], bt.bwt + (size_t)b * bt.S, lens[k0 + b],
                                        hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ptr + k0, bt.ptr, B * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(has_byte + k0 * 256, bt.hasbyte, (size_t)B * 256, hipMemcpyDeviceToHost,
                                    ctx->stream));
        HIP_TRY(ctx, bzh_stream_wait(ctx->stream));
    }
    if (ctx->profiling) stats_collect_sort(ctx);
    return BZH_OK;
    });
}

extern "C" int bzh_bwt(bzh_ctx *ctx, const uint8_t *in, size_t n, uint8_t *bwt_out, uint32_t *ptr,
                       uint8_t *has_byte)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !ptr || !has_byte) return BZH_E_ARG;
    if (n == 0) { // lib/bwt.rs:535-541
        memset(has_byte, 0, 256);
        *ptr = UINT32_MAX;
        return BZH_OK;
    }
    if (n > ctx->M) return BZH_E_ARG;
    uint64_t off = 0;
    uint32_t len = (uint32_t)n;
    return bzh_bwt_batch(ctx, in, &off, &len, 1, bwt_out, ptr, has_byte);
    });
}

// ---- verification tooling: inverse BWT -----------------------------------------------
extern "C" int bzh_unbwt_batch(bzh_ctx *ctx, const uint8_t *bwt, const uint64_t *offs, const uint32_t *lens,
                               const uint32_t *ptr, size_t nblk, uint8_t *out)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !bwt || !offs || !lens || !ptr || !out) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BZH_TRY(ensure_arena(ctx, (uint32_t)std::min<size_t>(nblk, ctx->max_batch)));
    Batch &bt = ctx->bt;
    for (size_t k0 = 0; k0 < nblk; k0 += ctx->max_batch) {
        const uint32_t B = (uint32_t)std::min<size_t>(ctx->max_batch, nblk - k0);
        uint32_t nmax = 0;
        for (uint32_t b = 0; b < B; b++) {
            const uint32_t n = lens[k0 + b];
            if (n == 0 || n > ctx->M || ptr[k0 + b] >= n) {
                bzh_set_error(ctx, "block %zu: length %u / pointer %u out of range", k0 + b, n, ptr[k0 + b]);
                return BZH_E_ARG;
            }
            nmax = std::max(nmax, n);
            HIP_TRY(ctx, hipMemcpyAsync(bt.bwt + (size_t)b * bt.S, bwt + offs[k0 + b], n, hipMemcpyHostToDevice, ctx->stream));
        }
        HIP_TRY(ctx, hipMemcpyAsync(bt.n, lens + k0, B * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(bt.ptr, ptr + k0, B * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        BZH_TRY(unbwt_run(ctx, B, nmax));
        for (uint32_t b = 0; b < B; b++)
            HIP_TRY(ctx, hipMemcpyAsync(out + offs[k0 + b], bt.mtfpos + (size_t)b * bt.S, lens[k0 + b], hipMemcpyDeviceToHost,
                                        ctx->stream));
        HIP_TRY(ctx, bzh_stream_wait(ctx->stream));
    }
    return BZH_OK;
    });
}

// Forward + inverse transform of plan blocks [b0, b1) entirely on the device: RLE1 bytes -> BWT -> inverse BWT,
// compared with the RLE1 bytes.  *mismatches = differing bytes (0 for a correct transform).
extern "C" int bzh_bwt_roundtrip_device(bzh_ctx *ctx, size_t b0, size_t b1, uint64_t *mismatches)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !mismatches || b0 > b1) return BZH_E_ARG;
    if (b1 > ctx->plan_blocks.size()) return BZH_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ctx->evnext = 0; // (the pool is reused from the start by every entry point that takes events)
    ctx->sort_spans.clear(); // (spans recorded against the old use of those events must not survive the rewind)
    kstats_reset(ctx);
    BZH_TRY(ensure_arena(ctx, (uint32_t)std::min<size_t>(b1 - b0, ctx->max_batch)));
    unsigned long long *d_acc = ctx->bt.stat_A; // (the forward sort has read it back by the time it is reused)
    unsigned long long total = 0;
    for (size_t k0 = b0; k0 < b1; k0 += ctx->max_batch) {
        const uint32_t B = (uint32_t)std::min<size_t>(ctx->max_batch, b1 - k0);
        uint32_t nmax = 0;
        uint64_t ntotal = 0;
        for (uint32_t b = 0; b < B; b++) {
            nmax = std::max(nmax, ctx->plan_blocks[k0 + b].rle_len);
            ntotal += ctx->plan_blocks[k0 + b].rle_len;
        }
        BZH_TRY(rle1_emit(ctx, k0, B));
        BZH_TRY(bwt_run(ctx, B, nmax, ntotal));
        BZH_TRY(unbwt_run(ctx, B, nmax));
        HIP_TRY(ctx, hipMemsetAsync(d_acc, 0, sizeof(unsigned long long), st));
        BZH_TRY(unbwt_compare(ctx, B, nmax, d_acc));
        unsigned long long part = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&part, d_acc, sizeof part, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, bzh_stream_wait(st));
        total += part;
    }
    *mismatches = total;
    return BZH_OK;
    });
}

// ---- stage seam: MTF + RLE2 ---------------------------------------------------------------------------
extern "C" int bzh_mtf(bzh_ctx *ctx, const uint8_t *bwt, size_t n, const uint8_t *has_byte, uint16_t *syms,
                       size_t *m, uint32_t *freqs, uint32_t *num_syms)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !bwt || !has_byte || !syms || !m || !freqs || !num_syms || n == 0 || n > ctx->M) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stats_begin(ctx);
    BZH_TRY(ensure_arena(ctx, 1));
    Batch &bt = ctx->bt;
    hipStream_t st = ctx->stream;
    uint32_t n32 = (uint32_t)n;
    HIP_TRY(ctx, hipMemcpyAsync(bt.bwt, bwt, n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.n, &n32, sizeof n32, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.hasbyte, has_byte, 256, hipMemcpyHostToDevice, st));
    BZH_TRY(mtf_run(ctx, 1, n32));
    uint32_t m32 = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&m32, bt.m, sizeof m32, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(num_syms, bt.nsyms, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(freqs, bt.freqs, 258 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    if (m32 == 0 || m32 > n32 + 1) {
        bzh_set_error(ctx, "mtf produced m=%u for n=%u", m32, n32);
        return BZH_E_HIP;
    }
    HIP_TRY(ctx, hipMemcpyAsync(syms, bt.syms, (size_t)m32 * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    *m = m32;
    return BZH_OK;
    });
}

// ---- stage seam: Huffman ----------------------------------------------------------------------------------
// The device path always writes a whole block (header + symbol map + payload).  For the seam the
// block is built with a fixed dummy header (crc 0, ptr 0, only byte 0 present: 105 + 32 bits) and
// the host strips those 137 bits, leaving exactly what huffman::encode writes (lib/huffman.rs:464-572).
extern "C" int bzh_huffman(bzh_ctx *ctx, const uint16_t *syms, size_t m, uint32_t num_syms, const uint32_t *freqs,
                           uint8_t *bits_out, size_t cap, uint64_t *nbits, uint8_t *code_lengths,
                           uint32_t *num_tables)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !syms || !freqs || !bits_out || !nbits || m == 0 || m > (size_t)ctx->M + 1 || num_syms < 3 ||
        num_syms > 258)
        return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    stats_begin(ctx);
    struct ModeGuard { // the seam is the reference's huffman::encode whatever mode the context is in
        bzh_ctx *c;
        int keep;
        ~ModeGuard() { c->mode = keep; }
    } guard{ctx, ctx->mode};
    ctx->mode = BZH_MODE_REFERENCE;
    BZH_TRY(ensure_arena(ctx, 1));
    Batch &bt = ctx->bt;
    hipStream_t st = ctx->stream;
    const uint32_t m32 = (uint32_t)m;
    uint8_t hb[256] = {1};
    BlockDesc d{};
    uint32_t zero = 0;
    HIP_TRY(ctx, hipMemcpyAsync(bt.syms, syms, m * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.m, &m32, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.nsyms, &num_syms, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.freqs, freqs, 258 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.hasbyte, hb, 256, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bt.desc, &d, sizeof d, hipMemcpyHostToDevice, st));
    bt.pdesc = bt.desc;
    HIP_TRY(ctx, hipMemcpyAsync(bt.ptr, &zero, 4, hipMemcpyHostToDevice, st));
    BZH_TRY(huff_prepare(ctx, 1, m32));
    uint64_t total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, bt.bitoff + 1, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    const size_t bytes = (size_t)((total + 31) / 32 * 4);
    BZH_TRY(ensure_stage(ctx, ctx->d_stage_out, ctx->stage_out_size, bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stage_out, 0, bytes, st));
    BZH_TRY(huff_pack(ctx, 1, m32, ctx->d_stage_out, 0));
    std::vector<uint8_t> tmp(bytes + 8, 0);
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ctx->d_stage_out, bytes, hipMemcpyDeviceToHost, st));
    uint32_t nt = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nt, bt.ntab, 4, hipMemcpyDeviceToHost, st));
    if (code_lengths) HIP_TRY(ctx, hipMemcpyAsync(code_lengths, bt.lens, 3 * 258, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    if (num_tables) *num_tables = nt;
    const uint64_t skip = 105 + 32;
    const uint64_t pay = total - skip;
    *nbits = pay;
    const size_t need = (size_t)((pay + 7) / 8);
    if (need > cap) return BZH_E_CAP;
    for (size_t k = 0; k < need; k++) { // shift left by 137 bits = 17 bytes + 1 bit
        const size_t src = k + skip / 8;
        const unsigned sh = skip % 8;
        bits_out[k] = (uint8_t)((tmp[src] << sh) | (tmp[src + 1] >> (8 - sh)));
    }
    if (pay % 8) bits_out[need - 1] &= (uint8_t)(0xFF << (8 - pay % 8));
    return BZH_OK;
    });
}

// ================================================================================================
// Whole path
// ================================================================================================
__device__ __forceinline__ void or_word_be(uint32_t *out, uint64_t word_idx, uint32_t v)
{
    if (v) atomicOr(out + word_idx, __builtin_bswap32(v));
}

// Copies nbits bits of src (MSB-first, from bit 0) to bit position dst_bit of dst (zeroed before).
// One thread per DESTINATION word: interior words are plain stores assembled from two source words
// (funnel shift); only the first and last word, shared with the neighbouring segments, are ORed.
__global__ void __launch_bounds__(256) concat_bits(uint32_t *dst, uint64_t dst_bit, const uint32_t *src, uint64_t nbits)
{
    const uint32_t sh = (uint32_t)(dst_bit & 31u);
    const uint64_t w0 = dst_bit >> 5;
    const uint64_t nsw = (nbits + 31) >> 5;            // source words
    const uint64_t ndw = (sh + nbits + 31) >> 5;       // destination words touched
    const uint32_t tailbits = (uint32_t)(nbits & 31u);
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < ndw; k += (uint64_t)gridDim.x * 256) {
        // destination word k holds source bits [32k - sh, 32k - sh + 32)
        uint32_t hi = 0, lo = 0; // source words k-1 and k (big-endian values)
        if (k >= 1 && k - 1 < nsw) {
            hi = __builtin_bswap32(src[k - 1]);
            if (k - 1 == nsw - 1 && tailbits) hi &= 0xFFFFFFFFu << (32 - tailbits);
        }
        if (k < nsw) {
            lo = __builtin_bswap32(src[k]);
            if (k == nsw - 1 && tailbits) lo &= 0xFFFFFFFFu << (32 - tailbits);
        }
        const uint32_t v = sh ? ((hi << (32 - sh)) | (lo >> sh)) : lo;
        if (k == 0 || k == ndw - 1)
            or_word_be(dst, w0 + k, v);
        else
            dst[w0 + k] = __builtin_bswap32(v);
    }
}

// Stream header "BZh"+level at bit 0 and footer + stream CRC at bit 32+body_bits (lib/lib.rs:18-22, :66-70).
__global__ void stream_frame(uint32_t *out, int level, uint64_t body_bits, uint32_t stream_crc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    or_word_be(out, 0, 0x425A6800u | (uint32_t)('0' + level));
    const uint32_t words[3] = {0x17724538u, 0x50900000u | (stream_crc >> 16), stream_crc << 16}; // 80 bits
    const uint64_t pos = 32 + body_bits;
    const uint32_t sh = (uint32_t)(pos & 31u);
    const uint64_t w0 = pos >> 5;
    for (int k = 0; k < 3; k++) {
        or_word_be(out, w0 + k, words[k] >> sh);
        if (sh) or_word_be(out, w0 + k + 1, words[k] << (32 - sh));
    }
}

static uint32_t fold_stream_crc(const uint32_t *crcs, size_t nb) // lib/lib.rs:107-108
{
    uint32_t s = 0;
    for (size_t k = 0; k < nb; k++) s = crcs[k] ^ ((s << 1) | (s >> 31));
    return s;
}

static float span_ms(hipEvent_t a, hipEvent_t b)
{
    float t = 0;
    return hipEventElapsedTime(&t, a, b) == hipSuccess ? t : 0.f;
}

// ---- lanes: two half-batch workers so that one half's latency-bound phases (tail rounds of the
// suffix sort, the Huffman heap, per-round read-backs) overlap the other half's bandwidth-bound kernels
static int ensure_lanes(bzh_ctx *ctx)
{
    if (!ctx->lanes.empty()) return BZH_OK;
    const uint32_t lane_mb = std::max<uint32_t>(1, ctx->max_batch / 2);
    Batch probe{};
    const size_t half = layout_batch(probe, nullptr, lane_mb, ctx->M);
    BZH_TRY(ensure_arena(ctx, ctx->max_batch, 2 * half)); // the lanes share the full-size arena, half each
    for (int k = 0; k < 2; k++) {
        bzh_ctx *l = new (std::nothrow) bzh_ctx();
        if (!l) return BZH_E_NOMEM;
        l->parent = ctx;
        l->device = ctx->device;
        l->level = ctx->level;
        l->M = ctx->M;
        l->max_batch = lane_mb;
        l->mode = ctx->mode;
        layout_batch(l->bt, ctx->arena + (size_t)k * half, lane_mb, ctx->M);
        l->S = l->bt.S;
        if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess ||
            hipHostMalloc((void **)&l->h_pinned, sizeof(uint32_t) * (lane_mb * 8 + 64 + (MAX_ROUNDS + 1) * SUMMARY_WORDS), hipHostMallocCoherent) != hipSuccess) {
            delete l;
            return BZH_E_NOMEM;
        }
        memset(l->h_pinned, 0, sizeof(uint32_t) * (lane_mb * 8 + 64 + (MAX_ROUNDS + 1) * SUMMARY_WORDS));
        ctx->lanes.push_back(l);
    }
    return BZH_OK;
}

struct RangeJob {
    size_t k0 = 0;
    uint32_t B = 0, nmax = 0, mmax = 0;
    uint64_t ntotal = 0, T = 0;
    int status = BZH_OK;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::promise<void> ready, packed;
};

// Everything of a batch up to its bit total, on the lane's stream and arena.
static int prepare_batch(bzh_ctx *lane, RangeJob &j, bool wait_total = true)
{
    hipStream_t st = lane->stream;
    auto mark = [&](int i) {
        if (lane->profiling) {
            j.ev[i] = bzh_event(lane);
            hipEventRecord(j.ev[i], st);
        }
    };
    lane->k_cur_ntotal = j.ntotal;
    mark(0);
    BZH_TRY(rle1_emit(lane, j.k0, j.B));
    mark(1);
    BZH_TRY(bwt_run(lane, j.B, j.nmax, j.ntotal));
    mark(2);
    BZH_TRY(mtf_run(lane, j.B, j.nmax, j.ntotal));
    mark(3);
    {   // the block headers carry the block CRCs: a whole-path plan left them to the owner's second stream (rle1_plan_split)
        bzh_ctx *pc = lane->parent ? lane->parent : lane;
        if (pc->crc_pending) HIP_TRY(lane, hipStreamWaitEvent(st, pc->plan_ev[1], 0));
    }
    BZH_TRY(huff_prepare(lane, j.B, j.mmax));
    mark(4);
    if (!wait_total) return BZH_OK; // (a call of one batch: the device carries on by itself, encode_range reads the total at the end)
    HIP_TRY(lane, hipMemcpyAsync(&j.T, lane->bt.bitoff + j.B, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(lane, bzh_stream_wait(st));
    return BZH_OK;
}

// Encodes plan blocks [b0, b1) into d_out starting at bit `bit_base`; words of d_out from
// bit_base/32 on are zeroed here as needed (words before that are the caller's).
// (`framed`: in/out -- the caller wants the whole stream's header and footer around these blocks; set back to false unless
// this call wrote them on the device: one batch, one lane, all blocks of the plan from bit 32 on)
static int encode_range(bzh_ctx *ctx, size_t b0, size_t b1, uint8_t *d_out, size_t cap, uint64_t bit_base,
                        uint64_t *nbits, const uint32_t *seed_word = nullptr, bool *framed = nullptr)
{
    const bool want_frame = framed && *framed;
    if (framed) *framed = false;
    if (((uintptr_t)d_out & 3u) != 0) {
        bzh_set_error(ctx, "output buffer must be 4-byte aligned");
        return BZH_E_ARG;
    }
    // default: one lane = this context (whole arena, caller's stream, no extra threads)
    std::vector<bzh_ctx *> lanes{ctx};
    if (ctx->nlanes == 2) {
        BZH_TRY(ensure_lanes(ctx));
        lanes = ctx->lanes;
    }
    const size_t NL = lanes.size();
    if (NL > 1) HIP_TRY(ctx, bzh_stream_wait(ctx->stream)); // the plan and whatever produced the input: the lanes' streams start behind it (one lane = this stream: in order anyway)
    const bzh_stats stats_in = ctx->stats;
    const size_t nb = b1 - b0;
    const uint32_t lane_mb = lanes[0]->max_batch;
    size_t njobs = (nb + lane_mb - 1) / lane_mb;
    if (njobs < NL && nb >= NL) njobs = NL; // give every lane work
    const size_t per = (nb + njobs - 1) / njobs;
    if (NL == 1) BZH_TRY(ensure_arena(ctx, (uint32_t)per));
    std::vector<std::unique_ptr<RangeJob>> jobs;
    for (size_t k0 = b0; k0 < b1; k0 += per) {
        auto j = std::make_unique<RangeJob>();
        j->k0 = k0;
        j->B = (uint32_t)std::min<size_t>(per, b1 - k0);
        for (uint32_t b = 0; b < j->B; b++) {
            const bzh_block &pb = ctx->plan_blocks[k0 + b];
            j->nmax = std::max(j->nmax, pb.rle_len);
            j->ntotal += pb.rle_len;
            ctx->stats.rle_bytes += pb.rle_len;
            ctx->stats.raw_bytes += pb.in_len;
        }
        j->mmax = j->nmax + 1; // m <= n + 1 (lib/mtf.rs:36)
        ctx->stats.blocks += j->B;
        jobs.push_back(std::move(j));
    }
    bzh_stats keep = stats_in; // what the counters were before this call touched them
    for (bzh_ctx *l : lanes) {
        l->profiling = ctx->profiling;
        l->sort_spans.clear();
        if (l != ctx) kstats_reset(l);
        if (l != ctx) {
            l->evnext = 0;
            memset(&l->stats, 0, sizeof l->stats);
            l->err[0] = 0;
        } else {
            ctx->stats.bwt_active_sum = 0;
            ctx->stats.bwt_rounds = 0;
            ctx->stats.bwt_sort_launches = 0;
            ctx->stats.bwt_sort_elems = 0;
        }
    }
    // lane k prepares jobs k, k+2, ...; it may reuse its arena only after the job was packed
    std::atomic<bool> abort{false}; // set when not every lane thread could be started: the ones that did start do nothing
    auto worker = [&](int k) {
        hipSetDevice(ctx->device);
        bool dead = false;
        for (size_t j = k; j < jobs.size(); j += NL) {
            RangeJob &job = *jobs[j];
            job.status = (dead || abort.load()) ? BZH_E_STATE : prepare_batch(lanes[k], job);
            if (job.status != BZH_OK) dead = true;
            job.ready.set_value();
            job.packed.get_future().wait();
        }
    };
    std::vector<std::thread> threads;
    if (NL > 1) {
        try {
            threads.reserve(NL);
            for (size_t k = 0; k < NL; k++) threads.emplace_back(worker, (int)k);
        } catch (...) { // no thread to be had: release the workers that did start (they skip their jobs), then report
            abort.store(true);
            for (auto &jp : jobs) jp->packed.set_value();
            for (auto &t : threads) t.join();
            for (bzh_ctx *l : lanes) hipStreamSynchronize(l->stream); // (a job may have been in flight already)
            ctx->stats = keep;
            bzh_set_error(ctx, "could not start the lane threads");
            return BZH_E_NOMEM;
        }
    }

    const uint64_t cap_words = cap / 4;
    uint64_t zeroed_upto = bit_base / 32; // first word not yet known to be zero
    uint64_t cur = 0;
    int status = BZH_OK;
    // A call of ONE batch (the usual case: up to max_batch blocks) needs the host for nothing between the Huffman tables and
    // the packed bits: the output words are zeroed, the capacity checked and the pack kernels gated on the device
    // (huff_pack_gate), and the bit total is read once, at the end -- instead of a copy, a wait, a memset and a launch in the
    // middle of the step (35-40 us of idle device).
    const bool one_batch = NL == 1 && jobs.size() == 1;
    for (size_t j = 0; j < jobs.size(); j++) {
        RangeJob &job = *jobs[j];
        bzh_ctx *lane = lanes[j % NL];
        if (NL == 1) { // no worker thread: prepare here
            job.status = status == BZH_OK ? prepare_batch(lane, job, !one_batch) : BZH_E_STATE;
            job.ready.set_value();
        }
        job.ready.get_future().wait();
        if (status == BZH_OK && job.status != BZH_OK) {
            status = job.status;
            if (lane != ctx) bzh_set_error(ctx, "%s", lane->err);
        }
        if (status == BZH_OK && one_batch) {
            hipStream_t st = lane->stream;
            uint64_t *rec = reinterpret_cast<uint64_t *>(lane->h_pinned); // (the first 64 words of the pinned block are free)
            const bool frame = want_frame && bit_base == 32 && b0 == 0 && !seed_word; // (block CRCs of the batch = of the stream)
            status = huff_pack_gate(lane, job.B, d_out, bit_base, cap_words, seed_word ? *seed_word : 0u, seed_word != nullptr, rec, frame ? 80u : 0u);
            if (status == BZH_OK) {
                status = huff_pack(lane, job.B, job.mmax, d_out, bit_base, true);
                if (status == BZH_OK && frame) {
                    status = huff_frame_stream(lane, job.B, d_out);
                    if (status == BZH_OK) *framed = true;
                }
                if (status != BZH_OK) (void)bzh_stream_wait(st); // (pack_gate is queued: nothing of this call may still run when the error is reported)
            }
            hipError_t he = hipSuccess;
            std::vector<uint32_t> hm;
            if (lane->profiling && status == BZH_OK) {
                job.ev[5] = bzh_event(lane);
                hipEventRecord(job.ev[5], st);
                hm.resize(job.B);
                he = hipMemcpyAsync(hm.data(), lane->bt.m, job.B * 4, hipMemcpyDeviceToHost, st);
            }
            if (status == BZH_OK) {
                if (he == hipSuccess) he = bzh_stream_wait(st);
                if (he != hipSuccess) {
                    bzh_set_error(ctx, "pack: %s", hipGetErrorString(he));
                    status = BZH_E_HIP;
                } else {
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                    job.T = reinterpret_cast<volatile uint64_t *>(rec)[0];
                    if (reinterpret_cast<volatile uint64_t *>(rec)[1] == 0) {
                        bzh_set_error(ctx, "output needs more than %zu bytes", cap);
                        status = BZH_E_CAP;
                    } else {
                        for (uint32_t b = 0; b < (uint32_t)hm.size(); b++) ctx->stats.mtf_syms += hm[b];
                        cur += job.T;
                    }
                }
            }
        } else
        if (status == BZH_OK) {
            hipStream_t st = lane->stream;
            const uint64_t need_upto = (bit_base + cur + job.T + 31) / 32 + 1;
            hipError_t he = hipSuccess;
            if (need_upto > cap_words) {
                bzh_set_error(ctx, "output needs more than %zu bytes", cap);
                status = BZH_E_CAP;
            } else {
                if (need_upto > zeroed_upto) {
                    he = hipMemsetAsync(d_out + zeroed_upto * 4, 0, (size_t)(need_upto - zeroed_upto) * 4, st);
                    if (he == hipSuccess && seed_word && zeroed_upto == bit_base / 32) // bits owed to the first word
                        he = hipMemcpyAsync(d_out + zeroed_upto * 4, seed_word, 4, hipMemcpyHostToDevice, st);
                    zeroed_upto = need_upto;
                }
                if (he == hipSuccess) status = huff_pack(lane, job.B, job.mmax, d_out, bit_base + cur);
                if (lane->profiling && status == BZH_OK) {
                    job.ev[5] = bzh_event(lane);
                    hipEventRecord(job.ev[5], st);
                    std::vector<uint32_t> hm(job.B);
                    he = hipMemcpyAsync(hm.data(), lane->bt.m, job.B * 4, hipMemcpyDeviceToHost, st);
                    if (he == hipSuccess) he = bzh_stream_wait(st);
                    for (uint32_t b = 0; b < job.B; b++) ctx->stats.mtf_syms += hm[b];
                }
                if (he == hipSuccess) he = bzh_stream_wait(st); // the lane's arena is free again
                if (he != hipSuccess) {
                    bzh_set_error(ctx, "pack: %s", hipGetErrorString(he));
                    status = BZH_E_HIP;
                }
                cur += job.T;
            }
        }
        job.packed.set_value();
    }
    for (auto &t : threads) t.join();
    if (status != BZH_OK) return status;
    if (ctx->profiling) {
        for (auto &jp : jobs) {
            RangeJob &job = *jp;
            ctx->stats.ms_rle1 += span_ms(job.ev[0], job.ev[1]);
            ctx->stats.ms_bwt += span_ms(job.ev[1], job.ev[2]);
            ctx->stats.ms_mtf += span_ms(job.ev[2], job.ev[3]);
            ctx->stats.ms_huff += span_ms(job.ev[3], job.ev[4]);
            ctx->stats.ms_pack += span_ms(job.ev[4], job.ev[5]);
        }
        for (bzh_ctx *l : lanes) {
            stats_collect_sort(l);
            if (l == ctx) continue;
            ctx->stats.ms_bwt_sort += l->stats.ms_bwt_sort;
            ctx->stats.bwt_sort_launches += l->stats.bwt_sort_launches;
            ctx->stats.bwt_sort_elems += l->stats.bwt_sort_elems;
            for (int k = 0; k < K_COUNT; k++) {
                ctx->k_ms[k] += l->k_ms[k];
                ctx->k_bytes[k] += l->k_bytes[k];
                ctx->k_launch[k] += l->k_launch[k];
            }
        }
    }
    for (bzh_ctx *l : lanes) {
        if (l == ctx) continue;
        ctx->stats.bwt_active_sum += l->stats.bwt_active_sum;
        ctx->stats.bwt_rounds = std::max(ctx->stats.bwt_rounds, l->stats.bwt_rounds);
    }
    ctx->stats.out_bits += cur;
    *nbits = cur;
    return BZH_OK;
}

static int check_in_ptr(bzh_ctx *ctx, const void *d_in)
{
    if (((uintptr_t)d_in & 15u) != 0) {
        bzh_set_error(ctx, "device input must be 16-byte aligned");
        return BZH_E_ARG;
    }
    return BZH_OK;
}

static int plan_device(bzh_ctx *ctx, const void *d_in, size_t n, size_t *nblocks, bool with_crc)
{
    if (ctx) stream_join(ctx);
    if (!ctx || (!d_in && n) || !nblocks) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BZH_TRY(check_in_ptr(ctx, d_in));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->profiling) {
        ctx->evnext = 0;
        e0 = bzh_event(ctx);
        hipEventRecord(e0, ctx->stream);
    }
    BZH_TRY(rle1_plan(ctx, (const uint8_t *)d_in, n, with_crc));
    if (ctx->profiling) {
        e1 = bzh_event(ctx);
        hipEventRecord(e1, ctx->stream);
        HIP_TRY(ctx, bzh_stream_wait(ctx->stream));
        memset(&ctx->stats, 0, sizeof ctx->stats);
        ctx->stats.ms_plan = span_ms(e0, e1);
    }
    *nblocks = ctx->plan_blocks.size();
    return BZH_OK;
}

extern "C" int bzh_plan_device(bzh_ctx *ctx, const void *d_in, size_t n, size_t *nblocks)
{
    return bzh_guard(ctx, [&]() -> int {
    return plan_device(ctx, d_in, n, nblocks, true);
    });
}

extern "C" int bzh_plan_device_nocrc(bzh_ctx *ctx, const void *d_in, size_t n, size_t *nblocks)
{
    return bzh_guard(ctx, [&]() -> int {
    return plan_device(ctx, d_in, n, nblocks, false);
    });
}

extern "C" int bzh_plan_tables_device(bzh_ctx *ctx, const void *d_in, size_t n)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (!d_in && n)) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BZH_TRY(check_in_ptr(ctx, d_in));
    if (ctx->profiling) {
        ctx->evnext = 0;
        kstats_reset(ctx);
    }
    return rle1_plan_tables(ctx, (const uint8_t *)d_in, n);
    });
}

extern "C" int bzh_plan_split_device(bzh_ctx *ctx, size_t start, size_t stop, int with_crc, size_t *nblocks)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !nblocks) return BZH_E_ARG;
    if (start > ctx->plan_n) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BZH_TRY(rle1_plan_split(ctx, start, with_crc != 0, stop));
    *nblocks = ctx->plan_blocks.size();
    return BZH_OK;
    });
}

extern "C" int bzh_plan_crc_range(bzh_ctx *ctx, size_t b0, size_t b1)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || b0 > b1) return BZH_E_ARG;
    if (b1 > ctx->plan_blocks.size()) return BZH_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rle1_plan_crc(ctx, b0, b1);
    });
}

extern "C" int bzh_plan_open(const bzh_ctx *ctx, uint8_t *out, size_t max_blocks)
{
    return bzh_guard(const_cast<bzh_ctx *>(ctx), [&]() -> int {
    if (!ctx || !out) return BZH_E_ARG;
    if (max_blocks < ctx->plan_open.size()) return BZH_E_CAP;
    for (size_t k = 0; k < ctx->plan_open.size(); k++) out[k] = ctx->plan_open[k];
    return BZH_OK;
    });
}

extern "C" int bzh_plan_blocks(const bzh_ctx *ctx, bzh_block *out, size_t max_blocks)
{
    return bzh_guard(const_cast<bzh_ctx *>(ctx), [&]() -> int {
    if (!ctx || !out) return BZH_E_ARG;
    if (max_blocks < ctx->plan_blocks.size()) return BZH_E_CAP;
    for (size_t k = 0; k < ctx->plan_blocks.size(); k++) out[k] = ctx->plan_blocks[k];
    return BZH_OK;
    });
}

extern "C" int bzh_encode_range_device(bzh_ctx *ctx, size_t b0, size_t b1, void *d_out, size_t cap,
                                       uint64_t *nbits)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !d_out || !nbits || b0 > b1) return BZH_E_ARG;
    if (b1 > ctx->plan_blocks.size()) return BZH_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const double keep_plan = ctx->stats.ms_plan;
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.ms_plan = keep_plan;
    ctx->sort_spans.clear();
    // the plan's kernel classes (bzh_plan_tables_device / bzh_plan_split_device recorded them) stay in the table: their
    // spans are turned into milliseconds now, before the event pool is rewound under them
    double keep_ms[2];
    uint64_t keep_bytes[2], keep_launch[2];
    if (!ctx->profiling) {
        kstats_reset(ctx);
    } else {
        HIP_TRY(ctx, bzh_stream_wait(ctx->stream));
        kstats_collect(ctx);
        const int cls[2] = {K_PLAN, K_CRC};
        for (int k = 0; k < 2; k++) {
            keep_ms[k] = ctx->k_ms[cls[k]];
            keep_bytes[k] = ctx->k_bytes[cls[k]];
            keep_launch[k] = ctx->k_launch[cls[k]];
        }
        kstats_reset(ctx);
        for (int k = 0; k < 2; k++) {
            ctx->k_ms[cls[k]] = keep_ms[k];
            ctx->k_bytes[cls[k]] = keep_bytes[k];
            ctx->k_launch[cls[k]] = keep_launch[k];
        }
    }
    ctx->evnext = 0;
    *nbits = 0;
    if (b0 == b1) return BZH_OK;
    BZH_TRY(rle1_plan_crc(ctx, b0, b1)); // no-op unless the plan left the CRCs to the encoder
    return encode_range(ctx, b0, b1, (uint8_t *)d_out, cap, 0, nbits);
    });
}

extern "C" int bzh_assemble_device(bzh_ctx *ctx, const void *const *d_segs, const uint64_t *seg_bits, size_t nseg,
                                   const uint32_t *block_crcs, size_t nblocks, void *d_out, size_t cap,
                                   size_t *out_len)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || !d_out || !out_len || (nseg && (!d_segs || !seg_bits)) || (nblocks && !block_crcs)) return BZH_E_ARG;
    if (((uintptr_t)d_out & 3u) != 0) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint64_t body = 0;
    for (size_t k = 0; k < nseg; k++) body += seg_bits[k];
    const uint64_t total_bits = 32 + body + 80;
    const size_t bytes = (size_t)((total_bits + 7) / 8);
    const size_t words = (size_t)((total_bits + 31) / 32) + 1;
    *out_len = bytes;
    if (words * 4 > cap) return BZH_E_CAP;
    HIP_TRY(ctx, hipMemsetAsync(d_out, 0, words * 4, st));
    uint64_t pos = 32;
    for (size_t k = 0; k < nseg; k++) {
        if (seg_bits[k] == 0) continue;
        if (((uintptr_t)d_segs[k] & 3u) != 0) return BZH_E_ARG;
        const uint64_t nw = (seg_bits[k] + 31) / 32 + 1;
        uint32_t grid = (uint32_t)std::min<uint64_t>((nw + 255) / 256, 8192);
        concat_bits<<<dim3(grid), 256, 0, st>>>((uint32_t *)d_out, pos, (const uint32_t *)d_segs[k], seg_bits[k]);
        pos += seg_bits[k];
    }
    stream_frame<<<1, 64, 0, st>>>((uint32_t *)d_out, ctx->level, body, fold_stream_crc(block_crcs, nblocks));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, bzh_stream_wait(st));
    return BZH_OK;
    });
}

extern "C" int bzh_encode_device(bzh_ctx *ctx, const void *d_in, size_t n, void *d_out, size_t cap, size_t *out_len,
                                 size_t *consumed)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (!d_in && n) || !d_out || !out_len) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (((uintptr_t)d_out & 3u) != 0 || cap < 16) return BZH_E_ARG;
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;
    ctx->evnext = 0;
    ctx->sort_spans.clear();
    kstats_reset(ctx);
    memset(&ctx->stats, 0, sizeof ctx->stats);
    if (ctx->profiling) {
        t0 = bzh_event(ctx);
        hipEventRecord(t0, st);
    }
    BZH_TRY(check_in_ptr(ctx, d_in));
    BZH_TRY(rle1_plan(ctx, (const uint8_t *)d_in, n, true, true)); // (the block CRCs beside the main stream: joined below)
    if (ctx->profiling) {
        t1 = bzh_event(ctx);
        hipEventRecord(t1, st);
    }
    // words 0 and 1 hold the stream header and the first body bits
    HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 4, st));
    uint64_t body = 0;
    const size_t nb = ctx->plan_blocks.size();
    bool framed = true; // (a stream of one batch gets its header and footer on the device, behind the pack: no host round trip)
    if (nb) {
        BZH_TRY(encode_range(ctx, 0, nb, (uint8_t *)d_out, cap, 32, &body, nullptr, &framed));
    } else {
        framed = false;
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 16, st));
    }
    const uint64_t total_bits = 32 + body + 80;
    const size_t bytes = (size_t)((total_bits + 7) / 8);
    *out_len = bytes;
    if (framed) { // (everything is written and waited for; the CRCs are collected for bzh_plan_blocks' sake)
        BZH_TRY(rle1_plan_crc_join(ctx));
        if (ctx->profiling) {
            t2 = bzh_event(ctx);
            hipEventRecord(t2, st);
            HIP_TRY(ctx, bzh_stream_wait(st));
            ctx->stats.ms_plan = span_ms(t0, t1);
            ctx->stats.ms_total = span_ms(t0, t2);
        }
        if (consumed) *consumed = n;
        return BZH_OK;
    }
    // the footer may reach one word past what encode_range zeroed
    const uint64_t zero_from = (32 + body + 31) / 32 + (nb ? 1 : 0), zero_to = (total_bits + 31) / 32 + 1;
    if (zero_to * 4 > cap) return BZH_E_CAP;
    if (zero_to > zero_from && nb)
        HIP_TRY(ctx, hipMemsetAsync((uint8_t *)d_out + zero_from * 4, 0, (size_t)(zero_to - zero_from) * 4, st));
    BZH_TRY(rle1_plan_crc_join(ctx));
    std::vector<uint32_t> crcs(nb);
    for (size_t k = 0; k < nb; k++) crcs[k] = ctx->plan_blocks[k].crc;
    stream_frame<<<1, 64, 0, st>>>((uint32_t *)d_out, ctx->level, body, fold_stream_crc(crcs.data(), nb));
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->profiling) {
        t2 = bzh_event(ctx);
        hipEventRecord(t2, st);
    }
    HIP_TRY(ctx, bzh_stream_wait(st));
    if (ctx->profiling) {
        ctx->stats.ms_plan = span_ms(t0, t1);
        ctx->stats.ms_total = span_ms(t0, t2);
    }
    if (consumed) *consumed = n;
    return BZH_OK;
    });
}

extern "C" int bzh_encode(bzh_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len,
                          size_t *consumed)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (!in && n) || !out || !out_len) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    BZH_TRY(ensure_stage(ctx, ctx->d_stage_in, ctx->stage_in_size, n + 16));
    const size_t dcap = n + n / 4 + (n / ((size_t)ctx->M * 4 / 5) + 2) * 4096 + 65536;
    BZH_TRY(ensure_stage(ctx, ctx->d_stage_out, ctx->stage_out_size, dcap));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage_in, in, n, hipMemcpyHostToDevice, st));
    size_t len = 0;
    BZH_TRY(bzh_encode_device(ctx, ctx->d_stage_in, n, ctx->d_stage_out, ctx->stage_out_size & ~(size_t)3, &len,
                              consumed));
    *out_len = len;
    if (len > cap) return BZH_E_CAP;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_stage_out, len, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, bzh_stream_wait(st));
    return BZH_OK;
    });
}

// ---- stage seams: RLE1 split and CRC -------------------------------------------------------------------------
extern "C" int bzh_rle1_split(bzh_ctx *ctx, const uint8_t *in, size_t n, bzh_block *blocks, size_t max_blocks,
                              size_t *nblocks, uint8_t *rle_out, size_t rle_cap)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (!in && n) || !blocks || !nblocks) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    stats_begin(ctx);
    BZH_TRY(ensure_stage(ctx, ctx->d_stage_in, ctx->stage_in_size, n + 16));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage_in, in, n, hipMemcpyHostToDevice, st));
    BZH_TRY(rle1_plan(ctx, ctx->d_stage_in, n));
    const size_t nb = ctx->plan_blocks.size();
    *nblocks = nb;
    if (nb > max_blocks) return BZH_E_CAP;
    for (size_t k = 0; k < nb; k++) blocks[k] = ctx->plan_blocks[k];
    if (rle_out) {
        size_t pos = 0;
        BZH_TRY(ensure_arena(ctx, (uint32_t)std::min<size_t>(nb, ctx->max_batch)));
        Batch &bt = ctx->bt;
        for (size_t k0 = 0; k0 < nb; k0 += ctx->max_batch) {
            const uint32_t B = (uint32_t)std::min<size_t>(ctx->max_batch, nb - k0);
            BZH_TRY(rle1_emit(ctx, k0, B));
            for (uint32_t b = 0; b < B; b++) {
                const size_t len = ctx->plan_blocks[k0 + b].rle_len;
                if (pos + len > rle_cap) return BZH_E_CAP;
                HIP_TRY(ctx, hipMemcpyAsync(rle_out + pos, bt.rle + (size_t)b * bt.S, len, hipMemcpyDeviceToHost, st));
                pos += len;
            }
            HIP_TRY(ctx, bzh_stream_wait(st));
        }
    }
    return BZH_OK;
    });
}

extern "C" int bzh_crc32(bzh_ctx *ctx, const uint8_t *in, size_t n, uint32_t *crc)
{
    return bzh_guard(ctx, [&]() -> int {
    if (ctx) stream_join(ctx);
    if (!ctx || (!in && n) || !crc) return BZH_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BZH_TRY(ensure_stage(ctx, ctx->d_stage_in, ctx->stage_in_size, n + 16));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage_in, in, n, hipMemcpyHostToDevice, ctx->stream));
    BZH_TRY(ensure_arena(ctx, 1)); // (crc_device borrows two words of it)
    return crc_device(ctx, ctx->d_stage_in, n, crc);
    });
}


// ================================================================================================
// Streaming
// ================================================================================================

static inline void put_be32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

// Waits for the pass in flight (if any); its results stay in strm.pass until stream_collect.
static void stream_join(bzh_ctx *ctx)
{
    auto &s = ctx->strm;
    if (s.worker.joinable()) s.worker.join();
}

extern "C" int bzh_stream_begin(bzh_ctx *ctx)
{
    return bzh_guard(ctx, [&]() -> int {
    if (!ctx) return BZH_E_ARG;
    auto &s = ctx->strm;
    stream_join(ctx);
    s.inflight = false;
    s.active = true;
    s.header_done = false;
    s.pending = 0;
    s.fill = 0;
    s.head = 0;
    s.bitpos = 0;
    s.carry_word = 0;
    s.stream_crc = 0;
    s.consumed = 0;
    return BZH_OK;
    });
}

extern "C" size_t bzh_stream_bound(const bzh_ctx *ctx, size_t n)
{
    if (!ctx) return 0;
    // everything not yet handed out may be released by this call: the pass in flight, the bytes waiting
    // for the next pass, n, and a carried tail (< 52 MB: one block of a maximal run), at worst-case
    // expansion, plus framing
    const auto &s = ctx->strm;
    const size_t raw = n + s.pending + (s.inflight ? s.pass.total : 0) + ((size_t)52 << 20);
    return raw + raw / 4 + (raw / 70000 + 4) * 4096 + 65536;
}

extern "C" size_t bzh_stream_consumed(const bzh_ctx *ctx) { return ctx ? ctx->strm.consumed : 0; }

extern "C" int bzh_stream_set_chunk(bzh_ctx *ctx, size_t bytes)
{
    return bzh_guard(ctx, [&]() -> int {
    if (!ctx || bytes == 0 || bytes > ((size_t)1 << 30)) return BZH_E_ARG;
    ctx->strm.min_feed = bytes;
    return BZH_OK;
    });
}

// Headroom kept in front of the fed bytes for the tail a pass leaves unconsumed (normally well below
// one block of raw input; larger tails -- a block inside one enormous run -- regrow the buffer).
static const size_t STREAM_HEAD = (size_t)4 << 20;

// Makes d_buf[fill] hold `head` bytes of headroom + the pending bytes + `extra` more, keeping the
// pending bytes.
static int stream_reserve(bzh_ctx *ctx, size_t head, size_t extra)
{
    auto &s = ctx->strm;
    const int f = s.fill;
    if (s.d_buf[f] && head <= s.head && s.head + s.pending + extra + 16 <= s.cap[f]) return BZH_OK;
    const size_t nhead = align_up(std::max(head, std::max(s.head, STREAM_HEAD)), 4096);
    const size_t want = align_up(nhead + std::max(s.pending + extra, s.min_feed) + ((size_t)16 << 20), 4096);
    uint8_t *nb = nullptr;
    if (hipMalloc((void **)&nb, want) != hipSuccess) {
        bzh_set_error(ctx, "hipMalloc(%zu) failed", want);
        return BZH_E_NOMEM;
    }
    if (s.pending) HIP_TRY(ctx, hipMemcpyAsync(nb + nhead, s.d_buf[f] + s.head, s.pending, hipMemcpyDeviceToDevice, s.copy_stream));
    HIP_TRY(ctx, bzh_stream_wait(s.copy_stream));
    if (s.d_buf[f]) hipFree(s.d_buf[f]);
    s.d_buf[f] = nb;
    s.cap[f] = want;
    s.head = nhead;
    return BZH_OK;
}

// The pass in flight, on its own thread: split, encode the blocks whose cut cannot move any more,
// bring their bits to pinned host memory.  Touches the context's plan / batch state and ctx->stream
// only; the feeding thread meanwhile uses the other buffer and the copy stream.
static void stream_pass(bzh_ctx *ctx)
{
    auto &s = ctx->strm;
    auto &p = s.pass;
    p.used = 0;
    p.nbits = 0;
    p.out_bytes = 0;
    p.lastw = 0;
    p.crcs.clear();
    p.rc = bzh_guard(ctx, [&]() -> int {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const uint8_t *buf = s.d_buf[p.buf] + p.off;
        BZH_TRY(rle1_plan(ctx, buf, p.total, true, true));
        const size_t nb = ctx->plan_blocks.size();
        size_t F = nb; // blocks that are final
        if (!p.eof) {
            F = 0;
            while (F < nb && !ctx->plan_open[F]) F++;
        }
        if (F == 0) return BZH_OK;
        p.used = F == nb ? p.total : (size_t)ctx->plan_blocks[F].in_off;
        size_t raw = 0;
        for (size_t k = 0; k < F; k++) raw += ctx->plan_blocks[k].in_len;
        const size_t dcap = (raw + raw / 4 + (F + 2) * 4096 + 65536) & ~(size_t)3;
        BZH_TRY(ensure_stage(ctx, s.d_out[p.obuf], s.d_out_cap[p.obuf], dcap));
        uint8_t *d_o = s.d_out[p.obuf];
        uint8_t seed_be[4];
        put_be32(seed_be, p.seed);
        uint32_t seed;
        memcpy(&seed, seed_be, 4);
        memset(&ctx->stats, 0, sizeof ctx->stats);
        ctx->sort_spans.clear();
        ctx->evnext = 0;
        kstats_reset(ctx);
        BZH_TRY(encode_range(ctx, 0, F, d_o, s.d_out_cap[p.obuf] & ~(size_t)3, p.phase, &p.nbits, p.phase ? &seed : nullptr));
        const uint64_t bits_in_buf = p.phase + p.nbits;
        const size_t full_words = (size_t)(bits_in_buf / 32);
        if (!s.h_out) {
            if (hipHostMalloc((void **)&s.h_out, 4096) != hipSuccess) {
                bzh_set_error(ctx, "hipHostMalloc(4096) failed");
                return BZH_E_NOMEM;
            }
            s.h_out_cap = 4096;
        }
        // the whole words stay on the device until the caller's next feed collects them; only the partial word behind
        // them (it seeds the next pass) comes back now
        if (bits_in_buf & 31u) HIP_TRY(ctx, hipMemcpyAsync(s.h_out, d_o + full_words * 4, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, bzh_stream_wait(st));
        p.out_bytes = full_words * 4;
        if (bits_in_buf & 31u) {
            const uint8_t *w = s.h_out;
            p.lastw = ((uint32_t)w[0] << 24) | ((uint32_t)w[1] << 16) | ((uint32_t)w[2] << 8) | w[3];
        }
        BZH_TRY(rle1_plan_crc_join(ctx));
        for (size_t k = 0; k < F; k++) p.crcs.push_back(ctx->plan_blocks[k].crc);
        return BZH_OK;
    });
}

extern "C" int bzh_stream_feed(bzh_ctx *ctx, const uint8_t *in, size_t n, int eof, uint8_t *out, size_t cap,
                               size_t *out_len)
{
    return bzh_guard(ctx, [&]() -> int {
    if (!ctx || (!in && n) || !out || !out_len) return BZH_E_ARG;
    auto &s = ctx->strm;
    if (!s.active) return BZH_E_STATE;
    *out_len = 0;
    const size_t PIECE = (size_t)256 << 20; // keeps every plan far inside 32-bit positions
    if (n > PIECE) {
        size_t done = 0, produced = 0;
        while (done < n) {
            const size_t k = n - done < PIECE ? n - done : PIECE;
            size_t got = 0;
            BZH_TRY(bzh_stream_feed(ctx, in + done, k, eof && done + k == n, out + produced, cap - produced, &got));
            done += k;
            produced += got;
        }
        *out_len = produced;
        return BZH_OK;
    }
    if (cap < bzh_stream_bound(ctx, n)) return BZH_E_CAP; // before anything is consumed: the call can be repeated
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!s.copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking));

    // 1. the fed bytes go straight to the device, behind what is already waiting there; this runs
    //    while the previous pass (if any) is still encoding
    if (n) {
        BZH_TRY(stream_reserve(ctx, s.head, n));
        HIP_TRY(ctx, hipMemcpyAsync(s.d_buf[s.fill] + s.head + s.pending, in, n, hipMemcpyHostToDevice, s.copy_stream));
        HIP_TRY(ctx, bzh_stream_wait(s.copy_stream)); // `in` belongs to the caller again on return
        s.pending += n;
    }
    if (!eof && s.pending < s.min_feed) return BZH_OK;

    size_t opos = 0;
    if (!s.header_done) { // lib/lib.rs:18-22
        out[0] = 0x42;
        out[1] = 0x5A;
        out[2] = 0x68;
        out[3] = (uint8_t)('0' + ctx->level);
        opos = 4;
        s.bitpos = 32;
        s.header_done = true;
    }
    // 2. take the results of the pass in flight: its bits are final now
    size_t left = 0;
    const uint8_t *tail = nullptr;
    struct Done {
        int obuf;
        size_t bytes;
    } done[2];
    int ndone = 0;
    // the finished passes' words: device -> the caller's buffer (called with the next pass already running)
    // (A failure here loses words whose bits are already counted in bitpos / carry_word / stream_crc: the stream cannot go
    // on -- later feeds get BZH_E_STATE, not a stream with a hole.)
    auto drain = [&]() -> int {
        const int rc = [&]() -> int {
            for (int k = 0; k < ndone; k++) {
                HIP_TRY(ctx, hipMemcpyAsync(out + opos, s.d_out[done[k].obuf], done[k].bytes, hipMemcpyDeviceToHost, s.copy_stream));
                opos += done[k].bytes;
            }
            if (ndone) HIP_TRY(ctx, bzh_stream_wait(s.copy_stream));
            return BZH_OK;
        }();
        ndone = 0;
        if (rc != BZH_OK) s.active = false;
        return rc;
    };
    auto collect = [&]() -> int {
        stream_join(ctx);
        s.inflight = false;
        const auto &p = s.pass;
        if (p.rc != BZH_OK) {
            s.active = false;
            return p.rc;
        }
        if (p.out_bytes) done[ndone++] = {p.obuf, p.out_bytes};
        if (p.nbits) {
            s.carry_word = p.lastw;
            s.bitpos += p.nbits;
        }
        for (uint32_t c : p.crcs) s.stream_crc = c ^ ((s.stream_crc << 1) | (s.stream_crc >> 31)); // lib/lib.rs:107-108
        s.consumed += p.used;
        left = p.total - p.used;
        tail = s.d_buf[p.buf] + p.off + p.used;
        return BZH_OK;
    };
    if (s.inflight) BZH_TRY(collect());

    // 3. start the next pass on [tail of the previous pass | fed bytes]
    for (;;) {
        const size_t total = left + s.pending;
        if (total == 0) break;
        BZH_TRY(stream_reserve(ctx, left, 0));
        if (left) {
            HIP_TRY(ctx, hipMemcpyAsync(s.d_buf[s.fill] + s.head - left, tail, left, hipMemcpyDeviceToDevice, s.copy_stream));
            HIP_TRY(ctx, bzh_stream_wait(s.copy_stream));
        }
        auto &p = s.pass;
        p.buf = s.fill;
        p.off = s.head - left;
        p.total = total;
        p.eof = eof != 0;
        p.phase = (uint32_t)(s.bitpos & 31u);
        p.seed = s.carry_word;
        p.obuf = s.osel;
        s.osel ^= 1;
        s.inflight = true;
        s.worker = std::thread(stream_pass, ctx);
        s.fill ^= 1; // the other buffer is free: its pass was collected above, its tail copied
        s.head = STREAM_HEAD;
        s.pending = 0;
        BZH_TRY(drain()); // (the pass before this one: its output buffer is the one the pass after this one will use)
        if (!eof) break;
        // 4. end of input: wait for this last pass too (it consumes everything it was given)
        left = 0;
        BZH_TRY(collect());
        if (left == 0) break; // always, at eof; the loop guards against a pass that could not finish its tail
    }
    BZH_TRY(drain());

    if (eof) { // footer + stream CRC (lib/lib.rs:66-70), zero padding to a byte (lib/out.rs:22-28)
        const uint32_t phase = (uint32_t)(s.bitpos & 31u);
        uint8_t tailb[24];
        memset(tailb, 0, sizeof tailb);
        put_be32(tailb, phase ? s.carry_word : 0u);
        const uint8_t foot[10] = {0x17, 0x72, 0x45, 0x38, 0x50, 0x90, (uint8_t)(s.stream_crc >> 24),
                                  (uint8_t)(s.stream_crc >> 16), (uint8_t)(s.stream_crc >> 8), (uint8_t)s.stream_crc};
        for (uint32_t k = 0; k < 80; k++) {
            const uint32_t bit = (foot[k >> 3] >> (7 - (k & 7))) & 1u;
            const uint32_t pos = phase + k;
            tailb[pos >> 3] |= (uint8_t)(bit << (7 - (pos & 7)));
        }
        const size_t nbytes = (phase + 80 + 7) / 8;
        memcpy(out + opos, tailb, nbytes);
        opos += nbytes;
        s.bitpos += 80;
        s.active = false;
        // (the two input buffers stay with the context for its next stream -- bzh_destroy frees them: a hipMalloc of 150 MB at
        // the first feed and two synchronizing hipFree at the end were 0.3-0.5 ms of every stream of the API path)
        s.head = 0;
    }
    *out_len = opos;
    return BZH_OK;
    });
}
