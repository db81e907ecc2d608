// multi.hip -- several MI355X behind ONE handle: banzai::encode(reader, writer, level) (reference lib/lib.rs:84-88) for a
// caller that holds a node, not a launcher.  The loop being sharded is lib/lib.rs:101-126: it carries only `raw`, the
// stream CRC, `consumed` and the bit cursor from one block to the next, so the blocks that start in a byte range can be
// cut and encoded by whoever holds that range (+ a look-ahead), given the offset its first block starts at.
//
// One host thread and one context per listed device (a device listed twice gets two contexts: how the whole flow is
// tested on a box with one GPU).  Worker r holds input[B_r, B_{r+1} + look-ahead), builds the split's tables over it while
// the start of its first block is on its way from worker r-1 -- a host variable behind a condition variable, not a network
// hop --, cuts from it until a block starts at or after B_{r+1}, hands that start on, encodes its blocks into a bit string
// from bit 0 and copies the string to device 0 (hipMemcpyPeerAsync: xGMI between devices of one node).  Device 0
// funnel-shifts the strings into the stream (bzh_assemble_device).  No data-path collective; the chain of splits is the
// only sequential part (about 0.3 us a block since round 5).  The ranges, the look-ahead and the slab heuristic are those
// of banzai_amd/sharded.py (the launcher flow: one process per GPU over torch.distributed), which stays.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr size_t LOOKAHEAD = (size_t)64 << 20; // bytes held beyond the own range (a block inside one enormous run eats < 52 MB)
constexpr double PLAN_COST = 0.002;            // a worker waits for the splits before it: ranges shrink by this much from one to the next
constexpr double ROOT_DISCOUNT = 0.96;         // worker 0 also receives the strings and assembles the stream

struct Worker {
    int device = 0;
    bzh_ctx *ctx = nullptr;
    uint8_t *d_in = nullptr;   // input[lo, lo + resident)
    size_t in_cap = 0;
    uint8_t *d_part = nullptr; // the worker's bit string (its slab)
    size_t part_cap = 0;
    uint8_t *d_seg = nullptr;  // where the string lands on device 0 (worker 0: its slab itself)
    size_t seg_cap = 0;
    // the call in flight
    size_t lo = 0, own_hi = 0, resident = 0;
    uint64_t nbits = 0;
    std::vector<uint32_t> crcs;
    int status = BZH_OK;
    std::string err;
    double ms_load = 0, ms_wait = 0, ms_plan = 0, ms_encode = 0, ms_copy = 0;
};

} // namespace

struct bzh_multi {
    int level = 9;
    std::vector<Worker> w;
    uint8_t *d_out = nullptr; // the assembled stream, on device 0
    size_t out_cap = 0, out_len = 0;
    size_t n = 0;             // bytes of the loaded input
    bool loaded = false;
    std::vector<size_t> bounds;
    size_t slab_override = 0; // test hook (bzh_multi_debug_slab): the slab size of the next runs instead of the heuristic
    char err[768] = {0};
    // the chain: start[r] = where worker r's first block begins (-1: a worker before it failed), ready[r]
    std::mutex mu;
    std::condition_variable cv;
    std::vector<long long> start;
    std::vector<char> ready;
};

namespace {

void set_err(bzh_multi *m, const char *fmt, const char *a = "", const char *b = "")
{
    snprintf(m->err, sizeof m->err, fmt, a, b);
}

std::vector<size_t> offsets(size_t n, int world) // range boundaries B_0..B_world: geometric lengths (sharded.offsets)
{
    std::vector<double> wt(world);
    const double q = 1.0 / (1.0 + PLAN_COST);
    double tot = 0;
    for (int r = 0; r < world; r++) {
        wt[r] = std::pow(q, r) * ((r == 0 && world > 1) ? ROOT_DISCOUNT : 1.0);
        tot += wt[r];
    }
    std::vector<size_t> out{0};
    double acc = 0;
    for (int r = 0; r + 1 < world; r++) {
        acc += wt[r];
        out.push_back((size_t)((double)n * acc / tot));
    }
    out.push_back(n);
    return out;
}

// Slab for a worker's bit string -- a BOUND, not an estimate: RLE1 expands a range by at most 5/4 (lib/rle.rs:210-234: five
// bytes for four), MTF + RLE2 leave at most one symbol per RLE1 byte + EOB (lib/mtf.rs:36), no code is longer than 17 bits
// (build_table_from_freqs rescales until that holds, lib/huffman.rs:293-296), a selector costs at most 6 bits per 50 symbols
// (one in the reference's mode), and a block's header, symbol map and coding tables fit 4,400 bytes (HDR_BYTES + the fixed
// mode's six tables are below it per table).  2.2 bytes per RLE1 byte covers 17 + 6/50 bits.  (Rounds 2-5 sized the slab at
// 5/4 of the range + 4 KiB a block -- what text needs five times over, but not a bound; the retry with a doubled slab stays
// behind it and is exercised through bzh_multi_debug_slab.)
size_t worst_case_slab(const std::vector<size_t> &b, int level)
{
    size_t rng = 0;
    for (size_t r = 0; r + 1 < b.size(); r++) rng = std::max(rng, b[r + 1] - b[r]);
    const size_t M = (size_t)(100000 * level - 1);
    const size_t blocks = rng / (M * 4 / 5) + 2;
    const size_t rle = rng + rng / 4 + blocks * 8; // RLE1 bytes of the range's blocks (the last block may start in the range and end behind it: + M)
    return ((rle + M) * 22 / 10 + blocks * 4400 + 65536 + 3) & ~(size_t)3;
}

int ensure_dev(uint8_t *&p, size_t &cap, size_t need)
{
    if (need <= cap) return BZH_OK;
    if (p) hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = (need + need / 8 + 4096 + 255) & ~(size_t)255;
    if (hipMalloc((void **)&p, want) != hipSuccess) return BZH_E_NOMEM;
    cap = want;
    return BZH_OK;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void fail(Worker &k, int status, const std::string &what)
{
    k.status = status;
    k.err = what;
}

// phase 1: the worker's byte range (+ look-ahead) to its device
void load_one(bzh_multi *m, int r, const uint8_t *in)
{
    Worker &k = m->w[r];
    k.status = BZH_OK;
    k.err.clear();
    const double t0 = now_ms();
    if (hipSetDevice(k.device) != hipSuccess) return fail(k, BZH_E_HIP, "hipSetDevice failed");
    k.lo = m->bounds[r];
    k.own_hi = m->bounds[r + 1];
    k.resident = std::min(m->n, k.own_hi + LOOKAHEAD) - k.lo;
    if (ensure_dev(k.d_in, k.in_cap, k.resident + 16) != BZH_OK) return fail(k, BZH_E_NOMEM, "hipMalloc for the input range failed");
    if (k.resident && hipMemcpy(k.d_in, in + k.lo, k.resident, hipMemcpyHostToDevice) != hipSuccess)
        return fail(k, BZH_E_HIP, "H2D copy of the input range failed");
    k.ms_load = now_ms() - t0;
}

// phase 2: tables, chain, split, encode, string to device 0
void run_one(bzh_multi *m, int r)
{
    Worker &k = m->w[r];
    const int W = (int)m->w.size();
    long long next = -1; // what worker r + 1 is told (-1: failed)
    bool published = false;
    auto publish = [&]() {
        if (published || r + 1 >= W) return;
        published = true;
        {
            std::lock_guard<std::mutex> g(m->mu);
            m->start[r + 1] = next;
            m->ready[r + 1] = 1;
        }
        m->cv.notify_all();
    };
    k.status = BZH_OK;
    k.err.clear();
    k.nbits = 0;
    k.crcs.clear();
    k.ms_wait = k.ms_plan = k.ms_encode = k.ms_copy = 0;
    auto ctx_err = [&](int st, const char *what) { fail(k, st, std::string(what) + ": " + bzh_strerror(st) + ": " + bzh_last_error(k.ctx)); };
    double t0 = now_ms();
    int st = hipSetDevice(k.device) == hipSuccess ? BZH_OK : BZH_E_HIP;
    if (st != BZH_OK) fail(k, st, "hipSetDevice failed");
    if (st == BZH_OK && k.resident) {
        st = bzh_plan_tables_device(k.ctx, k.d_in, k.resident);
        if (st != BZH_OK) ctx_err(st, "tables");
    }
    double t1 = now_ms();
    long long s0 = 0;
    if (r > 0) { // (always: the worker before always publishes, also when it failed)
        std::unique_lock<std::mutex> g(m->mu);
        m->cv.wait(g, [&] { return m->ready[r] != 0; });
        s0 = m->start[r];
    }
    double t2 = now_ms();
    k.ms_wait = t2 - t1;
    if (st == BZH_OK && s0 < 0) {
        st = BZH_E_STATE;
        fail(k, st, "a worker before this one failed");
    }
    size_t b1 = 0;
    std::vector<bzh_block> blocks;
    if (st == BZH_OK) {
        const size_t start = (size_t)s0;
        if (start >= k.own_hi || start >= m->n) { // a block of an earlier worker runs over this whole range
            next = (long long)start;
        } else if (start < k.lo) {
            st = BZH_E_STATE;
            fail(k, st, "the chain handed over a start in front of the resident bytes");
        } else {
            size_t nb = 0;
            st = bzh_plan_split_device(k.ctx, start - k.lo, k.own_hi - k.lo, 0, &nb);
            if (st != BZH_OK) ctx_err(st, "split");
            std::vector<uint8_t> open(nb);
            blocks.resize(nb);
            if (st == BZH_OK && nb) {
                st = bzh_plan_blocks(k.ctx, blocks.data(), nb);
                if (st == BZH_OK) st = bzh_plan_open(k.ctx, open.data(), nb);
                if (st != BZH_OK) ctx_err(st, "plan read-back");
            }
            if (st == BZH_OK) {
                while (b1 < nb && blocks[b1].in_off + k.lo < k.own_hi) b1++; // first block that starts at or after the range's end
                const bool sees_end = k.lo + k.resident >= m->n;
                // exact if the split saw the end of the input, or if a block starting at / after the range's end exists whose
                // predecessor's cut is final (sharded.own_blocks)
                if (!(sees_end || (b1 < nb && (b1 == 0 || !open[b1 - 1])))) {
                    st = BZH_E_STATE;
                    fail(k, st, "the look-ahead behind the worker's range does not settle its last cut");
                } else {
                    next = b1 < nb ? (long long)(blocks[b1].in_off + k.lo) : (long long)m->n;
                }
            }
        }
    }
    if (st != BZH_OK) next = -1;
    publish();
    double t3 = now_ms();
    k.ms_plan = (t1 - t0) + (t3 - t2);
    if (st == BZH_OK && b1) {
        for (int attempt = 0; attempt < 2; attempt++) {
            st = bzh_encode_range_device(k.ctx, 0, b1, k.d_part, k.part_cap & ~(size_t)3, &k.nbits);
            if (st != BZH_E_CAP || attempt) break;
            // a slab that turned out too small (the slab size is a heuristic): once more with twice the room
            const size_t want = k.part_cap * 2;
            if (ensure_dev(k.d_part, k.part_cap, want) != BZH_OK) {
                st = BZH_E_NOMEM;
                break;
            }
            if (r == 0) {
                k.d_seg = k.d_part;
                k.seg_cap = k.part_cap;
            }
        }
        if (st != BZH_OK) {
            ctx_err(st, "encode");
        } else {
            st = bzh_plan_blocks(k.ctx, blocks.data(), blocks.size()); // (the encode filled in the CRCs of its range)
            if (st != BZH_OK) ctx_err(st, "CRC read-back");
            for (size_t b = 0; b < b1 && st == BZH_OK; b++) k.crcs.push_back(blocks[b].crc);
        }
    }
    double t4 = now_ms();
    k.ms_encode = t4 - t3;
    // the string to device 0 (whole words + the one the assembly may read behind them)
    if (st == BZH_OK && r > 0 && k.nbits) {
        const size_t bytes = (size_t)((k.nbits + 31) / 32 + 1) * 4;
        if (bytes > k.seg_cap || bytes > k.part_cap) {
            st = BZH_E_CAP;
            fail(k, st, "the worker's bit string does not fit its landing buffer on device 0");
        } else {
            const hipError_t e = k.device == m->w[0].device
                                     ? hipMemcpy(k.d_seg, k.d_part, bytes, hipMemcpyDeviceToDevice)
                                     : hipMemcpyPeer(k.d_seg, m->w[0].device, k.d_part, k.device, bytes);
            if (e != hipSuccess) {
                st = BZH_E_HIP;
                fail(k, st, std::string("copy of the bit string to device 0: ") + hipGetErrorString(e));
            }
        }
    }
    k.ms_copy = now_ms() - t4;
    if (st != BZH_OK && k.status == BZH_OK) fail(k, st, "failed");
}

int first_error(bzh_multi *m, const char *phase)
{
    for (size_t r = 0; r < m->w.size(); r++)
        if (m->w[r].status != BZH_OK) {
            snprintf(m->err, sizeof m->err, "%s, worker %zu (device %d): %s", phase, r, m->w[r].device, m->w[r].err.c_str());
            // (a worker that only heard of another's failure is not the one to report)
            if (m->w[r].status != BZH_E_STATE || r + 1 == m->w.size()) return m->w[r].status;
            for (size_t q = 0; q < m->w.size(); q++)
                if (m->w[q].status != BZH_OK && m->w[q].status != BZH_E_STATE) {
                    snprintf(m->err, sizeof m->err, "%s, worker %zu (device %d): %s", phase, q, m->w[q].device, m->w[q].err.c_str());
                    return m->w[q].status;
                }
            return m->w[r].status;
        }
    return BZH_OK;
}

template <typename F>
int guard(bzh_multi *m, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        if (m) set_err(m, "out of host memory");
        return BZH_E_NOMEM;
    } catch (const std::exception &e) {
        if (m) set_err(m, "unexpected C++ exception: %s", e.what());
        return BZH_E_HIP;
    } catch (...) {
        if (m) set_err(m, "unexpected C++ exception");
        return BZH_E_HIP;
    }
}

} // namespace

extern "C" int bzh_create_multi(bzh_multi **out, const int *devices, int ndev, int level)
{
    if (!out) return BZH_E_ARG;
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64 || level < 1 || level > 9) return BZH_E_ARG;
    return guard(nullptr, [&]() -> int {
        bzh_multi *m = new bzh_multi();
        m->level = level;
        m->w.resize(ndev);
        int st = BZH_OK;
        for (int r = 0; r < ndev && st == BZH_OK; r++) {
            m->w[r].device = devices[r];
            st = bzh_create(&m->w[r].ctx, devices[r], level, 0);
        }
        if (st != BZH_OK) {
            bzh_destroy_multi(m);
            return st;
        }
        // the strings travel device to device: over xGMI where the devices can reach each other (else the runtime stages them)
        if (hipSetDevice(devices[0]) == hipSuccess)
            for (int r = 1; r < ndev; r++) {
                int can = 0;
                if (devices[r] != devices[0] && hipDeviceCanAccessPeer(&can, devices[0], devices[r]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(devices[r], 0); // ("already enabled" is fine)
            }
        (void)hipGetLastError();
        *out = m;
        return BZH_OK;
    });
}

extern "C" void bzh_destroy_multi(bzh_multi *m)
{
    if (!m) return;
    for (Worker &k : m->w) {
        if (!k.ctx) continue; // (a handle whose creation failed half way: nothing was ever done on that device)
        bzh_destroy(k.ctx);   // (waits for the context's own streams)
        if ((k.d_in || k.d_part) && hipSetDevice(k.device) == hipSuccess) {
            if (k.d_in) hipFree(k.d_in);
            if (k.d_part) hipFree(k.d_part);
        }
    }
    if (!m->w.empty() && m->w[0].ctx && hipSetDevice(m->w[0].device) == hipSuccess) {
        for (size_t r = 1; r < m->w.size(); r++)
            if (m->w[r].d_seg) hipFree(m->w[r].d_seg);
        if (m->d_out) hipFree(m->d_out);
    }
    (void)hipGetLastError(); // (no sticky status of this teardown is left for the thread's next HIP call to trip over)
    delete m;
}

extern "C" const char *bzh_multi_last_error(const bzh_multi *m) { return m ? m->err : "null handle"; }
extern "C" int bzh_multi_device_count(const bzh_multi *m) { return m ? (int)m->w.size() : 0; }

extern "C" int bzh_multi_load(bzh_multi *m, const uint8_t *in, size_t n)
{
    if (!m || (!in && n)) return BZH_E_ARG;
    return guard(m, [&]() -> int {
        m->err[0] = 0;
        m->loaded = false;
        m->n = n;
        const int W = (int)m->w.size();
        m->bounds = offsets(n, W);
        std::vector<std::thread> th;
        for (int r = 0; r < W; r++)
            th.emplace_back([m, r, in]() {
                try {
                    load_one(m, r, in);
                } catch (...) { // (nothing may unwind out of a thread: the failure is a status like any other)
                    fail(m->w[r], BZH_E_NOMEM, "exception in the worker (out of host memory?)");
                }
            });
        for (auto &t : th) t.join();
        const int st = first_error(m, "load");
        m->loaded = st == BZH_OK;
        return st;
    });
}

extern "C" int bzh_multi_run(bzh_multi *m, size_t *out_len)
{
    if (!m || !out_len) return BZH_E_ARG;
    return guard(m, [&]() -> int {
        *out_len = 0;
        m->err[0] = 0;
        if (!m->loaded) {
            set_err(m, "bzh_multi_run without a loaded input");
            return BZH_E_STATE;
        }
        const int W = (int)m->w.size();
        // slabs (on every worker's device) and their landing buffers (on device 0), sized before the workers start
        const size_t slab = m->slab_override ? (m->slab_override + 3) & ~(size_t)3 : worst_case_slab(m->bounds, m->level);
        for (int r = 0; r < W; r++) {
            Worker &k = m->w[r];
            if (m->slab_override && k.d_part) { // (the hook wants exactly this size: a buffer that only grows would hide it)
                if (hipSetDevice(k.device) == hipSuccess) hipFree(k.d_part);
                k.d_part = nullptr;
                k.part_cap = 0;
            }
            if (hipSetDevice(k.device) != hipSuccess || ensure_dev(k.d_part, k.part_cap, slab) != BZH_OK) {
                set_err(m, "hipMalloc for a worker's slab failed");
                return BZH_E_NOMEM;
            }
            if (m->slab_override) k.part_cap = slab; // (what the worker offers the encoder)
        }
        if (hipSetDevice(m->w[0].device) != hipSuccess) return BZH_E_HIP;
        m->w[0].d_seg = m->w[0].d_part;
        m->w[0].seg_cap = m->w[0].part_cap;
        for (int r = 1; r < W; r++)
            if (ensure_dev(m->w[r].d_seg, m->w[r].seg_cap, m->slab_override ? 2 * slab : slab) != BZH_OK) { // (the test hook's slab may be doubled)
                set_err(m, "hipMalloc for a landing buffer on device 0 failed");
                return BZH_E_NOMEM;
            }
        m->start.assign(W, 0);
        m->ready.assign(W, 0);
        m->ready[0] = 1;
        std::vector<std::thread> th;
        for (int r = 0; r < W; r++)
            th.emplace_back([m, r, W]() {
                try {
                    run_one(m, r);
                } catch (...) {
                    fail(m->w[r], BZH_E_NOMEM, "exception in the worker (out of host memory?)");
                    if (r + 1 < W) { // (the worker behind must not wait for ever: whatever was or was not published, it hears -1 now)
                        {
                            std::lock_guard<std::mutex> g(m->mu);
                            if (!m->ready[r + 1]) {
                                m->start[r + 1] = -1;
                                m->ready[r + 1] = 1;
                            }
                        }
                        m->cv.notify_all();
                    }
                }
            });
        for (auto &t : th) t.join();
        int st = first_error(m, "encode");
        if (st != BZH_OK) return st;
        // assembly on device 0: stream header, the strings funnel-shifted into place in worker order, footer
        std::vector<const void *> segs;
        std::vector<uint64_t> bits;
        std::vector<uint32_t> crcs;
        uint64_t body = 0;
        for (Worker &k : m->w) {
            segs.push_back(k.nbits ? k.d_seg : nullptr);
            bits.push_back(k.nbits);
            crcs.insert(crcs.end(), k.crcs.begin(), k.crcs.end());
            body += k.nbits;
        }
        if (hipSetDevice(m->w[0].device) != hipSuccess) return BZH_E_HIP;
        const size_t need = (size_t)((32 + body + 80 + 31) / 32 + 2) * 4;
        if (ensure_dev(m->d_out, m->out_cap, need) != BZH_OK) {
            set_err(m, "hipMalloc for the stream failed");
            return BZH_E_NOMEM;
        }
        size_t len = 0;
        st = bzh_assemble_device(m->w[0].ctx, segs.data(), bits.data(), segs.size(), crcs.data(), crcs.size(), m->d_out, m->out_cap & ~(size_t)3, &len);
        if (st != BZH_OK) {
            snprintf(m->err, sizeof m->err, "assembly: %s: %s", bzh_strerror(st), bzh_last_error(m->w[0].ctx));
            return st;
        }
        m->out_len = len;
        *out_len = len;
        return BZH_OK;
    });
}

extern "C" int bzh_multi_fetch(bzh_multi *m, uint8_t *out, size_t cap)
{
    if (!m || !out) return BZH_E_ARG;
    return guard(m, [&]() -> int {
        if (cap < m->out_len) return BZH_E_CAP;
        if (hipSetDevice(m->w[0].device) != hipSuccess) return BZH_E_HIP;
        if (m->out_len && hipMemcpy(out, m->d_out, m->out_len, hipMemcpyDeviceToHost) != hipSuccess) {
            set_err(m, "D2H copy of the stream failed");
            return BZH_E_HIP;
        }
        return BZH_OK;
    });
}

extern "C" const void *bzh_multi_output_device(const bzh_multi *m) { return m ? m->d_out : nullptr; }

// Test hook: the slab a worker encodes into is `bytes` instead of the heuristic's size (0: the heuristic again) -- a slab that is
// too small makes the worker's encode return BZH_E_CAP, which it answers ONCE with a slab of twice the size.
extern "C" int bzh_multi_debug_slab(bzh_multi *m, size_t bytes)
{
    if (!m) return BZH_E_ARG;
    m->slab_override = bytes;
    return BZH_OK;
}

extern "C" int bzh_multi_encode(bzh_multi *m, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, size_t *consumed)
{
    if (!m || (!in && n) || !out || !out_len) return BZH_E_ARG;
    int st = bzh_multi_load(m, in, n);
    if (st != BZH_OK) return st;
    size_t len = 0;
    st = bzh_multi_run(m, &len);
    if (st != BZH_OK) return st;
    *out_len = len;
    if (cap < len) return BZH_E_CAP;
    st = bzh_multi_fetch(m, out, cap);
    if (st == BZH_OK && consumed) *consumed = n;
    return st;
}

// per-worker wall clocks of the last bzh_multi_run (milliseconds): load, wait for the chain, tables + split, encode, copy
extern "C" int bzh_multi_times(const bzh_multi *m, double *out, size_t max_workers)
{
    if (!m || !out || max_workers < m->w.size()) return BZH_E_ARG;
    for (size_t r = 0; r < m->w.size(); r++) {
        const Worker &k = m->w[r];
        out[5 * r + 0] = k.ms_load;
        out[5 * r + 1] = k.ms_wait;
        out[5 * r + 2] = k.ms_plan;
        out[5 * r + 3] = k.ms_encode;
        out[5 * r + 4] = k.ms_copy;
    }
    return BZH_OK;
}
