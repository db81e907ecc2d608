/*
 * banzai_oracle.c -- CPU restatement of jgbyrne/banzai v0.3.1's bzip2 encode path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libbzhip.so) never links,
 * loads or calls anything in this directory.
 *
 * Parity status: PINNED against (i) the three known-answer tests the reference holds
 * (lib/bwt.rs:758-772, lib/mtf.rs:139-158, lib/out.rs:107-133), (ii) fixtures produced
 * in the build container by running the reference's own debug/bwt.py and debug/rle1.py
 * (tests/golden/, generator script committed), (iii) libbz2 1.0.8 as decoder of every
 * stream (the reference's fuzz/fuzz_targets/round_trip.rs check).  The Rust crate itself
 * cannot be compiled here (no rustc), so Huffman code lengths and RLE1 cut points are
 * pinned by the source text only; see DESIGN.md.
 *
 * Every function cites the reference file:line it follows.  Single-threaded, like the
 * reference.  Plain C99, no dependencies.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * Bit sink -- lib/out.rs:7-105 (OutputStream: strand byte + strand_bits, MSB first)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *buf;
    size_t cap;
    size_t len; /* bytes that WOULD have been written (may exceed cap) */
    uint8_t strand;
    unsigned strand_bits;
} orc_sink;

static void sink_put(orc_sink *o, uint8_t byte)
{
    if (o->len < o->cap)
        o->buf[o->len] = byte;
    o->len++;
}

/* lib/out.rs:31-55 */
static void sink_write_bits(orc_sink *o, uint8_t chunk, unsigned num_bits)
{
    unsigned rptr = o->strand_bits + num_bits;
    if (rptr < 8) {
        o->strand |= (uint8_t)(chunk << (8 - rptr));
        o->strand_bits = rptr;
    } else if (rptr == 8) {
        sink_put(o, o->strand | chunk);
        o->strand = 0;
        o->strand_bits = 0;
    } else {
        unsigned spill = rptr - 8;
        sink_put(o, o->strand | (uint8_t)(chunk >> spill));
        o->strand = (uint8_t)(chunk << (8 - spill));
        o->strand_bits = spill;
    }
}

/* lib/out.rs:79-81 */
static void sink_write_byte(orc_sink *o, uint8_t byte) { sink_write_bits(o, byte, 8); }

/* lib/out.rs:58-76 -- big-endian bytes of chunk, partial top byte first */
static void sink_write_bits_u32(orc_sink *o, uint32_t chunk, unsigned num_bits)
{
    uint8_t be[4] = { (uint8_t)(chunk >> 24), (uint8_t)(chunk >> 16), (uint8_t)(chunk >> 8),
                      (uint8_t)chunk };
    unsigned full = num_bits / 8, rem = num_bits % 8;
    unsigned bptr = 3 - full;
    if (rem != 0)
        sink_write_bits(o, be[bptr], rem);
    bptr += 1;
    while (bptr < 4) {
        sink_write_byte(o, be[bptr]);
        bptr += 1;
    }
}

/* lib/out.rs:84-104 */
static void sink_write_bytes(orc_sink *o, const uint8_t *bytes, size_t n)
{
    if (o->strand_bits == 0) {
        for (size_t k = 0; k < n; k++)
            sink_put(o, bytes[k]);
    } else {
        unsigned rshift = o->strand_bits, lshift = 8 - o->strand_bits;
        uint8_t strand = o->strand;
        for (size_t k = 0; k < n; k++) {
            sink_put(o, (uint8_t)(bytes[k] >> rshift) | strand);
            strand = (uint8_t)(bytes[k] << lshift);
        }
        o->strand = strand;
    }
}

/* lib/out.rs:22-28 */
static void sink_close(orc_sink *o)
{
    if (o->strand_bits != 0)
        sink_put(o, o->strand);
}

/* Exposed for the lib/out.rs:107-133 known-answer test.  ops: (kind, value, nbits) triples,
 * kind 0 = write_bits, 1 = write_byte, 2 = write_bits_u32; kind 3 = write_bytes of the
 * next `value` bytes taken from `blob` (consumed left to right). */
ORC_API size_t orc_bitsink_run(const uint32_t *ops, size_t nops, const uint8_t *blob,
                               uint8_t *out, size_t cap)
{
    orc_sink o = { out, cap, 0, 0, 0 };
    size_t bpos = 0;
    for (size_t k = 0; k < nops; k++) {
        uint32_t kind = ops[3 * k], val = ops[3 * k + 1], nb = ops[3 * k + 2];
        if (kind == 0)
            sink_write_bits(&o, (uint8_t)val, nb);
        else if (kind == 1)
            sink_write_byte(&o, (uint8_t)val);
        else if (kind == 2)
            sink_write_bits_u32(&o, val, nb);
        else {
            sink_write_bytes(&o, blob + bpos, val);
            bpos += val;
        }
    }
    sink_close(&o);
    return o.len;
}

/* ------------------------------------------------------------------------------------
 * Block CRC -- lib/crc32.rs:31-48.  The reference bit-reverses every byte, runs
 * crc 3.0.0's CRC_32_ISO_HDLC (reflected 0xEDB88320, init/xorout 0xFFFFFFFF -- the
 * published zlib CRC; crate not vendored, Cargo.lock:20-30) and bit-reverses the result.
 * Restated literally (without mutating the caller's buffer).
 * ---------------------------------------------------------------------------------- */
static uint32_t hdlc_table[256];
static int hdlc_ready = 0;

static void hdlc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        hdlc_table[i] = c;
    }
    hdlc_ready = 1;
}

static uint8_t rev8(uint8_t b)
{
    uint8_t r = 0;
    for (int k = 0; k < 8; k++)
        r |= (uint8_t)(((b >> k) & 1) << (7 - k));
    return r;
}

ORC_API uint32_t orc_crc32(const uint8_t *buf, size_t n)
{
    if (!hdlc_ready)
        hdlc_init();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t k = 0; k < n; k++)
        c = hdlc_table[(c ^ rev8(buf[k])) & 0xFF] ^ (c >> 8);
    c ^= 0xFFFFFFFFu;
    uint32_t sum = 0; /* lib/crc32.rs:41-45 */
    for (int i = 0; i < 32; i++) {
        sum <<= 1;
        sum |= (c >> i) & 1;
    }
    return sum;
}

/* ------------------------------------------------------------------------------------
 * RLE1 + block splitter -- lib/rle.rs:102-253, for a reader that hands over the whole
 * remaining input in one fill_buf (the in-memory slice case; SURVEY T16 cannot occur).
 * `raw[0..n)` is everything not yet encoded.  Writes at most 100000*level-1 bytes to
 * `out`, returns the number of raw bytes consumed by this block.
 * ---------------------------------------------------------------------------------- */
ORC_API size_t orc_rle_one(const uint8_t *raw, size_t n, int level, uint8_t *out,
                           size_t *out_len, uint32_t *chk)
{
    *out_len = 0;
    *chk = 0;
    if (n == 0) /* lib/rle.rs:111-118 */
        return 0;

    size_t bound = (size_t)100000 * (size_t)level - 1; /* lib/rle.rs:121 */
    size_t olen = 0;
#define PUSH(x) do { out[olen++] = (x); bound--; } while (0)

    size_t floor = 0, i = 0;
    uint8_t b = raw[0];

    for (;;) {
        /* lib/rle.rs:136-151 */
        if (bound == 0)
            break;
        if (bound == 1) {
            PUSH(b);
            i += 1;
            break;
        }
        PUSH(b);

        /* lib/rle.rs:153-165; margin_call (:58-91) on a fully buffered input = min(n-i,256) */
        size_t avail = n - i;
        if (avail == 1) {
            i += 1;
            break;
        }
        if (avail == 2) {
            PUSH(raw[i + 1]);
            i += 2;
            break;
        }

        uint8_t hop = raw[i + 2];
        PUSH(raw[i + 1]); /* lib/rle.rs:169 */

        if (b == hop && b == raw[i + 1]) { /* lib/rle.rs:172 */
            int run = 0;
            if (i > floor && b == raw[i - 1]) { /* lib/rle.rs:177-186 */
                if (bound < 2) {
                    i += 2;
                    break;
                }
                PUSH(hop);
                i += 3;
                run = 1;
            }
            if (!run && i + 3 < n) { /* lib/rle.rs:189-208 */
                uint8_t step = raw[i + 3];
                if (b == step) {
                    if (bound == 0) {
                        i += 2;
                        break;
                    }
                    PUSH(hop);
                    if (bound < 2) {
                        i += 3;
                        break;
                    }
                    PUSH(step);
                    i += 4;
                    run = 1;
                }
            }
            if (run) { /* lib/rle.rs:210-234 */
                uint8_t rep = 0;
                while (rep < 251 && i < n && raw[i] == b) {
                    rep++;
                    i++;
                }
                PUSH(rep);
                floor = i;
                if (i >= n)
                    break;
                b = raw[i];
                continue;
            }
        }
        i += 2; /* lib/rle.rs:237-239 */
        b = hop;
    }
#undef PUSH
    *out_len = olen;
    *chk = orc_crc32(raw, i); /* lib/rle.rs:242-244: CRC over the consumed raw bytes */
    return i;
}

/* ------------------------------------------------------------------------------------
 * BWT by SA-IS on the doubled block -- lib/bwt.rs:115-756.
 * Idx = i32 (lib/bwt.rs:7).  !x is bitwise NOT, used as a "do not induce" mark.
 * ---------------------------------------------------------------------------------- */
typedef int32_t idx_t;

typedef struct { /* lib/bwt.rs:115-119 */
    uint32_t *sigma;
    size_t nsigma;
    uint32_t *sizes;
    uint32_t *bptrs;
    size_t cap;
} buckets_t;

static void buckets_heads(buckets_t *bk) /* lib/bwt.rs:122-128 */
{
    uint32_t acc = 0;
    for (size_t k = 0; k < bk->nsigma; k++) {
        uint32_t w = bk->sigma[k];
        bk->bptrs[w] = acc;
        acc += bk->sizes[w];
    }
}

static void buckets_tails(buckets_t *bk) /* lib/bwt.rs:130-136 */
{
    uint32_t acc = 0;
    for (size_t k = 0; k < bk->nsigma; k++) {
        uint32_t w = bk->sigma[k];
        acc += bk->sizes[w];
        bk->bptrs[w] = acc - 1;
    }
}

#define DATA_AT(data, wbytes, k) \
    ((wbytes) == 1 ? (uint32_t)((const uint8_t *)(data))[k] : ((const uint32_t *)(data))[k])

/* lib/bwt.rs:138-149 (layout) + :151-173 (build / rebuild): sigma is produced in
 * ascending order by the enumeration, the reference's sort_unstable is then a no-op. */
static void buckets_layout(buckets_t *bk, const void *data, int wbytes, size_t n, size_t max_sigma)
{
    if (max_sigma > bk->cap) {
        bk->sigma = (uint32_t *)realloc(bk->sigma, max_sigma * sizeof(uint32_t));
        bk->sizes = (uint32_t *)realloc(bk->sizes, max_sigma * sizeof(uint32_t));
        bk->bptrs = (uint32_t *)realloc(bk->bptrs, max_sigma * sizeof(uint32_t));
        bk->cap = max_sigma;
    }
    memset(bk->sizes, 0, max_sigma * sizeof(uint32_t));
    memset(bk->bptrs, 0, max_sigma * sizeof(uint32_t));
    bk->nsigma = 0;
    for (size_t k = 0; k < n; k++)
        bk->sizes[DATA_AT(data, wbytes, k)]++;
    for (size_t w = 0; w < max_sigma; w++)
        if (bk->sizes[w] > 0)
            bk->sigma[bk->nsigma++] = (uint32_t)w;
}

static void buckets_free(buckets_t *bk)
{
    free(bk->sigma);
    free(bk->sizes);
    free(bk->bptrs);
}

/* lib/bwt.rs:176-192 */
#define TAIL_PUSH(sa, bk, w, v) do { uint32_t *bp_ = &(bk)->bptrs[w]; (sa)[*bp_] = (v); *bp_ -= 1u; } while (0)
#define HEAD_PUSH(sa, bk, w, v) do { uint32_t *bp_ = &(bk)->bptrs[w]; (sa)[*bp_] = (v); *bp_ += 1u; } while (0)

/* lib/bwt.rs:199-236 */
static void induced_sort_fwd(const void *data, int wb, idx_t *sa, size_t n, buckets_t *bk, int wipe)
{
    buckets_heads(bk);
    idx_t i = (idx_t)n, i_sup = i - 1, i_sup2 = i - 2;
    idx_t push = DATA_AT(data, wb, i_sup2) < DATA_AT(data, wb, i_sup) ? ~i_sup : i_sup;
    HEAD_PUSH(sa, bk, DATA_AT(data, wb, i_sup), push);
    for (size_t p = 0; p < n; p++) {
        i = sa[p];
        if (i > 0) {
            i_sup = i - 1;
            i_sup2 = i - 2;
            uint32_t c1 = DATA_AT(data, wb, i_sup);
            push = (i_sup2 < 0 || DATA_AT(data, wb, i_sup2) < c1) ? ~i_sup : i_sup;
            HEAD_PUSH(sa, bk, c1, push);
            sa[p] = wipe ? 0 : ~sa[p];
        } else if (i < 0) {
            sa[p] = ~sa[p];
        }
    }
}

/* lib/bwt.rs:238-271 */
static void induced_sort_bck(const void *data, int wb, idx_t *sa, size_t n, buckets_t *bk, int wipe,
                             int unflip)
{
    buckets_tails(bk);
    for (size_t p = n; p-- > 0;) {
        idx_t i = sa[p];
        if (i > 0) {
            idx_t i_sup = i - 1, i_sup2 = i - 2;
            uint32_t c1 = DATA_AT(data, wb, i_sup);
            idx_t push = (i_sup2 < 0 || DATA_AT(data, wb, i_sup2) > c1) ? ~i_sup : i_sup;
            TAIL_PUSH(sa, bk, c1, push);
            if (wipe)
                sa[p] = 0;
        } else if (unflip && i < 0) {
            sa[p] = ~sa[p];
        }
    }
}

static int substrings_equal(const void *data, int wb, size_t a, size_t b, size_t len) /* :91-93 */
{
    if (wb == 1)
        return memcmp((const uint8_t *)data + a, (const uint8_t *)data + b, len) == 0;
    return memcmp((const uint32_t *)data + a, (const uint32_t *)data + b, len * 4) == 0;
}

/* lib/bwt.rs:273-377.  Returns lms_count, *names = number of distinct LMS-substring names. */
static size_t encode_reduced(const void *data, int wb, idx_t *sa, size_t n, size_t *names)
{
#define LOOKUP(cnt, li) ((cnt) + (size_t)((li) >> 1))
    size_t lms_count = 0;
    for (size_t p = 0; p < n; p++) {
        if (sa[p] < ~0) { /* marked entries other than !0 */
            sa[lms_count] = ~sa[p];
            lms_count++;
        }
    }
    for (size_t p = lms_count; p < n; p++)
        sa[p] = INT32_MAX;

    /* right-to-left typing with a phantom sentinel; record LMS-substring lengths */
    {
        idx_t i_sub = (idx_t)n;
        int ty_sub_is_S = 0;
        uint32_t w_sub = DATA_AT(data, wb, n - 1);
        size_t unseen = lms_count;
        idx_t last_lms = i_sub - 1;
        for (size_t q = n - 1; q-- > 0;) {
            uint32_t w = DATA_AT(data, wb, q);
            i_sub -= 1;
            if (!ty_sub_is_S) {
                if (w < w_sub)
                    ty_sub_is_S = 1;
            } else if (w > w_sub) {
                sa[LOOKUP(lms_count, i_sub)] = (1 + last_lms) - i_sub;
                last_lms = i_sub;
                unseen -= 1;
                if (unseen == 0)
                    break;
                ty_sub_is_S = 0;
            }
            w_sub = w;
        }
    }

    /* name LMS substrings in sorted order (lib/bwt.rs:334-365) */
    uint32_t rword = 0;
    size_t prv = 0, prv_len = 0;
    for (size_t k = 0; k < lms_count; k++) {
        idx_t cur = sa[k];
        size_t look = LOOKUP(lms_count, cur);
        size_t cur_len = (size_t)sa[look];
        int eq = 0;
        if (prv != 0 && prv_len == cur_len && prv_len + cur_len < n)
            eq = substrings_equal(data, wb, prv, (size_t)cur, prv_len);
        if (!eq) {
            if (prv != 0)
                rword += 1;
            prv = (size_t)cur;
            prv_len = cur_len;
        }
        sa[look] = (idx_t)rword;
    }

    /* compact names to the array tail (lib/bwt.rs:368-374) */
    size_t wp = n - 1;
    for (size_t p = n; p-- > lms_count;) {
        if (sa[p] != INT32_MAX) {
            sa[wp] = sa[p];
            wp--;
        }
    }
    *names = (size_t)rword + 1;
    return lms_count;
#undef LOOKUP
}

/* lib/bwt.rs:379-421 */
static void decode_reduced(const void *data, int wb, idx_t *sa, size_t n, size_t lms_count)
{
    size_t wp = n - 1;
    idx_t i_sub = (idx_t)n;
    int ty_sub_is_S = 0;
    uint32_t w_sub = DATA_AT(data, wb, n - 1);
    for (size_t q = n - 1; q-- > 0;) {
        uint32_t w = DATA_AT(data, wb, q);
        i_sub -= 1;
        if (!ty_sub_is_S) {
            if (w < w_sub)
                ty_sub_is_S = 1;
        } else if (w > w_sub) {
            sa[wp] = i_sub;
            wp--;
            ty_sub_is_S = 0;
        }
        w_sub = w;
    }
    for (size_t p = 0; p < lms_count; p++)
        sa[p] = sa[n - lms_count + (size_t)sa[p]];
    for (size_t p = lms_count; p < n; p++)
        sa[p] = 0;
}

/* Shared first step: push LMS suffixes into bucket tails, right to left.
 * lib/bwt.rs:435-464 (sais) and :577-606 (bwt).  Optionally records has_byte. */
static size_t bucket_lms(const void *data, int wb, idx_t *sa, size_t n, buckets_t *bk, uint8_t *has_byte)
{
    size_t lms_count = 0;
    buckets_tails(bk);
    idx_t i_sub = (idx_t)n;
    int ty_sub_is_S = 0;
    uint32_t w_sub = DATA_AT(data, wb, n - 1);
    if (has_byte)
        has_byte[w_sub] = 1;
    for (size_t q = n - 1; q-- > 0;) {
        uint32_t w = DATA_AT(data, wb, q);
        if (has_byte)
            has_byte[w] = 1;
        i_sub -= 1;
        if (!ty_sub_is_S) {
            if (w < w_sub)
                ty_sub_is_S = 1;
        } else if (w > w_sub) {
            TAIL_PUSH(sa, bk, w_sub, i_sub);
            lms_count++;
            ty_sub_is_S = 0;
        }
        w_sub = w;
    }
    return lms_count;
}

/* lib/bwt.rs:423-518.  `data` is a u32 string of length n stored in the tail of the
 * caller's array (Array::split, :20-30); `sa` has n usable entries, zero-filled. */
static void sais_u32(size_t sigma_size, uint32_t *data, idx_t *sa, size_t n, buckets_t *bk)
{
    size_t lms_count = bucket_lms(data, 4, sa, n, bk, NULL);
    if (lms_count > 1) {
        induced_sort_fwd(data, 4, sa, n, bk, 1);
        induced_sort_bck(data, 4, sa, n, bk, 1, 0);
        size_t names;
        lms_count = encode_reduced(data, 4, sa, n, &names);
        if (names != lms_count) {
            /* Array::split: sa region = everything before the last lms_count entries, zeroed */
            for (size_t p = 0; p < n - lms_count; p++)
                sa[p] = 0;
            uint32_t *rdata = (uint32_t *)(sa + (n - lms_count));
            buckets_layout(bk, rdata, 4, lms_count, names); /* rebuild, :484 */
            sais_u32(names, rdata, sa, lms_count, bk);
        } else {
            for (size_t p = 0; p < lms_count; p++) {
                size_t w_rank = (size_t)sa[n - lms_count + p];
                sa[w_rank] = (idx_t)p;
            }
        }
        decode_reduced(data, 4, sa, n, lms_count);
        buckets_layout(bk, data, 4, n, sigma_size); /* :499 */
        buckets_tails(bk);
        for (size_t p = lms_count; p-- > 0;) {
            idx_t li = sa[p];
            sa[p] = 0;
            TAIL_PUSH(sa, bk, data[li], li);
        }
    }
    induced_sort_fwd(data, 4, sa, n, bk, 0);
    induced_sort_bck(data, 4, sa, n, bk, 0, 1);
}

/* lib/bwt.rs:526-756.  bwt_out must hold n bytes; has_byte 256 bytes (0/1).
 * Returns ptr (SIZE_MAX for the n == 0 / oversize early-outs, :535-562). */
ORC_API size_t orc_bwt(const uint8_t *input, size_t n, uint8_t *bwt_out, uint8_t *has_byte)
{
    memset(has_byte, 0, 256);
    if (n == 0)
        return (size_t)-1;
    if (n == 1) {
        has_byte[input[0]] = 1;
        bwt_out[0] = input[0];
        return 0;
    }
    if (n >= (size_t)(INT32_MAX / 4) - 1)
        return (size_t)-1;

    size_t buf_n = n * 2;
    uint8_t *data = (uint8_t *)malloc(buf_n);
    memcpy(data, input, n);
    memcpy(data + n, input, n); /* :566-567 */
    idx_t *sa = (idx_t *)calloc(buf_n, sizeof(idx_t));

    buckets_t bk = { 0, 0, 0, 0, 0 };
    buckets_layout(&bk, data, 1, buf_n, 256);

    size_t lms_count = bucket_lms(data, 1, sa, buf_n, &bk, has_byte);

    if (lms_count > 1) {
        induced_sort_fwd(data, 1, sa, buf_n, &bk, 1);
        induced_sort_bck(data, 1, sa, buf_n, &bk, 1, 0);
        size_t names;
        lms_count = encode_reduced(data, 1, sa, buf_n, &names);
        if (names != lms_count) {
            for (size_t p = 0; p < buf_n - lms_count; p++)
                sa[p] = 0;
            uint32_t *rdata = (uint32_t *)(sa + (buf_n - lms_count));
            buckets_t rbk = { 0, 0, 0, 0, 0 };
            buckets_layout(&rbk, rdata, 4, lms_count, names);
            sais_u32(names, rdata, sa, lms_count, &rbk);
            buckets_free(&rbk);
        } else {
            for (size_t p = 0; p < lms_count; p++) {
                size_t w_rank = (size_t)sa[buf_n - lms_count + p];
                sa[w_rank] = (idx_t)p;
            }
        }
        decode_reduced(data, 1, sa, buf_n, lms_count);
        buckets_tails(&bk);
        for (size_t p = lms_count; p-- > 0;) {
            idx_t li = sa[p];
            sa[p] = 0;
            TAIL_PUSH(sa, &bk, data[li], li);
        }
    }

    /* Final forward pass: induce L-types, replacing each scanned suffix by the bitwise
     * NOT of its BWT character (256 = "suffix starts in the second copy").  :653-690 */
    buckets_heads(&bk);
    {
        idx_t i = (idx_t)buf_n, i_sup = i - 1, i_sup2 = i - 2;
        idx_t push = data[i_sup2] < data[i_sup] ? ~i_sup : i_sup;
        HEAD_PUSH(sa, &bk, data[i_sup], push);
        for (size_t p = 0; p < buf_n; p++) {
            i = sa[p];
            if (i > 0) {
                i_sup = i - 1;
                i_sup2 = i - 2;
                sa[p] = ((size_t)i < n) ? ~(idx_t)data[i_sup] : ~(idx_t)256;
                push = (i_sup2 < 0 || data[i_sup2] < data[i_sup]) ? ~i_sup : i_sup;
                HEAD_PUSH(sa, &bk, data[i_sup], push);
            } else if (i < 0) {
                sa[p] = ~sa[p];
            }
        }
    }

    /* Final backward pass: induce S-types, writing characters.  :692-731 */
    buckets_tails(&bk);
    size_t start_suffix = (size_t)-1;
    for (size_t p = buf_n; p-- > 0;) {
        idx_t i = sa[p];
        if (i > 0) {
            idx_t i_sup = i - 1, i_sup2 = i - 2;
            sa[p] = ((size_t)i < n) ? (idx_t)data[i_sup] : 256;
            idx_t push;
            if (i_sup2 < 0)
                push = 0;
            else if (data[i_sup2] > data[i_sup])
                push = ((size_t)i_sup < n) ? ~(idx_t)data[i_sup2] : ~(idx_t)256;
            else
                push = i_sup;
            TAIL_PUSH(sa, &bk, data[i_sup], push);
        } else if (i < 0) {
            sa[p] = ~sa[p];
        } else {
            start_suffix = p;
        }
    }

    /* Compaction: keep entries < 256, insert S[n-1] at suffix 0's slot.  :735-749 */
    size_t start_ptr = (size_t)-1, j = 0;
    for (size_t p = 0; p < buf_n; p++) {
        if (p == start_suffix) {
            bwt_out[j] = input[n - 1];
            start_ptr = j;
            j++;
        } else if (sa[p] < 256) {
            bwt_out[j] = (uint8_t)sa[p];
            j++;
        }
    }

    buckets_free(&bk);
    free(sa);
    free(data);
    return start_ptr;
}

/* Independent check of orc_bwt: the definition itself (the reference's
 * debug/bwt.py:8-23): suffixes of S||S that start in the first copy, in suffix order.
 * O(n^2 log n) worst case -- small inputs only. */
static const uint8_t *naive_s2;
static size_t naive_len2;
static int naive_cmp(const void *a, const void *b)
{
    uint32_t i = *(const uint32_t *)a, j = *(const uint32_t *)b;
    size_t li = naive_len2 - i, lj = naive_len2 - j, l = li < lj ? li : lj;
    int c = memcmp(naive_s2 + i, naive_s2 + j, l);
    if (c)
        return c;
    return li < lj ? -1 : (li > lj ? 1 : 0);
}

ORC_API size_t orc_bwt_naive(const uint8_t *input, size_t n, uint8_t *bwt_out)
{
    if (n == 0)
        return (size_t)-1;
    uint8_t *s2 = (uint8_t *)malloc(2 * n);
    memcpy(s2, input, n);
    memcpy(s2 + n, input, n);
    uint32_t *ord = (uint32_t *)malloc(n * sizeof(uint32_t));
    for (size_t k = 0; k < n; k++)
        ord[k] = (uint32_t)k;
    naive_s2 = s2;
    naive_len2 = 2 * n;
    qsort(ord, n, sizeof(uint32_t), naive_cmp);
    size_t ptr = (size_t)-1;
    for (size_t k = 0; k < n; k++) {
        if (ord[k] == 0) {
            ptr = k;
            bwt_out[k] = input[n - 1];
        } else {
            bwt_out[k] = input[ord[k] - 1];
        }
    }
    free(ord);
    free(s2);
    return ptr;
}

/* ------------------------------------------------------------------------------------
 * MTF + RLE2 + histogram -- lib/mtf.rs:14-121.
 * out must hold n+1 u16.  freqs: 258 x u32.  Returns m (symbols incl. EOB).
 * ---------------------------------------------------------------------------------- */
static void rle2_flush(uint16_t *out, size_t *m, uint32_t *freqs, size_t zero_count)
{
    /* lib/mtf.rs:46-65: bijective base-2 digits of zero_count, RUNA=0 / RUNB=1 */
    size_t code = zero_count + 1;
    for (;;) {
        size_t bit = code & 1;
        code >>= 1;
        if (code == 0)
            break;
        out[(*m)++] = (uint16_t)bit;
        freqs[bit]++;
    }
}

ORC_API size_t orc_mtf_and_rle(const uint8_t *buf, size_t n, const uint8_t *has_byte, uint16_t *out,
                               uint32_t *freqs, uint32_t *num_syms)
{
    uint16_t names[256];
    uint16_t num_names = 0;
    memset(names, 0, sizeof names);
    for (int b = 0; b < 256; b++) /* :17-24 */
        if (has_byte[b])
            names[b] = num_names++;

    uint16_t eob = (uint16_t)(num_names + 1);
    memset(freqs, 0, 258 * sizeof(uint32_t));

    uint16_t recency[256];
    memset(recency, 0, sizeof recency);
    for (uint16_t k = 0; k < num_names; k++) /* :39-43 */
        recency[k] = k;

    size_t m = 0, zero_count = 0;
    for (size_t i = 0; i < n; i++) { /* :69-104 */
        uint16_t name = names[buf[i]];
        uint16_t primary = recency[0];
        if (name == primary) {
            zero_count++;
        } else {
            if (zero_count != 0) {
                rle2_flush(out, &m, freqs, zero_count);
                zero_count = 0;
            }
            uint16_t n0 = primary;
            for (size_t r = 1; r < 256; r++) { /* :85-97 */
                uint16_t t = recency[r];
                recency[r] = n0;
                n0 = t;
                if (name == n0) {
                    out[m++] = (uint16_t)(r + 1);
                    freqs[r + 1]++;
                    break;
                }
            }
            recency[0] = name;
        }
    }
    if (zero_count != 0) /* :106-109 */
        rle2_flush(out, &m, freqs, zero_count);
    out[m++] = eob; /* :112-113 */
    freqs[eob] = 1;
    *num_syms = (uint32_t)num_names + 2;
    return m;
}

/* ------------------------------------------------------------------------------------
 * Huffman -- lib/huffman.rs.
 * ---------------------------------------------------------------------------------- */
#define HUF_MAX_SYMS 258
#define HUF_MAX_LEN 17 /* lib/huffman.rs:13 */

typedef struct { /* lib/huffman.rs:144-145: Priority(sum_frequency, max_dist) */
    uint64_t w;
    uint8_t d;
} prio_t;

typedef struct {
    uint16_t sym;
    prio_t p;
} hitem_t;

/* derived PartialOrd on the tuple struct: lexicographic */
static int prio_lt(prio_t a, prio_t b) { return a.w < b.w || (a.w == b.w && a.d < b.d); }

typedef struct {
    hitem_t heap[HUF_MAX_SYMS + 1]; /* 1-indexed view via heap[idx-1] */
    size_t len;
} fqueue_t;

static void fq_insert(fqueue_t *q, uint16_t sym, prio_t pr) /* lib/huffman.rs:196-222 */
{
    size_t init_idx = q->len + 1;
    q->heap[q->len].sym = sym;
    q->heap[q->len].p = pr;
    q->len++;
    if (init_idx == 1)
        return;
    size_t this_idx = init_idx;
    for (;;) {
        size_t above = this_idx >> 1;
        hitem_t ab = q->heap[above - 1];
        if (prio_lt(pr, ab.p)) {
            q->heap[this_idx - 1] = ab;
            this_idx = above;
            if (this_idx == 1)
                break;
        } else {
            break;
        }
    }
    if (this_idx != init_idx) {
        q->heap[this_idx - 1].sym = sym;
        q->heap[this_idx - 1].p = pr;
    }
}

static hitem_t fq_extract(fqueue_t *q) /* lib/huffman.rs:225-267 */
{
    hitem_t last = q->heap[q->len - 1];
    q->len--;
    if (q->len == 0)
        return last;
    hitem_t root = q->heap[0];
    q->heap[0] = last;
    size_t heap_size = q->len, this_idx = 1, final_idx;
    for (;;) {
        size_t left = this_idx << 1;
        if (left > heap_size) {
            final_idx = this_idx;
            break;
        }
        size_t right = left + 1, below;
        if (right <= heap_size && prio_lt(q->heap[right - 1].p, q->heap[left - 1].p))
            below = right;
        else
            below = left;
        hitem_t bl = q->heap[below - 1];
        if (prio_lt(last.p, bl.p)) {
            final_idx = this_idx;
            break;
        }
        q->heap[this_idx - 1] = bl;
        this_idx = below;
    }
    q->heap[final_idx - 1] = last;
    return root;
}

/* lib/huffman.rs:271-298 (+ Tree :20-102).  Node ids: 0 root, 1..n leaves, n+1.. inner. */
ORC_API void orc_build_table_from_freqs(uint32_t num_syms, const uint32_t *freqs, uint8_t *lengths)
{
    int16_t lch[2 * HUF_MAX_SYMS], rch[2 * HUF_MAX_SYMS];
    uint64_t scaling = 1;
    for (;;) {
        size_t nnodes = num_syms + 1; /* root + leaves */
        for (size_t k = 0; k < 2 * (size_t)num_syms; k++)
            lch[k] = rch[k] = -1;
        fqueue_t q;
        q.len = 0;
        for (uint32_t s = 0; s < num_syms; s++) { /* :171-180 */
            prio_t p = { (uint64_t)freqs[s] / scaling + 1, 0 };
            fq_insert(&q, (uint16_t)(s + 1), p);
        }
        for (;;) {
            hitem_t a = fq_extract(&q);
            hitem_t b = fq_extract(&q);
            size_t parent;
            if (nnodes == (size_t)num_syms * 2 - 1) { /* Tree::tie :60-74 */
                lch[0] = (int16_t)a.sym;
                rch[0] = (int16_t)b.sym;
                parent = 0;
            } else {
                parent = nnodes++;
                lch[parent] = (int16_t)a.sym;
                rch[parent] = (int16_t)b.sym;
            }
            if (parent == 0)
                break;
            prio_t s = { a.p.w + b.p.w, (uint8_t)((a.p.d > b.p.d ? a.p.d : b.p.d) + 1) }; /* :147-158 */
            fq_insert(&q, (uint16_t)parent, s);
        }
        /* coding_lengths :78-102 -- leaf depth; traversal order does not affect the result */
        unsigned max_len = 0;
        struct { int16_t id; uint16_t len; } stack[2 * HUF_MAX_SYMS];
        size_t sp = 0;
        stack[sp].id = 0;
        stack[sp].len = 0;
        sp++;
        while (sp) {
            sp--;
            int16_t cur = stack[sp].id;
            uint16_t len = stack[sp].len;
            if (lch[cur] >= 0 && rch[cur] >= 0) {
                stack[sp].id = lch[cur];
                stack[sp].len = (uint16_t)(len + 1);
                sp++;
                stack[sp].id = rch[cur];
                stack[sp].len = (uint16_t)(len + 1);
                sp++;
            } else {
                lengths[cur - 1] = (uint8_t)len;
                if (len > max_len)
                    max_len = len;
            }
        }
        if (max_len <= HUF_MAX_LEN)
            break;
        scaling <<= 1;
    }
}

/* lib/huffman.rs:313-575.  Writes into the sink.  If dbg_tables != NULL it receives
 * num_tables x 258 final code lengths, dbg_ntables the table count. */
static void huffman_encode(orc_sink *o, const uint16_t *input, size_t input_size, uint32_t num_syms,
                           const uint32_t *freqs, uint8_t *dbg_tables, uint32_t *dbg_ntables)
{
    unsigned num_tables = num_syms <= 199 ? 2 : 3; /* :319-326; num_syms <= 258 */
    uint8_t tables[6][HUF_MAX_SYMS];
    uint32_t table_freqs[6][HUF_MAX_SYMS];

    /* initial tables :333-376 */
    size_t freq_remaining = input_size, sym_left = 0;
    for (unsigned t = 0; t < num_tables; t++) {
        size_t target = freq_remaining / (num_tables - t);
        size_t acc = 0, sym_right = sym_left;
        for (;;) {
            acc += freqs[sym_right];
            if (acc >= target || sym_right + 1 == num_syms)
                break;
            sym_right++;
        }
        if (sym_right > sym_left && t != 0 && t != num_tables - 1 && (t % 2) == 1) {
            acc -= freqs[sym_right];
            sym_right--;
        }
        for (size_t s = 0; s < num_syms; s++)
            tables[t][s] = (s >= sym_left && s <= sym_right) ? 15 : 0;
        sym_left = sym_right + 1;
        freq_remaining -= acc;
    }

    memset(table_freqs, 0, sizeof table_freqs);
    size_t nsel = (input_size + 49) / 50;
    uint8_t *selectors = (uint8_t *)malloc(nsel ? nsel : 1);
    size_t sel_len = 0;

    for (unsigned it = 0; it < 4; it++) { /* :389-460 */
        int final_it = (it == 3);
        if (it != 0) /* :402-409 -- zeroes the code-length tables, not the frequencies */
            for (unsigned t = 0; t < num_tables; t++)
                memset(tables[t], 0, num_syms);
        size_t left = 0;
        for (;;) {
            size_t right = left + 50 - 1;
            if (right >= input_size)
                right = input_size - 1;
            unsigned best = 0;
            uint64_t best_cost = UINT64_MAX;
            for (unsigned t = 0; t < num_tables; t++) {
                uint64_t cost = 0;
                for (size_t k = left; k <= right; k++)
                    cost += tables[t][input[k]];
                if (cost < best_cost) {
                    best = t;
                    best_cost = cost;
                }
            }
            for (size_t k = left; k <= right; k++)
                table_freqs[best][input[k]]++;
            if (final_it)
                selectors[sel_len++] = (uint8_t)best;
            left = right + 1;
            if (left >= input_size)
                break;
        }
        for (unsigned t = 0; t < num_tables; t++)
            orc_build_table_from_freqs(num_syms, table_freqs[t], tables[t]);
    }

    if (dbg_tables) {
        for (unsigned t = 0; t < num_tables; t++)
            memcpy(dbg_tables + (size_t)t * HUF_MAX_SYMS, tables[t], num_syms);
        *dbg_ntables = num_tables;
    }

    sink_write_bits(o, (uint8_t)num_tables, 3); /* :467 */
    sink_write_bits_u32(o, (uint32_t)sel_len, 15); /* :470-471 */

    /* selectors, MTF + unary :474-505 */
    {
        size_t smtf[6];
        uint8_t idx_codes[6];
        for (unsigned k = 0; k < num_tables; k++) {
            smtf[k] = k;
            idx_codes[k] = k == 0 ? 0 : (uint8_t)((1u << (k + 1)) - 2);
        }
        for (size_t k = 0; k < sel_len; k++) {
            size_t sel = selectors[k];
            size_t bump = smtf[0];
            if (bump == sel) {
                sink_write_bits(o, 0, 1);
            } else {
                size_t idx = 1;
                for (;;) {
                    size_t stack_sel = smtf[idx];
                    smtf[idx] = bump;
                    if (stack_sel == sel) {
                        sink_write_bits(o, idx_codes[idx], (unsigned)idx + 1);
                        break;
                    }
                    bump = stack_sel;
                    idx++;
                }
                smtf[0] = sel;
            }
        }
    }

    /* delta-coded tables + canonical codes :509-562 */
    uint32_t code_word[6][HUF_MAX_SYMS];
    uint8_t code_len[6][HUF_MAX_SYMS];
    for (unsigned t = 0; t < num_tables; t++) {
        uint8_t min_len = 255, max_len = 0;
        sink_write_bits(o, tables[t][0], 5);
        uint8_t acc = tables[t][0];
        for (size_t s = 0; s < num_syms; s++) {
            uint8_t l = tables[t][s];
            for (;;) {
                if (l == acc) {
                    sink_write_bits(o, 0, 1);
                    break;
                } else if (l > acc) {
                    sink_write_bits(o, 2, 2);
                    acc++;
                } else {
                    sink_write_bits(o, 3, 2);
                    acc--;
                }
            }
            if (l < min_len)
                min_len = l;
            if (l > max_len)
                max_len = l;
        }
        uint32_t word = 0;
        for (unsigned l = min_len; l <= max_len; l++) {
            for (size_t s = 0; s < num_syms; s++) {
                if (tables[t][s] == l) {
                    code_len[t][s] = (uint8_t)l;
                    code_word[t][s] = word;
                    word++;
                }
            }
            word <<= 1;
        }
    }

    /* symbols :565-572 */
    unsigned sel = selectors[0];
    for (size_t k = 0; k < input_size; k++) {
        if (k % 50 == 0)
            sel = selectors[k / 50];
        sink_write_bits_u32(o, code_word[sel][input[k]], code_len[sel][input[k]]);
    }
    free(selectors);
}

/* Stage seam for parity tests: Huffman payload of one block as a standalone zero-padded
 * bit string starting at bit 0.  Returns the number of BITS; tables_out (3*258) optional. */
ORC_API size_t orc_huffman_block(const uint16_t *syms, size_t m, uint32_t num_syms, const uint32_t *freqs,
                                 uint8_t *out, size_t cap, uint8_t *tables_out, uint32_t *ntables_out)
{
    orc_sink o = { out, cap, 0, 0, 0 };
    huffman_encode(&o, syms, m, num_syms, freqs, tables_out, ntables_out);
    size_t bits = o.len * 8 + o.strand_bits;
    sink_close(&o);
    return bits;
}

/* ------------------------------------------------------------------------------------
 * Framing + block loop -- lib/lib.rs:18-132
 * ---------------------------------------------------------------------------------- */
static void write_sym_map(orc_sink *o, const uint8_t *has_byte) /* lib/lib.rs:39-64 */
{
    uint16_t sector_map = 0, sectors[16];
    unsigned ns = 0;
    for (unsigned a = 0; a < 16; a++) {
        sector_map <<= 1;
        uint16_t sector = 0;
        for (unsigned b = 0; b < 16; b++) {
            sector <<= 1;
            if (has_byte[(a << 4) | b])
                sector |= 1;
        }
        if (sector != 0) {
            sector_map |= 1;
            sectors[ns++] = sector;
        }
    }
    uint8_t be[2] = { (uint8_t)(sector_map >> 8), (uint8_t)sector_map };
    sink_write_bytes(o, be, 2);
    for (unsigned k = 0; k < ns; k++) {
        be[0] = (uint8_t)(sectors[k] >> 8);
        be[1] = (uint8_t)sectors[k];
        sink_write_bytes(o, be, 2);
    }
}

typedef struct { /* per-block record for tests / statistics */
    uint64_t in_off, in_len, rle_len, m;
    uint32_t crc, ptr, num_syms, pad;
} orc_block_info;

/* encode() over an in-memory slice, lib/lib.rs:84-132.  Returns bytes consumed; *out_len is
 * the stream length (may exceed cap: nothing is written past cap).  blocks/max_blocks optional. */
ORC_API size_t orc_encode(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap,
                          size_t *out_len, orc_block_info *blocks, size_t max_blocks, size_t *nblocks)
{
    orc_sink o = { out, cap, 0, 0, 0 };
    const uint8_t hdr[4] = { 0x42, 0x5A, 0x68, (uint8_t)('0' + level) }; /* :18-22 */
    sink_write_bytes(&o, hdr, 4);

    size_t max_len = (size_t)100000 * (size_t)level - 1;
    uint8_t *rle = (uint8_t *)malloc(max_len + 1);
    uint8_t *bw = (uint8_t *)malloc(max_len + 1);
    uint16_t *syms = (uint16_t *)malloc((max_len + 2) * sizeof(uint16_t));
    uint8_t has_byte[256];
    uint32_t freqs[258];

    uint32_t stream_crc = 0;
    size_t consumed = 0, nb = 0;
    for (;;) {
        size_t rle_len;
        uint32_t chk;
        size_t used = orc_rle_one(in + consumed, n - consumed, level, rle, &rle_len, &chk);
        if (used == 0)
            break;
        stream_crc = chk ^ ((stream_crc << 1) | (stream_crc >> 31)); /* :108 */

        size_t ptr = orc_bwt(rle, rle_len, bw, has_byte);

        /* write_block_header :24-36 */
        const uint8_t magic[6] = { 0x31, 0x41, 0x59, 0x26, 0x53, 0x59 };
        sink_write_bytes(&o, magic, 6);
        uint8_t be[4] = { (uint8_t)(chk >> 24), (uint8_t)(chk >> 16), (uint8_t)(chk >> 8), (uint8_t)chk };
        sink_write_bytes(&o, be, 4);
        sink_write_bits(&o, 0, 1);
        uint8_t p3[3] = { (uint8_t)(ptr >> 16), (uint8_t)(ptr >> 8), (uint8_t)ptr };
        sink_write_bytes(&o, p3, 3);
        write_sym_map(&o, has_byte);

        uint32_t num_syms;
        size_t m = orc_mtf_and_rle(bw, rle_len, has_byte, syms, freqs, &num_syms);
        huffman_encode(&o, syms, m, num_syms, freqs, NULL, NULL);

        if (blocks && nb < max_blocks) {
            blocks[nb].in_off = consumed;
            blocks[nb].in_len = used;
            blocks[nb].rle_len = rle_len;
            blocks[nb].m = m;
            blocks[nb].crc = chk;
            blocks[nb].ptr = (uint32_t)ptr;
            blocks[nb].num_syms = num_syms;
            blocks[nb].pad = 0;
        }
        nb++;
        consumed += used;
        if (consumed >= n) /* rle_out.raw == None, :121-125 */
            break;
    }

    const uint8_t foot[6] = { 0x17, 0x72, 0x45, 0x38, 0x50, 0x90 }; /* :66-70 */
    sink_write_bytes(&o, foot, 6);
    uint8_t be[4] = { (uint8_t)(stream_crc >> 24), (uint8_t)(stream_crc >> 16), (uint8_t)(stream_crc >> 8),
                      (uint8_t)stream_crc };
    sink_write_bytes(&o, be, 4);
    sink_close(&o);

    free(rle);
    free(bw);
    free(syms);
    *out_len = o.len;
    if (nblocks)
        *nblocks = nb;
    return consumed;
}
