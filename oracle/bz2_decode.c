/* bz2_decode.c -- a small, strict bzip2 DECODER (CPU, C99).
 *
 * TEST INFRASTRUCTURE ONLY (SURVEY.md section 8f, row f3 "verification tooling"): the reference has no
 * decompressor (README.md:9) and checks its encoder by decoding with libbz2 in
 * fuzz/fuzz_targets/round_trip.rs:8-22.  This file gives the repo's round-trip tests the same
 * property without depending on the system's libbz2: tests/ decode what the HIP path produced and
 * compare with the input.  Nothing under banzai_amd/ links or loads it.
 *
 * It follows the published bzip2 1.0.x stream format (the format the reference writes at
 * lib/lib.rs:18-70 and lib/huffman.rs:464-572), and is deliberately unforgiving: every CRC, every
 * table constraint and the exact end of the stream are checked, so a malformed stream is an error,
 * never a best-effort result.  Randomised blocks (a bzip2 0.9.0 feature no current encoder emits)
 * are rejected.  Pinned against Python's bz2 (libbz2) in tests/test_decoder.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define ORC_API __attribute__((visibility("default")))
#else
#define ORC_API
#endif

enum {
    BZD_OK = 0,
    BZD_E_MAGIC = -1,    /* not "BZh1".."BZh9" */
    BZD_E_TRUNC = -2,    /* stream ends early */
    BZD_E_FORMAT = -3,   /* a field outside what the format allows */
    BZD_E_BLOCK_CRC = -4,
    BZD_E_STREAM_CRC = -5,
    BZD_E_CAP = -6,      /* output buffer too small */
    BZD_E_NOMEM = -7,
    BZD_E_TRAILING = -8  /* bytes after the end of the stream */
};

typedef struct {
    const uint8_t *p;
    size_t n, pos; /* pos in bits */
    int err;
} BitR;

static uint32_t get(BitR *r, int nbits) /* MSB first, nbits <= 32 */
{
    uint32_t v = 0;
    for (int k = 0; k < nbits; k++) {
        if ((r->pos >> 3) >= r->n) {
            r->err = BZD_E_TRUNC;
            return 0;
        }
        v = (v << 1) | ((r->p[r->pos >> 3] >> (7 - (r->pos & 7))) & 1u);
        r->pos++;
    }
    return v;
}

static uint32_t crc_table[256];
static void crc_init(void)
{
    if (crc_table[1]) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i << 24;
        for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
        crc_table[i] = c;
    }
}

#define MAX_SYMS 258
#define MAX_LEN 20
#define GROUP 50

typedef struct {
    int32_t limit[MAX_LEN + 2], base[MAX_LEN + 2], perm[MAX_SYMS];
    int minlen, maxlen;
} Code;

/* canonical code from lengths: symbols of one length are numbered in symbol order */
static int make_code(Code *c, const uint8_t *len, int nsyms)
{
    int count[MAX_LEN + 2];
    memset(count, 0, sizeof count);
    c->minlen = 32;
    c->maxlen = 0;
    for (int s = 0; s < nsyms; s++) {
        if (len[s] < 1 || len[s] > MAX_LEN) return BZD_E_FORMAT;
        count[len[s]]++;
        if (len[s] < c->minlen) c->minlen = len[s];
        if (len[s] > c->maxlen) c->maxlen = len[s];
    }
    int pp = 0;
    for (int l = c->minlen; l <= c->maxlen; l++)
        for (int s = 0; s < nsyms; s++)
            if (len[s] == l) c->perm[pp++] = s;
    int32_t code = 0, idx = 0;
    for (int l = c->minlen; l <= c->maxlen; l++) {
        c->base[l] = idx - code; /* perm index = code + base[l] */
        code += count[l];
        idx += count[l];
        c->limit[l] = code - 1; /* largest code of this length */
        if (code > (1 << l)) return BZD_E_FORMAT; /* over-subscribed */
        code <<= 1;
    }
    return BZD_OK;
}

/* Decodes a whole stream.  Returns BZD_OK and the byte count in *out_len, or a negative code. */
ORC_API int orc_bz2_decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len)
{
    crc_init();
    BitR r = {in, n, 0, 0};
    *out_len = 0;
    if (n < 4 || in[0] != 'B' || in[1] != 'Z' || in[2] != 'h' || in[3] < '1' || in[3] > '9') return BZD_E_MAGIC;
    const uint32_t block_max = 100000u * (uint32_t)(in[3] - '0');
    r.pos = 32;
    uint32_t *tt = (uint32_t *)malloc(sizeof(uint32_t) * block_max);
    uint8_t *sel = (uint8_t *)malloc(32768 + 8);
    if (!tt || !sel) {
        free(tt);
        free(sel);
        return BZD_E_NOMEM;
    }
    size_t opos = 0;
    uint32_t stream_crc = 0;
    int rc = BZD_OK;
#define FAIL(code)  \
    do {            \
        rc = (code); \
        goto done;   \
    } while (0)
    for (;;) {
        const uint32_t m_hi = get(&r, 24), m_lo = get(&r, 24);
        if (r.err) FAIL(r.err);
        if (m_hi == 0x177245u && m_lo == 0x385090u) { /* end of stream */
            const uint32_t want = get(&r, 32);
            if (r.err) FAIL(r.err);
            if (want != stream_crc) FAIL(BZD_E_STREAM_CRC);
            /* the rest of the last byte is padding; nothing may follow */
            if (((r.pos + 7) >> 3) != n) FAIL(BZD_E_TRAILING);
            break;
        }
        if (m_hi != 0x314159u || m_lo != 0x265359u) FAIL(BZD_E_FORMAT);
        const uint32_t block_crc = get(&r, 32);
        if (get(&r, 1)) FAIL(BZD_E_FORMAT); /* randomised block */
        const uint32_t orig_ptr = get(&r, 24);
        /* symbol map */
        uint8_t seq[256];
        int nin = 0;
        const uint32_t groups16 = get(&r, 16);
        for (int g = 0; g < 16; g++) {
            if (!((groups16 >> (15 - g)) & 1u)) continue;
            const uint32_t bits = get(&r, 16);
            for (int k = 0; k < 16; k++)
                if ((bits >> (15 - k)) & 1u) seq[nin++] = (uint8_t)(g * 16 + k);
        }
        if (r.err) FAIL(r.err);
        if (nin == 0) FAIL(BZD_E_FORMAT);
        const int alpha = nin + 2;
        const int ngroups = (int)get(&r, 3);
        const int nsel = (int)get(&r, 15);
        if (r.err) FAIL(r.err);
        if (ngroups < 2 || ngroups > 6 || nsel < 1) FAIL(BZD_E_FORMAT);
        { /* selectors: MTF over the group numbers, unary coded */
            uint8_t pos[6] = {0, 1, 2, 3, 4, 5};
            for (int i = 0; i < nsel; i++) {
                int j = 0;
                while (get(&r, 1)) {
                    if (++j >= ngroups) FAIL(BZD_E_FORMAT);
                }
                if (r.err) FAIL(r.err);
                const uint8_t v = pos[j];
                for (; j > 0; j--) pos[j] = pos[j - 1];
                pos[0] = v;
                sel[i] = v;
            }
        }
        Code codes[6];
        for (int t = 0; t < ngroups; t++) { /* delta-coded lengths */
            uint8_t len[MAX_SYMS];
            int cur = (int)get(&r, 5);
            for (int s = 0; s < alpha; s++) {
                for (;;) {
                    if (cur < 1 || cur > MAX_LEN) FAIL(BZD_E_FORMAT);
                    if (!get(&r, 1)) break;
                    cur += get(&r, 1) ? -1 : 1;
                    if (r.err) FAIL(r.err);
                }
                len[s] = (uint8_t)cur;
            }
            if (r.err) FAIL(r.err);
            if (make_code(&codes[t], len, alpha)) FAIL(BZD_E_FORMAT);
        }
        /* symbols -> MTF/RLE2 inverse, straight into tt[] (low byte) with byte counts */
        uint32_t unzftab[256];
        memset(unzftab, 0, sizeof unzftab);
        uint8_t mtf[256];
        for (int k = 0; k < 256; k++) mtf[k] = (uint8_t)k;
        uint32_t nblock = 0;
        const int eob = alpha - 1;
        int group_left = 0, gi = -1;
        const Code *gc = NULL;
        uint32_t run = 0, run_weight = 1;
        for (;;) {
            if (group_left == 0) {
                if (++gi >= nsel) FAIL(BZD_E_FORMAT);
                gc = &codes[sel[gi]];
                group_left = GROUP;
            }
            group_left--;
            int l = gc->minlen;
            int32_t code = (int32_t)get(&r, l);
            while (l <= gc->maxlen && code > gc->limit[l]) {
                code = (code << 1) | (int32_t)get(&r, 1);
                l++;
            }
            if (r.err) FAIL(r.err);
            if (l > gc->maxlen) FAIL(BZD_E_FORMAT);
            const int idx = code + gc->base[l];
            if (idx < 0 || idx >= alpha) FAIL(BZD_E_FORMAT);
            const int sym = gc->perm[idx];
            if (sym <= 1) { /* RUNA / RUNB: bijective base-2 digits of a run of the front symbol */
                if (run_weight > (1u << 21)) FAIL(BZD_E_FORMAT);
                run += run_weight << sym;
                run_weight <<= 1;
                continue;
            }
            if (run) {
                const uint8_t b = seq[mtf[0]];
                if (run > block_max - nblock) FAIL(BZD_E_FORMAT);
                unzftab[b] += run;
                while (run--) tt[nblock++] = b;
                run = 0;
                run_weight = 1;
            }
            if (sym == eob) break;
            { /* MTF position sym-1 */
                const int p = sym - 1;
                if (p >= nin) FAIL(BZD_E_FORMAT);
                const uint8_t v = mtf[p];
                memmove(mtf + 1, mtf, (size_t)p);
                mtf[0] = v;
                if (nblock >= block_max) FAIL(BZD_E_FORMAT);
                unzftab[seq[v]]++;
                tt[nblock++] = seq[v];
            }
        }
        if (nblock == 0 || orig_ptr >= nblock) FAIL(BZD_E_FORMAT);
        /* inverse BWT: T vector in the upper 24 bits of tt */
        uint32_t cftab[257];
        cftab[0] = 0;
        for (int k = 0; k < 256; k++) cftab[k + 1] = cftab[k] + unzftab[k];
        for (uint32_t i = 0; i < nblock; i++) {
            const uint8_t b = (uint8_t)(tt[i] & 0xFF);
            tt[cftab[b]] |= i << 8;
            cftab[b]++;
        }
        /* walk + inverse RLE1 (4 equal bytes, then a count byte) + block CRC */
        uint32_t crc = 0xFFFFFFFFu;
        uint32_t tpos = tt[orig_ptr] >> 8;
        int same = 0, prev = -1;
        for (uint32_t i = 0; i < nblock; i++) {
            const uint32_t e = tt[tpos];
            const uint8_t b = (uint8_t)(e & 0xFF);
            tpos = e >> 8;
            if (same == 4) { /* b is a repeat count */
                if (opos + b > cap) FAIL(BZD_E_CAP);
                for (int k = 0; k < b; k++) {
                    out[opos++] = (uint8_t)prev;
                    crc = (crc << 8) ^ crc_table[(crc >> 24) ^ (uint8_t)prev];
                }
                same = 0;
                prev = -1;
                continue;
            }
            if ((int)b == prev) {
                same++;
            } else {
                same = 1;
                prev = b;
            }
            if (opos >= cap) FAIL(BZD_E_CAP);
            out[opos++] = b;
            crc = (crc << 8) ^ crc_table[(crc >> 24) ^ b];
        }
        crc = ~crc;
        if (crc != block_crc) FAIL(BZD_E_BLOCK_CRC);
        stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ crc;
    }
done:
    free(tt);
    free(sel);
    *out_len = opos;
    return rc;
}
