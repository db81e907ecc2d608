/* abi_facade.c -- the calling sequence of rust/src/lib.rs:66-141 (banzai::encode over libbzhip.so), executed.
 *
 * The image has no Rust toolchain, so the Rust facade has never run; this is its twin in C, operation for operation:
 * a reader that hands out `slice`-byte pieces (BufReader's fill_buf / consume, 8 KiB by default: bnz/src/main.rs:263;
 * encode_file uses 16 MiB), small slices coalesced into a 4 MiB stage, a slice of at least 4 MiB passed straight
 * through ONLY while the stage is empty, end of input fed as (staged remainder, eof = 1) -- also when the remainder is
 * empty --, bzh_stream_bound() asked before every feed and the output vector grown to it, every status checked.
 * Reference surface: lib/lib.rs:84-153.
 *
 *   abi_facade <level> <slice bytes> <input file> <output file>     exit 0 and prints "consumed <n>"
 * Built by tests/test_gpu_parity.py::test_rust_facade_twin (gcc, links banzai_amd/libbzhip.so); test infrastructure.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bzhip.h"

#define STAGE ((size_t)4 << 20)

struct out_vec {
    uint8_t *p;
    size_t len;
};

static int feed(bzh_ctx *ctx, const uint8_t *chunk, size_t n, int eof, struct out_vec *out, FILE *writer)
{
    const size_t cap = bzh_stream_bound(ctx, n);
    if (out->len < cap) { /* out.resize(cap, 0) */
        uint8_t *q = (uint8_t *)realloc(out->p, cap);
        if (!q) return -100;
        memset(q + out->len, 0, cap - out->len);
        out->p = q;
        out->len = cap;
    }
    size_t out_len = 0;
    static const uint8_t none = 0; /* (an empty Rust slice still has a non-null pointer) */
    const int status = bzh_stream_feed(ctx, n ? chunk : &none, n, eof, out->p, out->len, &out_len);
    if (status != BZH_OK) {
        fprintf(stderr, "abi_facade: %s: %s\n", bzh_strerror(status), bzh_last_error(ctx));
        return status;
    }
    return fwrite(out->p, 1, out_len, writer) == out_len ? 0 : -101;
}

int main(int argc, char **argv)
{
    if (argc != 5) {
        fprintf(stderr, "usage: abi_facade <level> <slice bytes> <input> <output>\n");
        return 2;
    }
    const int level = atoi(argv[1]);
    const size_t slice = (size_t)strtoull(argv[2], NULL, 10);
    if (level < 1 || level > 9 || slice == 0) return 2; /* assert!(1 <= level && level <= 9) */
    FILE *reader = fopen(argv[3], "rb");
    FILE *writer = fopen(argv[4], "wb");
    if (!reader || !writer) return 3;
    const char *dev = getenv("BZHIP_DEVICE");
    bzh_ctx *ctx = NULL;
    int status = bzh_create(&ctx, dev ? atoi(dev) : 0, level, 0);
    if (status != BZH_OK) {
        fprintf(stderr, "abi_facade: bzh_create: %s\n", bzh_strerror(status));
        return 4;
    }
    status = bzh_stream_begin(ctx);
    if (status != BZH_OK) return 4;
    uint8_t *buf = (uint8_t *)malloc(slice);   /* the BufReader's buffer */
    uint8_t *stage = (uint8_t *)malloc(STAGE + slice); /* Vec::with_capacity(STAGE), extend_from_slice may grow it */
    size_t staged = 0;
    struct out_vec out = {NULL, 0};
    int rc = 0;
    if (!buf || !stage) return 5;
    for (;;) {
        const size_t got = fread(buf, 1, slice, reader); /* fill_buf */
        if (got == 0) {
            rc = feed(ctx, stage, staged, 1, &out, writer); /* end of input: whatever is staged, with the eof mark */
            break;
        }
        if (staged == 0 && got >= STAGE) {
            rc = feed(ctx, buf, got, 0, &out, writer);
        } else {
            memcpy(stage + staged, buf, got);
            staged += got;
            if (staged >= STAGE) {
                rc = feed(ctx, stage, staged, 0, &out, writer);
                staged = 0;
            }
        }
        if (rc) break;
        /* reader.consume(len) */
    }
    if (rc == 0 && fflush(writer) != 0) rc = -102;
    if (rc == 0) printf("consumed %zu\n", bzh_stream_consumed(ctx));
    bzh_destroy(ctx);
    fclose(reader);
    fclose(writer);
    free(buf);
    free(stage);
    free(out.p);
    return rc ? 6 : 0;
}
