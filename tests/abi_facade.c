/* abi_facade.c -- the calling sequence of rust/src/lib.rs:66-141 (banzai::encode over libbzhip.so), executed.
 *
 * The image has no Rust toolchain, so the Rust facade has never run; this is its twin in C, operation for operation:
 * a reader that hands out `slice`-byte pieces (BufReader's fill_buf / consume, 8 KiB by default: bnz/src/main.rs:263;
 * encode_file uses 16 MiB), small slices coalesced into a 4 MiB stage, a slice of at least 4 MiB passed straight
 * through ONLY while the stage is empty, end of input fed as (staged remainder, eof = 1) -- also when the remainder is
 * empty --, bzh_stream_bound() asked before every feed and the output vector grown to it, every status checked.
 * Reference surface: lib/lib.rs:84-153.
 *
 *   abi_facade <level> <slice bytes> <input file> <output file> [calls]    exit 0 and prints "consumed <n>"
 *
 * encode() takes its context from a process-wide pool keyed by (device, level) -- rust/src/lib.rs `POOL` / `checkout` /
 * `checkin` -- and the context keeps its two host buffers (the output vector, zero-filled up to bzh_stream_bound's worst
 * case, and the stage), so a caller that loops over files pays bzh_create, the arena (allocated and first-touched by the
 * first encode that needs it) and the zero-fill once: with [calls] > 1 the same input is encoded that many times and every
 * call's wall clock is printed ("call <k>: <ms> ms"), the first against the steady state.
 * <slice bytes> = 0: an in-memory reader -- the input is read into memory first (outside the clock) and fill_buf hands out
 * everything that is left, as Rust's `&[u8]` does as a BufRead (what bench.py's value_stream_api feeds from).
 * Built by tests/test_gpu_parity.py::test_rust_facade_twin (gcc, links banzai_amd/libbzhip.so); test infrastructure.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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

/* ---- the context pool (rust/src/lib.rs: POOL, checkout, checkin): idle contexts by (device, level) ------------------- */
#define POOL_SLOTS 16
struct pooled { /* rust/src/lib.rs: struct Ctx { handle, out, stage } */
    bzh_ctx *ctx;
    struct out_vec out;
    uint8_t *stage;
    size_t stage_cap;
};
static struct { int device, level, used; struct pooled c; } pool[POOL_SLOTS];

static void drop_pooled(struct pooled *c)
{
    if (c->ctx) bzh_destroy(c->ctx);
    free(c->out.p);
    free(c->stage);
    memset(c, 0, sizeof *c);
}

static int checkout(int device, int level, struct pooled *c)
{
    for (int k = 0; k < POOL_SLOTS; k++)
        if (pool[k].used && pool[k].device == device && pool[k].level == level) {
            *c = pool[k].c;
            pool[k].used = 0;
            return BZH_OK;
        }
    memset(c, 0, sizeof *c);
    return bzh_create(&c->ctx, device, level, 0);
}

static void checkin(int device, int level, struct pooled *c)
{
    for (int k = 0; k < POOL_SLOTS; k++)
        if (!pool[k].used) {
            pool[k].device = device;
            pool[k].level = level;
            pool[k].c = *c;
            pool[k].used = 1;
            return;
        }
    drop_pooled(c); /* (the pool keeps a bounded number of idle contexts) */
}

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* banzai::encode(reader, writer, level).  `mem` != NULL: the in-memory reader (fill_buf = everything that is left). */
static int encode(FILE *reader, const uint8_t *mem, size_t mem_len, FILE *writer, int level, size_t slice, size_t *consumed)
{
    const char *dev = getenv("BZHIP_DEVICE");
    const int device = dev ? atoi(dev) : 0;
    struct pooled c;
    int status = checkout(device, level, &c);
    if (status != BZH_OK) {
        fprintf(stderr, "abi_facade: bzh_create: %s\n", bzh_strerror(status));
        return 4;
    }
    bzh_ctx *ctx = c.ctx;
    status = bzh_stream_begin(ctx);
    if (status != BZH_OK) {
        drop_pooled(&c); /* (a context whose stream failed does not go back) */
        return 4;
    }
    uint8_t *buf = mem ? NULL : (uint8_t *)malloc(slice);   /* the BufReader's buffer */
    if (c.stage_cap < STAGE + slice) { /* Vec::with_capacity(STAGE), extend_from_slice may grow it */
        free(c.stage);
        c.stage = (uint8_t *)malloc(STAGE + slice);
        c.stage_cap = STAGE + slice;
    }
    uint8_t *stage = c.stage;
    size_t staged = 0, mem_pos = 0; /* stage.clear() */
    int rc = 0;
    if ((!mem && !buf) || !stage) return 5;
    for (;;) {
        const uint8_t *got_p = buf;
        size_t got;
        if (mem) { /* fill_buf of a slice reader: all that is left */
            got_p = mem + mem_pos;
            got = mem_len - mem_pos;
            mem_pos = mem_len;
        } else {
            got = fread(buf, 1, slice, reader); /* fill_buf */
        }
        if (got == 0) {
            rc = feed(ctx, stage, staged, 1, &c.out, writer); /* end of input: whatever is staged, with the eof mark */
            break;
        }
        if (staged == 0 && got >= STAGE) {
            rc = feed(ctx, got_p, got, 0, &c.out, writer);
        } else {
            memcpy(stage + staged, got_p, got);
            staged += got;
            if (staged >= STAGE) {
                rc = feed(ctx, stage, staged, 0, &c.out, writer);
                staged = 0;
            }
        }
        if (rc) break;
        /* reader.consume(len) */
    }
    if (rc == 0 && fflush(writer) != 0) rc = -102;
    if (rc == 0) {
        *consumed = bzh_stream_consumed(ctx);
        checkin(device, level, &c);
    } else {
        drop_pooled(&c);
    }
    free(buf);
    return rc ? 6 : 0;
}

int main(int argc, char **argv)
{
    if (argc != 5 && argc != 6) {
        fprintf(stderr, "usage: abi_facade <level> <slice bytes | 0 = in-memory reader> <input> <output> [calls]\n");
        return 2;
    }
    const int level = atoi(argv[1]);
    const size_t slice = (size_t)strtoull(argv[2], NULL, 10);
    const int calls = argc == 6 ? atoi(argv[5]) : 1;
    if (level < 1 || level > 9 || calls < 1) return 2; /* assert!(1 <= level && level <= 9) */
    uint8_t *mem = NULL;
    size_t mem_len = 0;
    if (slice == 0) { /* the whole input in memory, outside the clock */
        FILE *f = fopen(argv[3], "rb");
        if (!f) return 3;
        fseek(f, 0, SEEK_END);
        mem_len = (size_t)ftell(f);
        fseek(f, 0, SEEK_SET);
        mem = (uint8_t *)malloc(mem_len + 1);
        if (!mem || fread(mem, 1, mem_len, f) != mem_len) return 3;
        fclose(f);
    }
    size_t consumed = 0;
    for (int k = 0; k < calls; k++) {
        FILE *reader = mem ? NULL : fopen(argv[3], "rb");
        FILE *writer = fopen(argv[4], "wb");
        if ((!mem && !reader) || !writer) return 3;
        const double t0 = now_ms();
        const int rc = encode(reader, mem, mem_len, writer, level, slice, &consumed);
        const double t1 = now_ms();
        if (reader) fclose(reader);
        fclose(writer);
        if (rc) return rc;
        if (calls > 1) printf("call %d: %.2f ms\n", k + 1, t1 - t0);
    }
    printf("consumed %zu\n", consumed);
    for (int k = 0; k < POOL_SLOTS; k++)
        if (pool[k].used) drop_pooled(&pool[k].c);
    free(mem);
    return 0;
}
