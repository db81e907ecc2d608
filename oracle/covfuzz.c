/* covfuzz.c -- a small coverage-guided mutational fuzzer for the CPU oracle and the strict decoder.
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/): it never touches the product.
 *
 * The reference fuzzes encode -> decode -> compare with libFuzzer (fuzz/fuzz_targets/round_trip.rs:8-22).  This image
 * has no libFuzzer runtime, so the feedback loop is restated here on gcc's -fsanitize-coverage=trace-pc: the oracle and
 * the decoder are compiled with it (make -C oracle covfuzz), every basic block they enter calls
 * __sanitizer_cov_trace_pc below, edges are hashed into a 64 Ki map with AFL's hit-count buckets, and an input that
 * lights a new (edge, bucket) bit joins the corpus and becomes a parent of later mutations.
 * An input is [level byte][data]: level = 1 + byte % 9.
 *
 *   covfuzz run <corpus_dir> <seconds> [seed]     fuzz; new inputs are written to corpus_dir as <fnv64>.bin
 *   covfuzz min <corpus_dir> <out_dir>            greedy minimisation: smallest inputs first, keep what adds coverage
 *   covfuzz replay <corpus_dir>                   round-trip every input, print the covered (edge, bucket) bits
 * The minimised corpus is committed (tests/golden/fuzz_corpus.zip) and replayed through the HIP path against the
 * oracle's streams by tests/test_gpu_parity.py::test_coverage_guided_corpus. */
#define _POSIX_C_SOURCE 200809L
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

size_t orc_encode(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap, size_t *out_len, void *blocks,
                  size_t max_blocks, size_t *nblocks);
int orc_bz2_decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);

/* ---- coverage ------------------------------------------------------------------------------------------------ */
#define MAP (1u << 16)
static uint8_t cov[MAP];  /* hit counts of the run in flight */
static uint8_t seen[MAP]; /* (edge, bucket) bits of the whole session */
static uintptr_t prev_pc;

void __sanitizer_cov_trace_pc(void)
{
    const uintptr_t pc = (uintptr_t)__builtin_return_address(0);
    const uint32_t idx = (uint32_t)((pc ^ prev_pc) * 0x9E3779B1u >> 7) & (MAP - 1);
    if (cov[idx] != 255) cov[idx]++;
    prev_pc = pc >> 1;
}

static uint8_t bucket(uint8_t c)
{
    if (c == 0) return 0;
    if (c == 1) return 1;
    if (c == 2) return 2;
    if (c == 3) return 4;
    if (c < 8) return 8;
    if (c < 16) return 16;
    if (c < 32) return 32;
    if (c < 128) return 64;
    return 128;
}

/* -> number of new (edge, bucket) bits; merges them into `seen` when commit != 0 */
static int new_bits(int commit)
{
    int fresh = 0;
    for (uint32_t i = 0; i < MAP; i++) {
        if (!cov[i]) continue;
        const uint8_t b = bucket(cov[i]);
        if (b & ~seen[i]) {
            fresh++;
            if (commit) seen[i] |= b;
        }
    }
    return fresh;
}

/* ---- the target: encode, decode, compare (aborts on a mismatch: that input is the finding) -------------------- */
#define MAXN (300u * 1024u)
static uint8_t *enc_buf, *dec_buf;

static void target(const uint8_t *in, size_t len, const char *what)
{
    memset(cov, 0, sizeof cov);
    prev_pc = 0;
    if (len == 0) return;
    const int level = 1 + in[0] % 9;
    const uint8_t *d = in + 1;
    const size_t n = len - 1;
    const size_t cap = MAXN * 2 + 65536;
    size_t nb = 0, elen = 0, got = 0;
    const size_t used = orc_encode(d, n, level, enc_buf, cap, &elen, NULL, 0, &nb);
    const int rc = elen <= cap ? orc_bz2_decode(enc_buf, elen, dec_buf, MAXN + 64, &got) : -99;
    if (rc != 0 || used != n || got != n || memcmp(d, dec_buf, n) != 0) {
        fprintf(stderr, "covfuzz: ROUND TRIP FAILED (%s): level %d, n %zu, rc %d, %zu bytes back\n", what, level, n, rc, got);
        FILE *f = fopen("covfuzz_crash.bin", "wb");
        if (f) {
            fwrite(in, 1, len, f);
            fclose(f);
        }
        abort();
    }
}

/* ---- corpus ---------------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *d;
    size_t n;
} Item;
static Item *items;
static size_t nitems, capitems;

static void add_item(const uint8_t *d, size_t n)
{
    if (nitems == capitems) {
        capitems = capitems ? capitems * 2 : 256;
        items = realloc(items, capitems * sizeof *items);
    }
    items[nitems].d = malloc(n ? n : 1);
    memcpy(items[nitems].d, d, n);
    items[nitems].n = n;
    nitems++;
}

static uint64_t fnv64(const uint8_t *d, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ d[i]) * 1099511628211ull;
    return h;
}

static void save_item(const char *dir, const uint8_t *d, size_t n)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%016llx.bin", dir, (unsigned long long)fnv64(d, n));
    FILE *f = fopen(path, "wb");
    if (!f) return;
    fwrite(d, 1, n, f);
    fclose(f);
}

static int cmp_size(const void *a, const void *b)
{
    const Item *x = a, *y = b;
    if (x->n != y->n) return x->n < y->n ? -1 : 1;
    return memcmp(x->d, y->d, x->n);
}

static void load_dir(const char *dir)
{
    DIR *dp = opendir(dir);
    if (!dp) return;
    struct dirent *e;
    uint8_t *buf = malloc(MAXN + 8);
    while ((e = readdir(dp))) {
        const size_t l = strlen(e->d_name);
        if (l < 5 || strcmp(e->d_name + l - 4, ".bin")) continue;
        char path[1024];
        snprintf(path, sizeof path, "%s/%s", dir, e->d_name);
        FILE *f = fopen(path, "rb");
        if (!f) continue;
        const size_t n = fread(buf, 1, MAXN + 1, f);
        fclose(f);
        add_item(buf, n);
    }
    free(buf);
    closedir(dp);
    qsort(items, nitems, sizeof *items, cmp_size); /* (a fixed order whatever the directory says) */
}

/* ---- mutations --------------------------------------------------------------------------------------------------- */
static uint64_t rs;
static uint32_t rnd(void)
{
    rs ^= rs >> 12;
    rs ^= rs << 25;
    rs ^= rs >> 27;
    return (uint32_t)((rs * 0x2545F4914F6CDD1Dull) >> 32);
}
static const uint32_t RUNS[] = {2, 3, 4, 5, 6, 7, 8, 254, 255, 256, 257, 258, 259, 260, 509, 510, 511, 512, 1020, 1275, 4096, 70000};

static size_t mutate(uint8_t *b, size_t n)
{
    const int steps = 1 + (int)(rnd() % 4);
    for (int s = 0; s < steps; s++) {
        switch (rnd() % 14) {
        case 0: /* flip a bit */
            if (n) b[rnd() % n] ^= (uint8_t)(1u << (rnd() % 8));
            break;
        case 1: /* set a byte */
            if (n) b[rnd() % n] = (uint8_t)rnd();
            break;
        case 2: { /* insert a run of a telling length */
            const uint32_t len = RUNS[rnd() % (sizeof RUNS / sizeof *RUNS)];
            if (n + len > MAXN) break;
            const size_t at = n > 1 ? 1 + rnd() % n : n;
            memmove(b + at + len, b + at, n - at);
            memset(b + at, (rnd() & 1) && at > 1 ? b[at - 1] : (uint8_t)rnd(), len);
            n += len;
            break;
        }
        case 3: { /* copy an earlier stretch to a later place (repeats drive the suffix sort and the Huffman tables) */
            if (n < 8) break;
            const size_t len = 1 + rnd() % (n / 2), from = 1 + rnd() % (n - len), to = 1 + rnd() % (n - len);
            memmove(b + to, b + from, len);
            break;
        }
        case 4: { /* append a copy of the data (periodic blocks, block cuts) */
            if (n < 2 || 2 * n > MAXN) break;
            memcpy(b + n, b + 1, n - 1);
            n += n - 1;
            break;
        }
        case 5: { /* delete a stretch */
            if (n < 4) break;
            const size_t len = 1 + rnd() % (n / 2), at = 1 + rnd() % (n - len);
            memmove(b + at, b + at + len, n - at - len);
            n -= len;
            break;
        }
        case 6: { /* append random bytes over a small alphabet */
            const uint32_t len = 1 + rnd() % 4096, alpha = 1 + rnd() % 255;
            if (n + len > MAXN) break;
            for (uint32_t k = 0; k < len; k++) b[n + k] = (uint8_t)(rnd() % alpha);
            n += len;
            break;
        }
        case 7: /* another level */
            if (n) b[0] = (uint8_t)rnd();
            break;
        case 8: { /* truncate */
            if (n > 2) n = 1 + rnd() % n;
            break;
        }
        case 9: { /* splice with another corpus entry */
            const Item *o = &items[rnd() % nitems];
            if (o->n < 2 || n < 2) break;
            const size_t cut = 1 + rnd() % (n - 1), ocut = 1 + rnd() % (o->n - 1);
            size_t take = o->n - ocut;
            if (cut + take > MAXN) take = MAXN - cut;
            memcpy(b + cut, o->d + ocut, take);
            n = cut + take;
            break;
        }
        case 10: { /* restrict the alphabet of a stretch */
            if (n < 4) break;
            const size_t len = 1 + rnd() % (n - 1), at = 1 + rnd() % (n - len);
            const uint32_t alpha = 1 + rnd() % 4;
            for (size_t k = 0; k < len; k++) b[at + k] = (uint8_t)('a' + b[at + k] % alpha);
            break;
        }
        case 11: { /* grow towards a block boundary: runs of 4..259 with their count bytes fill the RLE1 budget fast */
            const uint32_t reps = 1 + rnd() % 400, len = 4 + rnd() % 256;
            if (n + (size_t)reps * len > MAXN) break;
            for (uint32_t r = 0; r < reps; r++) {
                memset(b + n, (uint8_t)(r + rnd() % 3), len);
                n += len;
            }
            break;
        }
        case 12: { /* many distinct bytes: the 258-symbol alphabet and three Huffman tables */
            if (n + 512 > MAXN) break;
            for (uint32_t k = 0; k < 512; k++) b[n + k] = (uint8_t)(k * 7 + rnd() % 3);
            n += 512;
            break;
        }
        default: { /* overwrite a stretch with a byte */
            if (n < 3) break;
            const size_t len = 1 + rnd() % (n - 1 < 600 ? n - 1 : 600), at = 1 + rnd() % (n - len);
            memset(b + at, b[at], len);
            break;
        }
        }
    }
    return n;
}

static int count_seen(void)
{
    int c = 0;
    for (uint32_t i = 0; i < MAP; i++) c += __builtin_popcount(seen[i]);
    return c;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: covfuzz run <corpus_dir> <seconds> [seed] | min <corpus_dir> <out_dir> | replay <corpus_dir>\n");
        return 2;
    }
    enc_buf = malloc(MAXN * 2 + 65536);
    dec_buf = malloc(MAXN + 64);
    const char *mode = argv[1], *dir = argv[2];
    load_dir(dir);
    if (!strcmp(mode, "replay") || !strcmp(mode, "min")) {
        size_t kept = 0, bytes = 0;
        if (!strcmp(mode, "min")) mkdir(argv[3], 0777);
        for (size_t k = 0; k < nitems; k++) { /* sorted by size: small inputs get the credit */
            target(items[k].d, items[k].n, "corpus");
            if (new_bits(1) && !strcmp(mode, "min")) {
                save_item(argv[3], items[k].d, items[k].n);
                kept++;
                bytes += items[k].n;
            }
        }
        printf("covfuzz %s: %zu inputs, %d (edge, bucket) bits", mode, nitems, count_seen());
        if (!strcmp(mode, "min")) printf("; kept %zu inputs, %zu bytes", kept, bytes);
        printf("\n");
        return 0;
    }
    const double seconds = argc > 3 ? atof(argv[3]) : 60;
    rs = argc > 4 ? strtoull(argv[4], NULL, 10) * 0x9E3779B97F4A7C15ull + 1 : 0x9E3779B97F4A7C15ull;
    mkdir(dir, 0777);
    if (nitems == 0) { /* seeds: empty, one byte, a phrase, a run, every level once */
        const char *seeds[] = {"\x08", "\x08" "a", "\x00" "It was the best of times, it was the worst of times, ", "\x08" "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa",
                               "\x01" "abababababababababab", "\x02" "\x00\x01\x02\x03\x04\x05\x06\x07"};
        const size_t lens[] = {1, 2, 54, 41, 21, 9};
        for (int k = 0; k < 6; k++) add_item((const uint8_t *)seeds[k], lens[k]);
    }
    for (size_t k = 0; k < nitems; k++) {
        target(items[k].d, items[k].n, "seed");
        new_bits(1);
    }
    uint8_t *buf = malloc(MAXN + 8);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned long execs = 0, found = 0;
    for (;;) {
        if ((execs & 63) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9 > seconds) break;
        }
        /* parent: mostly a recent or a small entry */
        const size_t pick = (rnd() % 3 == 0) ? nitems - 1 - rnd() % (nitems < 8 ? nitems : 8) : rnd() % nitems;
        size_t n = items[pick].n;
        memcpy(buf, items[pick].d, n);
        if (n == 0) buf[n++] = (uint8_t)rnd();
        n = mutate(buf, n);
        if (n > MAXN) n = MAXN;
        target(buf, n, "mutation");
        execs++;
        if (new_bits(1)) {
            add_item(buf, n);
            save_item(dir, buf, n);
            found++;
        }
    }
    printf("covfuzz run: %lu executions, %lu new inputs, corpus %zu, %d (edge, bucket) bits\n", execs, found, nitems, count_seen());
    return 0;
}
