/* san_roundtrip.c -- encode -> decode round trips of the oracle under AddressSanitizer / UBSan (make -C oracle san).
 * Test infrastructure only: it exercises oracle/banzai_oracle.c and oracle/bz2_decode.c, never the product.
 * Mirrors the reference's fuzz target fuzz/fuzz_targets/round_trip.rs:8-22 (encode, decode, compare) with a seeded
 * generator instead of libFuzzer.   usage: san_roundtrip [rounds] [seed] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

size_t orc_encode(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap, size_t *out_len, void *blocks,
                  size_t max_blocks, size_t *nblocks);
int orc_bz2_decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);

static uint64_t st;
static uint32_t rnd(void)
{
    st ^= st >> 12;
    st ^= st << 25;
    st ^= st >> 27;
    return (uint32_t)((st * 0x2545F4914F6CDD1Dull) >> 32);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    st = argc > 2 ? strtoull(argv[2], NULL, 10) : 0x9E3779B97F4A7C15ull;
    const size_t maxn = 260000;
    uint8_t *in = malloc(maxn), *enc = malloc(maxn * 2 + 65536), *dec = malloc(maxn + 64);
    for (int r = 0; r < rounds; r++) {
        const int level = 1 + (int)(rnd() % 9);
        const int mode = (int)(rnd() % 5);
        size_t n = rnd() % (r % 7 == 0 ? maxn : 20000);
        if (r < 4) n = (size_t)r; /* empty and tiny inputs */
        const uint32_t alpha = 1 + rnd() % 255;
        for (size_t i = 0; i < n;) {
            if (mode == 0) {
                in[i++] = (uint8_t)rnd();
            } else if (mode == 1) { /* runs of every length around 4, 255, 256 */
                size_t len = (rnd() % 3 == 0) ? 250 + rnd() % 20 : 1 + rnd() % 9;
                const uint8_t c = (uint8_t)(rnd() % alpha);
                while (len-- && i < n) in[i++] = c;
            } else if (mode == 2) { /* periodic */
                const size_t p = 1 + rnd() % 17;
                for (size_t k = 0; k < p && i < n; k++) in[i++] = (uint8_t)('a' + k % 7);
            } else if (mode == 3) {
                in[i++] = (uint8_t)(rnd() % alpha);
            } else { /* one enormous run with noise */
                in[i++] = (rnd() % 4096) ? 'z' : (uint8_t)rnd();
            }
        }
        size_t nb = 0, len = 0;
        const size_t used = orc_encode(in, n, level, enc, maxn * 2 + 65536, &len, NULL, 0, &nb);
        size_t got = 0;
        const int rc = len <= maxn * 2 + 65536 ? orc_bz2_decode(enc, len, dec, maxn + 64, &got) : -99;
        if (rc != 0 || used != n || got != n || memcmp(in, dec, n) != 0) {
            fprintf(stderr, "san_roundtrip: round %d (level %d, mode %d, n %zu): rc %d, %zu bytes back\n", r, level, mode, n, rc, got);
            return 1;
        }
    }
    printf("san_roundtrip: %d round trips clean\n", rounds);
    free(in);
    free(enc);
    free(dec);
    return 0;
}
