/* Sanitizer self-test of the C oracle (test infrastructure, like the oracle itself): built by `make -C oracle
 * check-sanitize` with -fsanitize=address,undefined and run over the reference's own end-to-end vector
 * (full_loop.rs:54-67,148: two participants [1,2,3,4], packed Shamir k=3 n=8 t=4 p=433 -> [2,4,6,8]), the additive
 * README walkthrough, the mask paths, the DRBG, the synthetic fill and the wire codec - every buffer exactly sized,
 * so an out-of-bounds access or a signed overflow in the oracle fails the run. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int sdao_additive_generate(int64_t q, int n, const int64_t* secrets, size_t len, const int64_t* rand, int64_t* out, int mode);
int sdao_combine(int64_t q, const int64_t* shares, size_t P, size_t L, size_t stride, int64_t* out, int mode);
int sdao_packed_generate(int64_t prime, int k, int t, int n, int64_t w2, int64_t w3, const int64_t* secrets, size_t len,
                         const int64_t* rand, int64_t* out);
int sdao_packed_reconstruct(int64_t prime, int k, int t, int64_t w2, int64_t w3, size_t dimension, const size_t* indices,
                            size_t n_idx, const int64_t* shares, size_t stride, int64_t* out);
void sdao_positive(const int64_t* v, size_t len, int64_t modulus, int64_t* out);
void sdao_chacha_expand(const int64_t* seed_words, size_t n_words, int64_t q, size_t count, int64_t* out);
void sdao_chacha_combine(const int64_t* seeds, size_t P, size_t n_words, int64_t q, size_t dimension, int64_t* out, int mode);
void sdao_addsub(const int64_t* a, const int64_t* b, size_t len, int64_t q, int subtract, int64_t* out, int mode);
void sdao_drbg_fill(const uint8_t key_bytes[32], int rounds, uint64_t stream, size_t batches, uint32_t T, int64_t modulus,
                    int64_t* out);
void sdao_fill_synthetic(int64_t* out, size_t participants, size_t len, size_t stride, uint64_t first_participant,
                         uint64_t seed, int64_t modulus);
size_t sdao_baseline_pass(int packed, int64_t modulus, int n, int k, int t, int64_t w2, int64_t w3, size_t participants,
                          size_t len, uint64_t first_participant, uint64_t seed, const uint8_t key_bytes[32],
                          int64_t* clerk_sums);
size_t sdao_varint_encode(const int64_t* values, size_t len, uint8_t* out);
size_t sdao_varint_decode(const uint8_t* src, size_t n_bytes, int64_t* out, size_t cap);

static int failures = 0;
#define CHECK(c, what) do { if (!(c)) { printf("FAIL: %s\n", what); ++failures; } } while (0)

int main(void) {
    const int64_t P62 = 4611686006577364993LL, W8 = 631229665360524489LL, W9 = 3451275676410824977LL;
    /* packed Shamir full loop at p = 433 and over the 62-bit prime */
    for (int big = 0; big < 2; ++big) {
        const int64_t p = big ? P62 : 433, w2 = big ? W8 : 354, w3 = big ? W9 : 150;
        const int k = 3, t = 4, n = 8;
        const size_t dim = 4, B = 2;
        const int64_t secrets[4] = {1, 2, 3, 4};
        int64_t* rnd = malloc(B * t * 8);
        int64_t* sh = malloc(2 * n * B * 8);                 /* [participant][clerk][B] */
        for (int part = 0; part < 2; ++part) {
            for (size_t i = 0; i < B * t; ++i) rnd[i] = (int64_t)((i * 2654435761u + 17u * (unsigned)part) % (uint64_t)p);
            CHECK(sdao_packed_generate(p, k, t, n, w2, w3, secrets, dim, rnd, sh + part * n * B) == 0, "packed_generate");
        }
        int64_t* sums = malloc(n * B * 8);
        int64_t* col = malloc(2 * B * 8);
        for (int c = 0; c < n; ++c) {
            memcpy(col, sh + c * B, B * 8);
            memcpy(col + B, sh + n * B + c * B, B * 8);
            CHECK(sdao_combine(p, col, 2, B, B, sums + c * B, 1) == 0, "combine");
        }
        const size_t idx[7] = {7, 5, 4, 3, 2, 1, 0};
        int64_t* pick = malloc(7 * B * 8);
        for (int i = 0; i < 7; ++i) memcpy(pick + i * B, sums + idx[i] * B, B * 8);
        int64_t out[4], pos[4];
        CHECK(sdao_packed_reconstruct(p, k, t, w2, w3, dim, idx, 7, pick, B, out) == 0, "packed_reconstruct");
        sdao_positive(out, 4, p, pos);
        CHECK(pos[0] == 2 && pos[1] == 4 && pos[2] == 6 && pos[3] == 8, "full_loop.rs:148 [2,4,6,8]");
        free(rnd); free(sh); free(sums); free(col); free(pick);
    }
    /* additive README walkthrough (README.md:157), signed mode like the reference */
    {
        const int64_t q = 433;
        const int64_t in[3][10] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, {0}, {0, 1, 0, 1, 0, 1, 0, 1, 0, 1}};
        int64_t sh[3][3 * 10], rnd[20], col[3 * 10], sums[3][10], tot[10], pos[10];
        for (int p = 0; p < 3; ++p) {
            for (int i = 0; i < 20; ++i) rnd[i] = (i * 97 + p * 31) % q;
            CHECK(sdao_additive_generate(q, 3, in[p], 10, rnd, sh[p], 1) == 0, "additive_generate");
        }
        for (int c = 0; c < 3; ++c) {
            for (int p = 0; p < 3; ++p) memcpy(col + p * 10, sh[p] + c * 10, 80);
            sdao_combine(q, col, 3, 10, 10, sums[c], 1);
        }
        for (int c = 0; c < 3; ++c) memcpy(col + c * 10, sums[c], 80);
        sdao_combine(q, col, 3, 10, 10, tot, 1);
        sdao_positive(tot, 10, q, pos);
        const int64_t want[10] = {0, 2, 2, 4, 4, 6, 6, 8, 8, 10};
        CHECK(memcmp(pos, want, 80) == 0, "README.md:157 walkthrough");
    }
    /* masks: unmask(combine(masks), sum(masked)) == sum(secrets) */
    {
        const int64_t q = P62;
        const size_t dim = 257;
        int64_t seeds[3][4] = {{1, 2, 3, 4}, {0, 0, 0, 0}, {0xFFFFFFFFLL, 7, 8, 9}};
        int64_t *mask = malloc(dim * 8), *masked = malloc(3 * dim * 8), *sec = malloc(dim * 8), *tm = malloc(dim * 8),
                *ts = malloc(dim * 8), *out = malloc(dim * 8);
        for (size_t i = 0; i < dim; ++i) sec[i] = (int64_t)i * 1000003 - 5;
        for (int p = 0; p < 3; ++p) {
            sdao_chacha_expand(seeds[p], 4, q, dim, mask);
            sdao_addsub(sec, mask, dim, q, 0, masked + p * dim, 0);
        }
        sdao_chacha_combine(&seeds[0][0], 3, 4, q, dim, tm, 0);
        sdao_combine(q, masked, 3, dim, dim, ts, 0);
        sdao_addsub(ts, tm, dim, q, 1, out, 0);
        int ok = 1;
        for (size_t i = 0; i < dim; ++i) {
            const __int128 w = ((__int128)3 * sec[i]) % q;
            ok &= out[i] == (int64_t)((w + q) % q);
        }
        CHECK(ok, "chacha mask / combine / unmask");
        free(mask); free(masked); free(sec); free(tm); free(ts); free(out);
    }
    /* DRBG, synthetic fill, baseline pass: exact-size buffers */
    {
        uint8_t key[32];
        for (int i = 0; i < 32; ++i) key[i] = (uint8_t)i;
        int64_t* d = malloc(37 * 3 * 8);
        sdao_drbg_fill(key, 20, 5, 37, 3, P62, d);
        int ok = 1;
        for (int i = 0; i < 111; ++i) ok &= d[i] >= 0 && d[i] < P62;
        CHECK(ok, "drbg range");
        int64_t* s = malloc(3 * 50 * 8);
        sdao_fill_synthetic(s, 3, 50, 50, 7, 0x5DA5DA5DA5DA5DA5ULL, P62);
        int64_t* sums = malloc(8 * 17 * 8);                    /* B = ceil(50/3) = 17 */
        CHECK(sdao_baseline_pass(1, P62, 8, 3, 1, W8, W9, 3, 50, 0, 1, key, sums) == 150, "baseline_pass packed");
        free(sums);
        sums = malloc(3 * 50 * 8);
        CHECK(sdao_baseline_pass(0, P62, 3, 1, 2, W8, W9, 3, 50, 0, 1, key, sums) == 150, "baseline_pass additive");
        free(d); free(s); free(sums);
    }
    /* wire codec: every length, exact-size output */
    {
        int64_t v[140];
        size_t n = 0;
        for (int b = 0; b < 64; ++b) { v[n++] = (int64_t)((uint64_t)1 << b); v[n++] = -(int64_t)((uint64_t)1 << (b < 63 ? b : 62)) - 1; }
        v[n++] = INT64_MAX; v[n++] = INT64_MIN; v[n++] = 0; v[n++] = -1;
        uint8_t* raw = malloc(n * 10);
        const size_t nb = sdao_varint_encode(v, n, raw);
        uint8_t* exact = malloc(nb);
        memcpy(exact, raw, nb);
        int64_t* back = malloc(n * 8);
        CHECK(sdao_varint_decode(exact, nb, back, n) == n && memcmp(back, v, n * 8) == 0, "varint round trip");
        free(raw); free(exact); free(back);
    }
    if (!failures) printf("oracle selftest: OK\n");
    return failures ? 1 : 0;
}
