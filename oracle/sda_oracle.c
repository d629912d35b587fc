/*
 * sda_oracle.c - CPU ORACLE for the SDA sharing / masking hot path (plain C, gcc).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.  The shipped
 * library (sda_amd/lib/libsda_hip.so) does not link, load or call anything in oracle/.
 *
 * It restates, loop for loop, the scalar single-threaded Rust of the reference
 * (paths relative to /root/reference), each function citing the file:line it follows.  64-bit
 * products are widened to unsigned __int128 - the mathematically exact result the reference would
 * need for the 62-bit configurations (tss 0.2 multiplies i64 without widening and is only defined
 * for p < ~3e9).
 *
 * Value modes:  mode 0 = canonical residues in [0,q);  mode 1 =
 * rust_signed, i.e. Rust's truncated `%` (C's `%` on int64_t has the same semantics), values in
 * (-q,q) exactly as the reference produces them.  Packed Shamir is canonical only here (the signed
 * FFT path lives in oracle/pyoracle.py, which pins this file in tests/test_oracle.py).
 *
 * Third-party algorithms absent from /root/reference and restated from their published form:
 * threshold-secret-sharing 0.2 (packed share / reconstruct) and rand 0.3 (ChaChaRng, gen_range);
 * see oracle/pyoracle.py for the KATs that pin them.  PARITY PINNING: end-to-end by the
 * reference's full_loop.rs / README vectors (tests/golden); share-level values are "parity
 * unpinned" by anything inside /root/reference (OsRng there) - pinned by recalled crate KATs only.
 *
 * Also restated here: "sda-drbg-v1", the product's own on-device CSPRNG stream layout,
 * so that device-generated randomness can be reproduced on the CPU.  That part has no reference
 * counterpart (the reference uses OsRng).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

#define SDAO_OK 0
#define SDAO_ERR_WRONG_DIMENSION (-5)
#define SDAO_ERR_NOT_ENOUGH_SHARES (-4)
#define SDAO_ERR_INVALID (-8)

/* ---- integer semantics ----------------------------------------------------------------------- */
static inline int64_t rem_mode(int64_t a, int64_t q, int mode) {
    int64_t r = a % q;                 /* truncated, like Rust */
    if (mode == 0 && r < 0) r += q;
    return r;
}
static inline uint64_t canon(int64_t a, uint64_t q) {
    __int128 r = (__int128)a % (__int128)q;
    if (r < 0) r += q;
    return (uint64_t)r;
}
static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    while (e) {
        if (e & 1) r = mulmod(r, b, q);
        b = mulmod(b, b, q);
        e >>= 1;
    }
    return r;
}
static int invmod(uint64_t a, uint64_t q, uint64_t* out) {
    __int128 r0 = q, r1 = a % q, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 d = r0 / r1;
        __int128 r2 = r0 - d * r1; r0 = r1; r1 = r2;
        __int128 t2 = t0 - d * t1; t0 = t1; t1 = t2;
    }
    if (r0 != 1) return -1;
    if (t0 < 0) t0 += q;
    *out = (uint64_t)t0;
    return 0;
}

/* RecipientOutput::positive - client/src/receive.rs:13-21 */
void sdao_positive(const int64_t* v, size_t len, int64_t modulus, int64_t* out) {
    for (size_t i = 0; i < len; ++i) out[i] = v[i] < 0 ? v[i] + modulus : v[i];
}

/* ================================================================================================
 * Additive sharing: batched.rs:18-53 driving additive.rs:32-51
 *   rand: the (n-1) OsRng draws per secret (additive.rs:42-44), injected, batch-major [len][n-1]
 *   out : clerk-major [n][len]  (batched.rs:46-48)
 * ============================================================================================== */
int sdao_additive_generate(int64_t q, int n, const int64_t* secrets, size_t len, const int64_t* rand,
                           int64_t* out, int mode) {
    if (q < 1 || n < 1) return SDAO_ERR_INVALID;
    for (size_t b = 0; b < len; ++b) {                 /* batch_input_size == 1: batch = element */
        int64_t acc = secrets[b];
        if (mode == 0) acc = (int64_t)canon(acc, (uint64_t)q);
        for (int i = 0; i < n - 1; ++i) {
            int64_t r = rand[b * (size_t)(n - 1) + i];
            if (mode == 0) r = (int64_t)canon(r, (uint64_t)q);
            out[(size_t)i * len + b] = r;
            acc = rem_mode(acc - r, q, mode);          /* additive.rs:47 fold */
        }
        out[(size_t)(n - 1) * len + b] = acc;
    }
    return SDAO_OK;
}

/* ================================================================================================
 * Combiner::combine - combiner.rs:15-29 (same loop: full.rs:37-52, additive.rs:55-73)
 *   shares: P rows of L values, row p at shares + p*stride
 * ============================================================================================== */
int sdao_combine(int64_t q, const int64_t* shares, size_t P, size_t L, size_t stride, int64_t* out, int mode) {
    for (size_t i = 0; i < L; ++i) out[i] = 0;
    for (size_t p = 0; p < P; ++p) {
        const int64_t* row = shares + p * stride;
        for (size_t ix = 0; ix < L; ++ix) {
            int64_t v = row[ix];
            if (mode == 0) v = (int64_t)canon(v, (uint64_t)q);
            out[ix] += v;                               /* combiner.rs:23 */
            out[ix] = rem_mode(out[ix], q, mode);       /* combiner.rs:24 : one idiv per element */
        }
    }
    return SDAO_OK;
}

/* ================================================================================================
 * Packed Shamir (tss 0.2, recalled - SURVEY.md Appendix B), matrix form.
 *   share_j = f(w3^(j+1)), f the polynomial with f(w2^0)=0, f(w2^i)=secret_i, f(w2^(k+j))=draw_j.
 * ============================================================================================== */
/* L[e][i] = l_i(evals[e]) on `nodes`; column 0 dropped; row-major [ne][nn-1] */
static int lagrange_matrix(const uint64_t* nodes, size_t nn, const uint64_t* evals, size_t ne, uint64_t p,
                           uint64_t* out) {
    for (size_t i = 1; i < nn; ++i) {
        uint64_t den = 1, den_inv;
        for (size_t l = 0; l < nn; ++l)
            if (l != i) den = mulmod(den, (nodes[i] + p - nodes[l]) % p, p);
        if (invmod(den, p, &den_inv)) return SDAO_ERR_INVALID;
        for (size_t e = 0; e < ne; ++e) {
            uint64_t num = 1;
            for (size_t l = 0; l < nn; ++l)
                if (l != i) num = mulmod(num, (evals[e] + p - nodes[l]) % p, p);
            out[e * (nn - 1) + (i - 1)] = mulmod(num, den_inv, p);
        }
    }
    return SDAO_OK;
}

int sdao_packed_share_matrix(int64_t prime, int k, int t, int n, int64_t omega_secrets, int64_t omega_shares,
                             uint64_t* M /* [n][k+t] */) {
    const uint64_t p = (uint64_t)prime;
    const int m = k + t;
    uint64_t* nodes = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    uint64_t* evals = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    for (int e = 0; e <= m; ++e) nodes[e] = powmod(canon(omega_secrets, p), (uint64_t)e, p);
    for (int j = 0; j < n; ++j) evals[j] = powmod(canon(omega_shares, p), (uint64_t)j + 1, p);
    int st = lagrange_matrix(nodes, (size_t)m + 1, evals, (size_t)n, p, M);
    free(nodes); free(evals);
    return st;
}

/* batched.rs:18-53 driving packed_shamir.rs:40-43; rand = t draws per batch, batch-major [B][t];
 * out clerk-major [n][B]; last batch zero-padded (batched.rs:37-43). */
int sdao_packed_generate(int64_t prime, int k, int t, int n, int64_t omega_secrets, int64_t omega_shares,
                         const int64_t* secrets, size_t len, const int64_t* rand, int64_t* out) {
    const uint64_t p = (uint64_t)prime;
    const int m = k + t;
    const size_t B = (len + (size_t)k - 1) / (size_t)k;               /* batched.rs:23 */
    uint64_t* M = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n * (size_t)m);
    uint64_t* v = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)m);
    int st = sdao_packed_share_matrix(prime, k, t, n, omega_secrets, omega_shares, M);
    if (st == SDAO_OK) {
        for (size_t b = 0; b < B; ++b) {
            for (int i = 0; i < k; ++i) {
                size_t e = b * (size_t)k + (size_t)i;
                v[i] = e < len ? canon(secrets[e], p) : 0;            /* pad */
            }
            for (int i = 0; i < t; ++i) v[k + i] = canon(rand[b * (size_t)t + (size_t)i], p);
            for (int j = 0; j < n; ++j) {                             /* generate_for_batch */
                uint64_t acc = 0;
                for (int i = 0; i < m; ++i) acc = (acc + mulmod(M[(size_t)j * m + i], v[i], p)) % p;
                out[(size_t)j * B + b] = (int64_t)acc;                /* scatter, batched.rs:46-48 */
            }
        }
    }
    free(M); free(v);
    return st;
}

/* The library's CSPRNG share map (include/sda_hip.h "CSPRNG share map"; no reference counterpart - the reference draws
 * from OsRng inside tss): per batch the t draws ARE shares 0..t-1 and shares t..n-1 are the values at w3^(t+1..n) of the
 * polynomial of degree <= t+k through (1, 0), (w2^i, secret_i) i = 1..k, (w3^(j+1), draw_j) j < t.  Restated with a
 * per-batch Newton interpolation (the product multiplies by one precomputed Lagrange matrix).  implied (may be NULL):
 * [B][t] = the same polynomial's values at w2^(k+1..k+t), i.e. the randomness tss's own share() would have needed for
 * these very shares - sdao_packed_generate(secrets, implied) must reproduce `out`. */
int sdao_packed_generate_systematic(int64_t prime, int k, int t, int n, int64_t omega_secrets, int64_t omega_shares,
                                    const int64_t* secrets, size_t len, const int64_t* draws, int64_t* out, int64_t* implied) {
    const uint64_t p = (uint64_t)prime;
    if (n < t) return SDAO_ERR_INVALID;
    const size_t np = (size_t)(k + t + 1);
    const size_t B = (len + (size_t)k - 1) / (size_t)k;
    uint64_t* pts = (uint64_t*)malloc(sizeof(uint64_t) * np);
    uint64_t* dd = (uint64_t*)malloc(sizeof(uint64_t) * np);
    uint64_t* invd = (uint64_t*)malloc(sizeof(uint64_t) * np * np);
    const uint64_t w2 = canon(omega_secrets, p), w3 = canon(omega_shares, p);
    for (int i = 0; i <= k; ++i) pts[i] = powmod(w2, (uint64_t)i, p);
    for (int j = 0; j < t; ++j) pts[k + 1 + j] = powmod(w3, (uint64_t)j + 1, p);
    int st = SDAO_OK;
    for (size_t i = 0; i < np && st == SDAO_OK; ++i)
        for (size_t j = 0; j < i; ++j)
            if (invmod((pts[i] + p - pts[j]) % p, p, &invd[i * np + j])) { st = SDAO_ERR_INVALID; break; }   /* colliding points */
    for (size_t b = 0; b < B && st == SDAO_OK; ++b) {
        dd[0] = 0;
        for (int i = 0; i < k; ++i) {
            size_t e = b * (size_t)k + (size_t)i;
            dd[1 + i] = e < len ? canon(secrets[e], p) : 0;                  /* pad, batched.rs:37-43 */
        }
        for (int j = 0; j < t; ++j) {
            dd[k + 1 + j] = canon(draws[b * (size_t)t + (size_t)j], p);
            out[(size_t)j * B + b] = (int64_t)dd[k + 1 + j];                 /* share j = draw j */
        }
        for (size_t j = 1; j < np; ++j)
            for (size_t i = np - 1; i >= j; --i)
                dd[i] = mulmod((dd[i] + p - dd[i - 1]) % p, invd[i * np + (i - j)], p);
        for (int q = t; q < n + (implied ? t : 0); ++q) {
            const int is_share = q < n;
            const uint64_t x = is_share ? powmod(w3, (uint64_t)q + 1, p) : powmod(w2, (uint64_t)(k + 1 + (q - n)), p);
            uint64_t acc = 0, prod = 1;
            for (size_t i = 0; i < np; ++i) {
                acc = (acc + mulmod(dd[i], prod, p)) % p;
                prod = mulmod(prod, (x + p - pts[i]) % p, p);
            }
            if (is_share) out[(size_t)q * B + b] = (int64_t)acc;
            else implied[b * (size_t)t + (size_t)(q - n)] = (int64_t)acc;
        }
    }
    free(pts); free(dd); free(invd);
    return st;
}

/* batched.rs:68-97 driving packed_shamir.rs:73-77 -> tss reconstruct: per batch, Newton interpolation
 * through (1,0) and (w3^(idx+1), share) and evaluation at w2^e, e=1..k - recomputed for EVERY batch,
 * as the reference does.  shares: row c at shares + c*stride. */
int sdao_packed_reconstruct(int64_t prime, int k, int t, int64_t omega_secrets, int64_t omega_shares,
                            size_t dimension, const size_t* indices, size_t n_idx, const int64_t* shares,
                            size_t stride, int64_t* out) {
    const uint64_t p = (uint64_t)prime;
    if (n_idx < (size_t)(k + t)) return SDAO_ERR_NOT_ENOUGH_SHARES;   /* packed_shamir.rs:75 */
    const size_t B = (dimension + (size_t)k - 1) / (size_t)k;         /* batched.rs:77 */
    const size_t np = n_idx + 1;
    uint64_t* pts = (uint64_t*)malloc(sizeof(uint64_t) * np);
    uint64_t* dd = (uint64_t*)malloc(sizeof(uint64_t) * np);
    uint64_t* invd = (uint64_t*)malloc(sizeof(uint64_t) * np * np);
    pts[0] = 1;
    for (size_t c = 0; c < n_idx; ++c) pts[c + 1] = powmod(canon(omega_shares, p), (uint64_t)indices[c] + 1, p);
    int st = SDAO_OK;
    /* the point differences' inverses are batch-independent; caching them only spares the oracle the
     * extended-Euclid calls, the O(n'^2) divided-difference table is still rebuilt per batch */
    for (size_t i = 0; i < np && st == SDAO_OK; ++i)
        for (size_t j = 0; j < i; ++j)
            if (invmod((pts[i] + p - pts[j]) % p, p, &invd[i * np + j])) { st = SDAO_ERR_INVALID; break; }
    for (size_t b = 0; b < B && st == SDAO_OK; ++b) {
        dd[0] = 0;
        for (size_t c = 0; c < n_idx; ++c) dd[c + 1] = canon(shares[c * stride + b], p);   /* batched.rs:83-85 */
        for (size_t j = 1; j < np; ++j)                                   /* newton_interpolation_general */
            for (size_t i = np - 1; i >= j; --i)
                dd[i] = mulmod((dd[i] + p - dd[i - 1]) % p, invd[i * np + (i - j)], p);
        for (int e = 1; e <= k; ++e) {                                    /* newton_evaluate at w2^e */
            size_t o = b * (size_t)k + (size_t)(e - 1);
            if (o >= dimension) break;                                    /* truncate, batched.rs:94 */
            uint64_t x = powmod(canon(omega_secrets, p), (uint64_t)e, p);
            uint64_t acc = 0, np_prod = 1;
            for (size_t i = 0; i < np; ++i) {
                acc = (acc + mulmod(dd[i], np_prod, p)) % p;
                np_prod = mulmod(np_prod, (x + p - pts[i]) % p, p);
            }
            out[o] = (int64_t)acc;
        }
    }
    free(pts); free(dd); free(invd);
    return st;
}

/* ================================================================================================
 * ChaCha (RFC 7539 block function), rand 0.3 ChaChaRng + gen_range  (recalled - Appendix C)
 * ============================================================================================== */
#define ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d)                 \
    a += b; d ^= a; d = ROTL32(d, 16); \
    c += d; b ^= c; b = ROTL32(b, 12); \
    a += b; d ^= a; d = ROTL32(d, 8);  \
    c += d; b ^= c; b = ROTL32(b, 7);

void sdao_chacha_block(const uint32_t in[16], int rounds, uint32_t out[16]) {
    uint32_t x[16];
    memcpy(x, in, sizeof x);
    for (int r = 0; r < rounds / 2; ++r) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

static const uint32_t CHACHA_CONST[4] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u};

typedef struct {
    uint32_t state[16];
    uint32_t buf[16];
    int index;
} chacha_rng;

static void rng_from_seed(chacha_rng* r, const uint32_t* seed, size_t n_words) {   /* ChaChaRng::from_seed */
    memcpy(r->state, CHACHA_CONST, 16);
    for (int i = 0; i < 8; ++i) r->state[4 + i] = (size_t)i < n_words ? seed[i] : 0;
    r->state[12] = r->state[13] = r->state[14] = r->state[15] = 0;
    r->index = 16;
}
static uint32_t rng_next_u32(chacha_rng* r) {
    if (r->index == 16) {
        sdao_chacha_block(r->state, 20, r->buf);
        r->index = 0;
        for (int i = 12; i < 16; ++i)                     /* 128-bit counter */
            if (++r->state[i] != 0) break;
    }
    return r->buf[r->index++];
}
static uint64_t rng_next_u64(chacha_rng* r) {             /* high word first */
    uint64_t hi = rng_next_u32(r);
    return (hi << 32) | rng_next_u32(r);
}
static int64_t rng_gen_range0(chacha_rng* r, int64_t high) {   /* gen_range(0_i64, high) */
    const uint64_t range = (uint64_t)high;
    const uint64_t zone = UINT64_MAX - (UINT64_MAX % range);
    for (;;) {
        uint64_t v = rng_next_u64(r);
        if (v < zone) return (int64_t)(v % range);
    }
}

/* chacha.rs:36-39: `count` masks from a seed (words used as u32, chacha.rs:62-64) */
void sdao_chacha_expand(const int64_t* seed_words, size_t n_words, int64_t q, size_t count, int64_t* out) {
    uint32_t seed[8] = {0};
    for (size_t i = 0; i < n_words && i < 8; ++i) seed[i] = (uint32_t)(uint64_t)seed_words[i];
    chacha_rng r;
    rng_from_seed(&r, seed, n_words < 8 ? n_words : 8);
    for (size_t i = 0; i < count; ++i) out[i] = rng_gen_range0(&r, q);
}

/* chacha.rs:56-77: re-expand every seed and accumulate; seeds: P rows of n_words at stride n_words */
void sdao_chacha_combine(const int64_t* seeds, size_t P, size_t n_words, int64_t q, size_t dimension,
                         int64_t* out, int mode) {
    for (size_t i = 0; i < dimension; ++i) out[i] = 0;
    for (size_t p = 0; p < P; ++p) {
        uint32_t seed[8] = {0};
        for (size_t i = 0; i < n_words && i < 8; ++i) seed[i] = (uint32_t)(uint64_t)seeds[p * n_words + i];
        chacha_rng r;
        rng_from_seed(&r, seed, n_words < 8 ? n_words : 8);
        for (size_t i = 0; i < dimension; ++i) {
            int64_t m = rng_gen_range0(&r, q);
            out[i] += m;
            out[i] = rem_mode(out[i], q, mode);           /* chacha.rs:70-71 */
        }
    }
}

/* element-wise (a + b) % q  /  (a - b) % q : full.rs:30,62; chacha.rs:44,88 */
void sdao_addsub(const int64_t* a, const int64_t* b, size_t len, int64_t q, int subtract, int64_t* out, int mode) {
    for (size_t i = 0; i < len; ++i) {
        int64_t x = a[i], y = b[i];
        if (mode == 0) { x = (int64_t)canon(x, (uint64_t)q); y = (int64_t)canon(y, (uint64_t)q); }
        out[i] = rem_mode(subtract ? x - y : x + y, q, mode);
    }
}

/* ================================================================================================
 * sda-drbg-v1 (product's own CSPRNG layout) - no reference counterpart
 * ============================================================================================== */
static int lemire(uint64_t x, uint64_t m, uint64_t thr, uint64_t* out) {
    u128 pr = (u128)x * m;
    *out = (uint64_t)(pr >> 64);
    return (uint64_t)pr >= thr;
}

/* the PAIRED rule for m <= 0x7F7F7F (m m < 2^46): one candidate word yields two draws - Lemire's method with range m^2, read off
 * in two steps (x m = ra 2^64 + l1, l1 m = rb 2^64 + lo; accept iff lo >= 2^64 mod m^2): (ra, rb) uniform in [0, m)^2 */
#define SDAO_DRBG_PAIRED_MAX 0x7F7F7Full
static int lemire_pair(uint64_t x, uint64_t m, uint64_t thr2, uint64_t* ra, uint64_t* rb) {
    u128 p1 = (u128)x * m;
    *ra = (uint64_t)(p1 >> 64);
    u128 p2 = (u128)(uint64_t)p1 * m;
    *rb = (uint64_t)(p2 >> 64);
    return (uint64_t)p2 >= thr2;
}

static void drbg_state(const uint32_t key[8], uint64_t I, uint64_t stream, uint32_t attempt, uint32_t st[16]) {
    memcpy(st, CHACHA_CONST, 16);
    memcpy(st + 4, key, 32);
    st[12] = (uint32_t)I;
    st[13] = (uint32_t)(I >> 32);
    st[14] = (uint32_t)stream;
    st[15] = ((uint32_t)(stream >> 32) & 0xFFFFFFu) | (attempt << 24);
}

/* sda-drbg-v1 call key (sda_amd/csrc/host_chacha.hpp): words 0..7 of the ChaCha20 block keyed with the
 * handle's master key, block counter = call index, nonce words "sdak" "dfv1". */
void sdao_drbg_call_key(const uint8_t master_bytes[32], uint64_t call_index, uint8_t out_bytes[32]) {
    uint32_t st[16], o[16];
    memcpy(st, CHACHA_CONST, 16);
    for (int i = 0; i < 8; ++i)
        st[4 + i] = (uint32_t)master_bytes[4 * i] | ((uint32_t)master_bytes[4 * i + 1] << 8) |
                    ((uint32_t)master_bytes[4 * i + 2] << 16) | ((uint32_t)master_bytes[4 * i + 3] << 24);
    st[12] = (uint32_t)call_index;
    st[13] = (uint32_t)(call_index >> 32);
    st[14] = 0x6b616473u;
    st[15] = 0x31766664u;
    sdao_chacha_block(st, 20, o);
    for (int i = 0; i < 8; ++i) {
        out_bytes[4 * i] = (uint8_t)o[i];
        out_bytes[4 * i + 1] = (uint8_t)(o[i] >> 8);
        out_bytes[4 * i + 2] = (uint8_t)(o[i] >> 16);
        out_bytes[4 * i + 3] = (uint8_t)(o[i] >> 24);
    }
}

/* out[b*T + i] for b < batches, i < T */
void sdao_drbg_fill(const uint8_t key_bytes[32], int rounds, uint64_t stream, size_t batches, uint32_t T,
                    int64_t modulus, int64_t* out) {
    uint32_t key[8];
    for (int i = 0; i < 8; ++i)
        key[i] = (uint32_t)key_bytes[4 * i] | ((uint32_t)key_bytes[4 * i + 1] << 8) |
                 ((uint32_t)key_bytes[4 * i + 2] << 16) | ((uint32_t)key_bytes[4 * i + 3] << 24);
    const uint64_t m = (uint64_t)modulus;
    const uint64_t thr = (uint64_t)((((u128)1) << 64) % m);
    const size_t groups = (batches + 7) / 8;
    uint32_t st[16], o[16];
    if (m <= SDAO_DRBG_PAIRED_MAX) {
        /* draws 2j, 2j + 1 of a batch from ONE candidate word: block counter g * ceil(T / 2) + j; retry blocks b * ceil(T / 2) + j */
        const uint64_t thr2 = (uint64_t)((((u128)1) << 64) % ((u128)m * m));
        const uint32_t T2 = (T + 1) / 2;
        for (size_t g = 0; g < groups; ++g)
            for (uint32_t j = 0; j < T2; ++j) {
                drbg_state(key, (uint64_t)g * T2 + j, stream, 0, st);
                sdao_chacha_block(st, rounds, o);
                for (int c = 0; c < 4; ++c)
                    for (int e = 0; e < 2; ++e) {
                        size_t b = g * 8 + (size_t)(2 * c + e);
                        if (b >= batches) continue;
                        uint64_t x = ((uint64_t)o[8 * e + c] << 32) | o[8 * e + 4 + c], ra, rb;
                        if (!lemire_pair(x, m, thr2, &ra, &rb)) {
                            int done = 0;
                            for (uint32_t a = 1; a < 256 && !done; ++a) {
                                uint32_t st2[16], o2[16];
                                drbg_state(key, (uint64_t)b * T2 + j, stream, a, st2);
                                sdao_chacha_block(st2, rounds, o2);
                                for (int q = 0; q < 8 && !done; ++q)
                                    done = lemire_pair(((uint64_t)o2[2 * q] << 32) | o2[2 * q + 1], m, thr2, &ra, &rb);
                            }
                        }
                        out[b * T + 2 * j] = (int64_t)ra;
                        if (2 * j + 1 < T) out[b * T + 2 * j + 1] = (int64_t)rb;
                    }
            }
        return;
    }
    for (size_t g = 0; g < groups; ++g) {
        for (uint32_t i = 0; i < T; ++i) {
            drbg_state(key, (uint64_t)g * T + i, stream, 0, st);
            sdao_chacha_block(st, rounds, o);
            for (int c = 0; c < 4; ++c) {
                for (int e = 0; e < 2; ++e) {
                    size_t b = g * 8 + (size_t)(2 * c + e);
                    if (b >= batches) continue;
                    uint64_t x = ((uint64_t)o[8 * e + c] << 32) | o[8 * e + 4 + c];
                    uint64_t val;
                    if (!lemire(x, m, thr, &val)) {           /* retry stream */
                        int done = 0;
                        for (uint32_t a = 1; a < 256 && !done; ++a) {
                            uint32_t st2[16], o2[16];
                            drbg_state(key, (uint64_t)b * T + i, stream, a, st2);
                            sdao_chacha_block(st2, rounds, o2);
                            for (int j = 0; j < 8 && !done; ++j) {
                                uint64_t x2 = ((uint64_t)o2[2 * j] << 32) | o2[2 * j + 1];
                                done = lemire(x2, m, thr, &val);
                            }
                        }
                    }
                    out[b * T + i] = (int64_t)val;
                }
            }
        }
    }
}

/* synthetic bench input, SURVEY.md 8d */
static uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void sdao_fill_synthetic(int64_t* out, size_t participants, size_t len, size_t stride, uint64_t first_participant,
                         uint64_t seed, int64_t modulus) {
    for (size_t p = 0; p < participants; ++p)
        for (size_t i = 0; i < len; ++i)
            out[p * stride + i] = (int64_t)(splitmix64(seed ^ (((first_participant + p) << 32) | i)) % (uint64_t)modulus);
}

/* ================================================================================================
 * cpu_baseline leg of bench.py: the reference-faithful scalar path (share-gen + clerk-sum) over
 * `participants` synthetic vectors, one after another like the reference's single-threaded loops.
 * Randomness comes from the buffered CSPRNG above (a concession in the reference's favour: it
 * makes one OsRng call per draw).  Returns the number of (participant, component) elements done;
 * clerk_sums is [n][B].
 * ============================================================================================== */
size_t sdao_baseline_pass(int packed, int64_t modulus, int n, int k, int t, int64_t omega_secrets,
                          int64_t omega_shares, size_t participants, size_t len, uint64_t first_participant,
                          uint64_t seed, const uint8_t key_bytes[32], int64_t* clerk_sums) {
    const size_t kk = packed ? (size_t)k : 1, T = packed ? (size_t)t : (size_t)(n - 1);
    const size_t B = (len + kk - 1) / kk;
    int64_t* secrets = (int64_t*)malloc(len * 8);
    int64_t* rnd = (int64_t*)malloc((B * T + 1) * 8);
    int64_t* shares = (int64_t*)malloc((size_t)n * B * 8);
    int64_t* tmp = (int64_t*)malloc(B * 8);
    memset(clerk_sums, 0, (size_t)n * B * 8);
    for (size_t p = 0; p < participants; ++p) {
        sdao_fill_synthetic(secrets, 1, len, len, first_participant + p, seed, modulus);
        sdao_drbg_fill(key_bytes, 20, first_participant + p, B, (uint32_t)T, modulus, rnd);
        if (packed) sdao_packed_generate(modulus, k, t, n, omega_secrets, omega_shares, secrets, len, rnd, shares);
        else sdao_additive_generate(modulus, n, secrets, len, rnd, shares, 0);
        for (int j = 0; j < n; ++j) {                     /* each clerk's running combine, combiner.rs:20-26 */
            int64_t* acc = clerk_sums + (size_t)j * B;
            const int64_t* row = shares + (size_t)j * B;
            for (size_t ix = 0; ix < B; ++ix) {
                acc[ix] += row[ix];
                acc[ix] %= modulus;
            }
        }
    }
    free(secrets); free(rnd); free(shares); free(tmp);
    return participants * len;
}

/* ================================================================================================
 * Share-vector wire codec: integer-encoding 1.0 `VarInt for i64` [recalled],
 * call sites client/src/crypto/encryption/sodium.rs:36-41 (encode) and :83-89 (decode).
 * ============================================================================================== */
size_t sdao_varint_encode(const int64_t* values, size_t len, uint8_t* out) {
    size_t pos = 0;
    for (size_t i = 0; i < len; ++i) {
        uint64_t n = ((uint64_t)values[i] << 1) ^ (uint64_t)(values[i] >> 63);   /* zig-zag */
        while (n >= 0x80) { out[pos++] = (uint8_t)(0x80 | (n & 0x7F)); n >>= 7; }
        out[pos++] = (uint8_t)n;
    }
    return pos;
}

/* decodes until the input is exhausted or `cap` values were produced; returns the value count */
size_t sdao_varint_decode(const uint8_t* src, size_t n_bytes, int64_t* out, size_t cap) {
    size_t pos = 0, count = 0;
    while (pos < n_bytes && count < cap) {
        uint64_t result = 0;
        unsigned shift = 0;
        while (pos < n_bytes) {
            uint8_t b = src[pos++];
            result |= (uint64_t)(b & 0x7F) << (shift & 63);
            shift += 7;
            if ((b & 0x80) == 0 || shift > 70) break;
        }
        out[count++] = (int64_t)((result >> 1) ^ (uint64_t)(-(int64_t)(result & 1)));
    }
    return count;
}
